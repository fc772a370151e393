"""On-disk formats (features CSV, phamer_scores.csv) against files written by the reference's own
writer functions (tests/golden/files.json), byte for byte.  The native FASTA reader against a plain
statement of Bio.SeqIO's FASTA record rules.  CPU-only except where marked."""
import argparse
import gzip
import os

import numpy as np
import pytest

from tests import helpers


def test_writers_are_byte_compatible(tmp_path):
    from phamers_amd import fileIO
    doc = helpers.load_json("files.json")
    arr = helpers.load_npz("files.npz")
    ids = np.array(doc["ids"])
    args = argparse.Namespace(**{k: doc["args"][k] for k in
                                 ("input_file", "kmer_length", "output_file", "symbols", "verbose", "sample",
                                  "file_identifier", "debug")})
    out = {}
    fileIO.save_counts(arr["counts"], ids, str(tmp_path / "f_noargs.csv"))
    fileIO.save_counts(arr["counts"], ids, str(tmp_path / "f_args.csv"), args=args)
    fileIO.save_phamer_scores(ids, arr["scores"], str(tmp_path / "s_noargs.csv"))
    fileIO.save_phamer_scores(ids, arr["scores"], str(tmp_path / "s_args.csv"), args=args)
    for name, want in doc["files"].items():
        out[name] = open(tmp_path / name).read()
        assert out[name] == want, name
    got = fileIO.read_phamer_output(str(tmp_path / "s_args.csv"))
    assert got == doc["scores_read_back"]


def test_native_writers_keep_row_order_across_chunks_and_threads(tmp_path):
    """More rows than one formatting chunk (4096) per worker thread: the chunks are formatted and written by different
    threads at offsets handed from chunk to chunk; the file must be the rows in order, byte for byte."""
    from phamers_amd import fileIO
    n = 5 * 4096 + 17
    rng = np.random.default_rng(3)
    counts = rng.integers(0, 10 ** rng.integers(1, 9, size=(n, 1)), size=(n, 6)).astype(np.uint32)
    ids = np.array(["contig_%d" % (i * 7919 % 100003) for i in range(n)])
    path = str(tmp_path / "features.csv")
    fileIO.save_counts(counts, ids, path, header="K-mer count file")
    body = [ln for ln in open(path).read().split("\n") if ln and not ln.startswith("#")]
    assert body == [",".join([ids[i]] + [str(int(v)) for v in counts[i]]) for i in range(n)]
    scores = rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8, size=n)
    spath = str(tmp_path / "scores.csv")
    fileIO.save_phamer_scores(ids, scores, spath)
    body = [ln for ln in open(spath).read().split("\n") if ln and not ln.startswith("#")]
    assert body == ["%s, %s" % (ids[i], str(np.float64(scores[i]))) for i in range(n)]


def test_native_float_notation_is_numpys():
    """phk_write_scores_csv writes a score as str(numpy.float64) does (what the reference's astype(str) yields)."""
    import ctypes
    from phamers_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(5000), np.tanh(rng.standard_normal(5000)) + rng.choice([-1.0, 1.0], 5000),
                           10.0 ** rng.uniform(-12, 20, 5000) * rng.choice([-1, 1], 5000),
                           [0.0, -0.0, 1.0, 1e-4, 1e-5, 123456.0, 1e16, 9.999e15, np.nan, np.inf, -np.inf, 5e-324,
                            1.7976931348623157e308, 0.1, 100.0, 0.00012345]])
    buf = ctypes.create_string_buffer(64)
    for v in vals:
        _lib.check(lib.phk_format_float(float(v), buf, 64))
        assert buf.value.decode() == str(np.float64(v)), repr(v)


def test_generate_summary_text():
    from phamers_amd import fileIO
    doc = helpers.load_json("files.json")
    args = argparse.Namespace(**{k: doc["args"][k] for k in
                                 ("input_file", "kmer_length", "output_file", "symbols", "verbose", "sample",
                                  "file_identifier", "debug")})
    text = fileIO.generate_summary(args, header="K-mer count file")
    want_header = "".join(line[2:] if line.startswith("# ") else line[1:]
                          for line in doc["files"]["f_args.csv"].splitlines(keepends=True) if line.startswith("#"))
    assert text == want_header.rstrip("\n") or text.rstrip("\n") == want_header.rstrip("\n")
    assert fileIO.generate_summary(None) == ""


def _seqio_like(text):
    """Bio.SeqIO FASTA record rules restated: skip text before the first '>' line; title = line[1:]
    right-stripped; id = first word; sequence = following lines right-stripped, joined, ' ' and '\\r'
    removed."""
    titles, seqs, cur = [], [], None
    for line in text.split("\n"):
        if line.startswith(">"):
            if cur is not None:
                seqs.append("".join(cur).replace(" ", "").replace("\r", ""))
            titles.append(line[1:].rstrip())
            cur = []
        elif cur is not None:
            cur.append(line.rstrip())
    if cur is not None:
        seqs.append("".join(cur).replace(" ", "").replace("\r", ""))
    return titles, seqs


FASTA_TEXT = ("leading junk\n>SuperContig_1_length_12_ID_7 extra words \nATGC ATGC\r\nNNatgc  \n\n"
              ">gi|1|ref|NC_000001.1| phage\nGGGG\nCC\n>empty\n>CP000084.1 Candidatus\nACGTNNNN\nTT\tA\n>last_ID_9\nA")


@pytest.mark.parametrize("threads", [1, 3, 0])
def test_native_fasta_reader_matches_seqio_rules(tmp_path, threads):
    from phamers_amd import _lib
    from phamers_amd import synth
    text = FASTA_TEXT
    for c in range(40):   # multi-line records of ragged width
        s = synth.synth_contig(5, c, 100 + 37 * c, invalid_ppm=20000)
        w = 7 + c % 60
        text += "\n>" + synth.contig_header(c, len(s)) + "\n" + "\n".join(s[i:i + w] for i in range(0, len(s), w))
    p = tmp_path / "a.fasta"
    p.write_text(text)
    want_titles, want_seqs = _seqio_like(text)
    for path in (str(p), str(p) + ".gz"):
        if path.endswith(".gz"):
            with gzip.open(path, "wt") as g:
                g.write(text)
        f = _lib.Fasta(path, threads=threads)
        assert f.n_records == len(want_titles)
        assert f.titles() == want_titles
        assert f.sequences() == want_seqs
        assert f.ids() == [(t.split(None, 1) or [""])[0] for t in want_titles]
        assert f.lengths().tolist() == [len(s) for s in want_seqs]
        # the index-only pass (phk_fasta_index: the length screen of a run that read its features from the cache):
        # same titles, ids and lengths, no sequences
        g = _lib.Fasta(path, threads=threads, index_only=True)
        assert g.titles() == want_titles and g.lengths().tolist() == f.lengths().tolist()
        assert g.ids() == f.ids()
        with pytest.raises(ValueError):
            g.sequences()
        f.close()
        g.close()
    with pytest.raises(IOError):
        _lib.Fasta(str(tmp_path / "missing.fa"))
    empty = tmp_path / "empty.fa"
    empty.write_text("")
    f = _lib.Fasta(str(empty))
    assert f.n_records == 0 and f.sequences() == []


def test_native_fasta_reader_slices_of_a_large_file(tmp_path):
    """A file big enough to be cut into one slice per thread (>= 1 MB each): records that straddle the cuts, long
    records spanning whole slices, CR / blanks inside lines, text before the first record -- every thread count gives
    the single-thread result, which equals the Bio.SeqIO restatement."""
    from phamers_amd import _lib
    rng = np.random.default_rng(11)
    parts = ["junk before the first record\n"]
    alphabet = np.frombuffer(b"ATGCN", dtype=np.uint8)
    for c in range(260):
        n = 2500000 if c == 77 else int(rng.choice([40, 3000, 70000, 9000]))
        seq = alphabet[rng.integers(0, 5, size=n)].tobytes().decode()
        w = int(rng.integers(20, 200))
        lines = [seq[i:i + w] for i in range(0, n, w)]
        if c % 7 == 0:
            lines = [ln[:5] + " " + ln[5:] + "\r" for ln in lines]       # blanks and CR inside / at the end of lines
        parts.append(">contig_%d_length_%d some description \n" % (c, n) + "\n".join(lines) + ("\n\n" if c % 5 == 0 else "\n"))
    text = "".join(parts)
    assert len(text) > 6 * (1 << 20)   # at least 6 slices of 1 MB
    path = tmp_path / "big.fasta"
    path.write_text(text)
    want_titles, want_seqs = _seqio_like(text)
    ref = None
    for threads in (1, 2, 5, 0):
        f = _lib.Fasta(str(path), threads=threads)
        got = (f.titles(), f.sequences(), f.lengths().tolist(), f.ids())
        f.close()
        g = _lib.Fasta(str(path), threads=threads, index_only=True)
        assert (g.titles(), g.lengths().tolist(), g.ids()) == (got[0], got[2], got[3]), threads
        g.close()
        if ref is None:
            ref = got
            assert got[0] == want_titles and got[1] == want_seqs
        else:
            assert got == ref, threads


def test_fasta_byte_ranges_give_every_record_to_exactly_one_range(tmp_path):
    """phk_fasta_read_range / phk_fasta_read_part (one rank's share of a file): for EVERY cut position of an awkward
    small file -- inside a sequence line, inside a title, on a '>' that does not begin a line, on blank lines, CR LF,
    before the first record -- the two ranges [0, c) and [c, end) together hold every record once, in order; the same
    for n equal parts, plain and gzip; a record belongs to the range its '>' line begins in."""
    from phamers_amd import _lib
    text = ("junk > not a record\n>a_ID_1 title with > inside\nATGC\nAT>GC\r\n\n>b_ID_2\n>c_ID_3 empty before\nGG GG\n"
            ">gi|1|ref|NC_1.1| x\nCCCC\nTT\n\n\n>last_ID_4\nA")
    for c in range(30):
        text += "\n>r_ID_%d\n" % (10 + c) + "\n".join("ATGCN"[(c + j) % 5] * (1 + (c * j) % 9) for j in range(c % 6))
    want_titles, want_seqs = _seqio_like(text)
    path = tmp_path / "awkward.fasta"
    path.write_bytes(text.encode("latin-1"))
    size = len(text.encode("latin-1"))

    def read(**kw):
        f = _lib.Fasta(str(kw.pop("p", path)), threads=1, **kw)
        out = (f.titles(), f.sequences())
        assert f.ids() == [(t.split(None, 1) or [""])[0] for t in out[0]]
        f.close()
        return out

    assert read() == (want_titles, want_seqs)
    starts = [i for i in range(size) if text[i] == ">" and (i == 0 or text[i - 1] == "\n")]
    for c in range(size + 2):
        a, b = read(byte_range=(0, c)), read(byte_range=(c, None))
        assert a[0] + b[0] == want_titles and a[1] + b[1] == want_seqs, c
        assert len(a[0]) == sum(1 for st in starts if st < c), c       # ownership: where the '>' line begins
    for lo, hi in ((7, 7), (40, 41), (size, size + 5)):
        f = _lib.Fasta(str(path), byte_range=(lo, hi))
        assert f.n_records == sum(1 for st in starts if lo <= st < hi)
        f.close()
    gz = tmp_path / "awkward.fasta.gz"
    with gzip.open(gz, "wb") as g:
        g.write(text.encode("latin-1"))
    for n in (1, 2, 3, 5, 8, 64):
        for p_ in (path, gz):
            parts = [read(p=p_, part=(i, n)) for i in range(n)]
            assert sum((x[0] for x in parts), []) == want_titles, (n, p_)
            assert sum((x[1] for x in parts), []) == want_seqs, (n, p_)
    with pytest.raises(_lib.PhkError):
        _lib.Fasta(str(path), part=(3, 3))
    with pytest.raises(IOError):
        _lib.Fasta(str(tmp_path / "missing.fa"), part=(0, 2))


def test_fasta_parts_of_a_large_file_are_balanced_and_complete(tmp_path):
    """Eight ranks' shares of a 6 MB file read by several threads each: complete, in order, and balanced in bytes to
    within the longest record."""
    from phamers_amd import _lib
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ATGC", dtype=np.uint8)
    recs = []
    for c in range(1200):
        n = int(rng.choice([500, 5000, 12000]))
        seq = alphabet[rng.integers(0, 4, size=n)].tobytes().decode()
        recs.append(">SuperContig_%d_length_%d_ID_%d\n" % (c, n, c) + "\n".join(seq[i:i + 70] for i in range(0, n, 70)) + "\n")
    text = "".join(recs)
    path = tmp_path / "many.fasta"
    path.write_text(text)
    whole = _lib.Fasta(str(path))
    want_ids, want_len = list(whole.phamers_ids()), whole.lengths().tolist()
    whole.close()
    got_ids, got_len, share = [], [], []
    for r in range(8):
        f = _lib.Fasta(str(path), threads=3, part=(r, 8))
        got_ids += list(f.phamers_ids())
        got_len += f.lengths().tolist()
        share.append(f.total_bases)
        f.close()
    assert got_ids == want_ids and got_len == want_len
    assert max(share) - min(share) <= 2 * 12000 + 1000


@pytest.mark.gpu
def test_count_file_and_feature_file_round_trip(tmp_path):
    """kmer.count_file on plain and gzip FASTA (ids by the reference's header rules), the soft IOError
    failure, and the features-CSV cache round trip (save_counts -> read_feature_file)."""
    from oracle import oracle
    from phamers_amd import fileIO, kmer, synth
    seqs = [synth.synth_contig(6, c, 5000 if c % 3 else 777, invalid_ppm=5000) for c in range(25)]
    text = "".join(">%s\n%s\n" % (synth.contig_header(c, len(s)), "\n".join(s[i:i + 60] for i in range(0, len(s), 60)))
                   for c, s in enumerate(seqs))
    p = tmp_path / "contigs.fasta"
    p.write_text(text)
    with gzip.open(str(p) + ".gz", "wt") as g:
        g.write(text)
    want = oracle.count(seqs, 4)
    for path in (str(p), str(p) + ".gz"):
        ids, counts = kmer.count_file(path, 4)
        assert ids.tolist() == [str(c) for c in range(25)]
        assert counts.dtype == np.int64 and np.array_equal(counts, want)
    ids, freq = kmer.count_file(str(p), 4, normalize=True)
    assert np.array_equal(freq, oracle.normalize_counts(want))
    assert kmer.count_file(str(tmp_path / "nope.fasta"), 4) == (None, None)
    rid, lens = kmer.fasta_lengths(str(p))
    assert lens.tolist() == [len(s) for s in seqs]
    cache = tmp_path / "contigs_features.csv"
    fileIO.save_counts(counts, ids, str(cache))
    rid, rcounts = fileIO.read_feature_file(str(cache))
    assert rid.tolist() == ids.tolist() and np.array_equal(rcounts, counts)
    rid, rnorm = fileIO.read_feature_file(str(cache), normalize=True)
    assert np.array_equal(rnorm, oracle.normalize_counts(want))


@pytest.mark.gpu
def test_config0_cli_end_to_end(tmp_path):
    """BASELINE configs[0]: 100 synthetic 5 kb contigs through the phamer.py command line (input
    directory with one FASTA, data directory with reference_features/, --equalize_reference); the
    written phamer_scores.csv is compared with the reference's scores for the same contigs."""
    from phamers_amd import fileIO, phamer, synth
    g = helpers.load_npz("scoring_k4.npz")
    ref = helpers.load_npz("ref_features.npz")
    indir, data = tmp_path / "input", tmp_path / "data" / "reference_features"
    indir.mkdir()
    data.mkdir(parents=True)
    fileIO.save_counts(ref["pos_counts"], ref["pos_ids"], str(data / "positive_features.csv"))
    fileIO.save_counts(ref["neg_counts"], ref["neg_ids"], str(data / "negative_features.csv"))
    with open(indir / "contigs.fasta", "w") as f:
        for c in range(100):
            s = synth.synth_contig(0, c, 5000)
            f.write(">%s\n" % synth.contig_header(c, 5000))
            f.write("\n".join(s[i:i + 70] for i in range(0, 5000, 70)) + "\n")
        f.write(">%s\nATGCATGCATGC\n" % synth.contig_header(100, 12))     # screened out: shorter than 5000
    scorer = phamer.main(["-in", str(indir), "-data", str(tmp_path / "data"), "--equalize_reference"])
    out = indir / "phamer_output" / "phamer_scores.csv"
    assert out.exists() and (indir / "contigs_features.csv").exists()      # scores + features cache
    # the run was device resident: the contigs were scored from the counts in HBM (100 of 101 after the length
    # screen) and the normalised float64 matrix was never brought to the host ...
    assert scorer._batch is not None and scorer._batch.n == 100 and scorer._rows is None
    # ... until somebody reads the reference's attribute
    assert scorer.data_points.shape == (100, 256) and np.array_equal(scorer.data_points, g["q"])
    got = fileIO.read_phamer_output(str(out))
    assert sorted(got, key=int) == [str(c) for c in range(100)]
    scores = np.array([got[str(c)] for c in range(100)])
    if np.allclose(scorer.positive_centroids, g["cpos_eq"], rtol=0, atol=1e-15):
        assert helpers.rel_err(scores, g["combo_eq"]) < 1e-6
    else:   # another scikit-learn build on this box: compare with the oracle on this run's centroids
        from oracle import oracle
        want = oracle.score_points(g["q"], scorer.positive_data, scorer.negative_data, "combo", 3,
                                   scorer.positive_centroids, scorer.negative_centroids)
        assert helpers.rel_err(scores, want) < 1e-6
    assert np.array_equal(np.sign(scores), np.sign(g["combo_eq"]))
    # second run picks up the features cache (counting skipped) and gives the same file
    first = out.read_text()
    warm = phamer.main(["-in", str(indir), "-data", str(tmp_path / "data"), "--equalize_reference", "-l", "0"])
    assert fileIO.read_phamer_output(str(out)).keys() >= got.keys()
    # the cached counts went up as integers (phk_batch_from_counts): the same resident matrix as the first run's, so the
    # same scores bit for bit, and again no float matrix on the host
    assert warm._batch is not None and warm._batch.n == 101 and warm._rows is None
    assert np.array_equal(warm.scores[:100], scorer.scores)
    again = fileIO.read_phamer_output(str(out))
    assert helpers.rel_err(np.array([again[str(c)] for c in range(100)]), scores) < 1e-12
    assert first.count("\n") == out.read_text().count("\n") - 1    # the 12-base contig is scored without the screen
    # a features file that is not a k-mer count matrix (a negative entry) keeps the reference's float rows
    from phamers_amd import _lib
    bad = np.array(ref["pos_counts"][:4], dtype=np.int64)
    bad[1, 7] = -3
    assert _lib.Batch.from_counts(_lib.get_context(), bad) is None
    assert _lib.Batch.from_counts(_lib.get_context(), np.ones((3, 100), dtype=np.int64)) is None
    b = _lib.Batch.from_counts(_lib.get_context(), ref["pos_counts"][:50])
    assert np.array_equal(b.counts(), ref["pos_counts"][:50]) and b.total_bases == 0
    b.close()


def test_native_feature_file_reader_equals_loadtxt_and_steps_aside(tmp_path, monkeypatch):
    """read_feature_file through the native reader == np.loadtxt's result on files save_counts writes (several formatting
    chunks, header block, CRLF, no trailing newline); files of any other shape are left to np.loadtxt."""
    from phamers_amd import fileIO
    rng = np.random.default_rng(5)
    n = 3 * 4096 + 5
    counts = rng.integers(0, 5000, size=(n, 16))
    ids = np.array(["NODE_%d_length_%d" % (i, 5000 + i % 977) for i in range(n)])
    path = str(tmp_path / "features.csv")
    fileIO.save_counts(counts, ids, path, header="K-mer count file\nsecond header line")
    got_ids, got = fileIO.read_feature_file(path)
    want = np.atleast_2d(np.loadtxt(path, delimiter=",", dtype=str))
    assert got.dtype == np.int64 and np.array_equal(got, want[:, 1:].astype(int)) and np.array_equal(got, counts)
    assert got_ids.dtype.kind == "U" and list(got_ids) == list(want[:, 0]) == list(ids)
    assert fileIO._read_feature_file_native(path) is not None
    # one row, CRLF line ends, no newline at the end
    one = str(tmp_path / "one.csv")
    open(one, "wb").write(b"# h\r\nabc,1,2,3")
    i1, f1 = fileIO.read_feature_file(one)
    assert list(i1) == ["abc"] and f1.tolist() == [[1, 2, 3]]
    # shapes the native reader refuses: inline comment, blanks, a float field, ragged rows -> np.loadtxt decides
    for text in ("a,1,2 # note\nb,3,4\n", "a, 1, 2\nb, 3, 4\n", "a,1.0,2\nb,3,4\n"):
        other = str(tmp_path / "other.csv")
        open(other, "w").write(text)
        assert fileIO._read_feature_file_native(other) is None
    open(other, "w").write("a,1,2 # note\nb,3,4\n")
    i2, f2 = fileIO.read_feature_file(other)
    assert f2.tolist() == [[1, 2], [3, 4]]
    open(other, "w").write("a,1,2\nb,3\n")
    assert fileIO._read_feature_file_native(other) is None
    with pytest.raises(ValueError):
        fileIO.read_feature_file(other)
    with pytest.raises((IOError, OSError)):
        fileIO.read_feature_file(str(tmp_path / "missing.csv"))


def test_fasta_line_scanner_simd_and_plain_agree(tmp_path):
    """The parser's line scanner has an AVX2 form (picked when the library is loaded) and a memchr form
    (PHK_FASTA_NO_SIMD=1): a child process parses the same awkward file with the plain one -- lines of every length from 0 to
    200 around the 32-byte step, blanks and CR inside and at the end of lines, no final newline -- and must see what this
    process sees."""
    import hashlib
    import subprocess
    import sys
    from phamers_amd import _lib
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ATGCN", dtype=np.uint8)
    parts = ["; comment before the first record\n"]
    for c in range(400):
        w = c % 201
        n = int(rng.integers(0, 3000))
        seq = alphabet[rng.integers(0, 5, size=n)].tobytes().decode()
        lines = [seq[i:i + max(w, 1)] for i in range(0, n, max(w, 1))] if w else [seq]
        if c % 5 == 1:
            lines = [ln[:len(ln) // 2] + " " + ln[len(ln) // 2:] for ln in lines]
        if c % 5 == 2:
            lines = [ln + "\r" for ln in lines]
        if c % 5 == 3:
            lines = [ln + "  \t" for ln in lines]
        if c % 7 == 0:
            lines.insert(len(lines) // 2, "")
        parts.append(">rec_%d some words\t%d \n" % (c, n) + "\n".join(lines) + "\n")
    text = "".join(parts).rstrip("\n")
    path = tmp_path / "lines.fasta"
    path.write_text(text)
    want_titles, want_seqs = _seqio_like(text)

    def digest(f):
        h = hashlib.sha256()
        for t, s in zip(f.titles(), f.sequences()):
            h.update(t.encode() + b"\0" + s.encode() + b"\1")
        h.update(str(f.lengths().tolist()).encode())
        return h.hexdigest()
    f = _lib.Fasta(str(path), threads=3)
    assert f.titles() == want_titles and f.sequences() == want_seqs
    mine = digest(f)
    f.close()
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "from phamers_amd import _lib\n"
            "f = _lib.Fasta(%r, threads=3)\n"
            "h = hashlib.sha256()\n"
            "[h.update(t.encode() + b'\\0' + s.encode() + b'\\1') for t, s in zip(f.titles(), f.sequences())]\n"
            "h.update(str(f.lengths().tolist()).encode()); print(h.hexdigest())\n") % (
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(path))
    env = dict(os.environ, PHK_FASTA_NO_SIMD="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == mine


@pytest.mark.gpu
def test_fasta_file_straight_into_the_upload_equals_read_then_upload(tmp_path):
    """phk_batch_from_fasta_file (the command line's cold path: sequences parsed straight into the 64 MB staging buffers of
    the upload) against phk_fasta_read + phk_batch_from_fasta on a 150 MB file: records of every size -- some longer than
    a staging buffer, so that they are cut by one or two chunk ends --, lines with blanks and CR inside (a chunk end inside
    such a line), empty records, non-symbols.  Counts, ids, titles and lengths must be identical."""
    from phamers_amd import _lib
    rng = np.random.default_rng(31)
    alphabet = np.frombuffer(b"ATGCN", dtype=np.uint8)
    path = tmp_path / "big.fasta"
    sizes = [70 << 20, 0, 5000, 300, 66 << 20, 1, 12 << 20] + [int(x) for x in rng.integers(0, 20000, size=300)]
    with open(path, "w") as f:
        f.write("text before the first record\n")
        for c, n in enumerate(sizes):
            seq = alphabet[rng.integers(0, 5 if c % 3 == 0 else 4, size=n)].tobytes().decode()
            w = 70 if n > (1 << 20) else int(rng.integers(1, 200))
            f.write(">contig_%d_length_%d_ID_%d some words\n" % (c, n, c))
            if c % 4 == 1 or n > (60 << 20):       # blanks and CR inside / at the end of lines
                body = "\n".join(seq[i:i + w // 2] + " " + seq[i + w // 2:i + w] + "\r" for i in range(0, n, w))
            else:
                body = "\n".join(seq[i:i + w] for i in range(0, n, w))
            f.write(body + ("\n\n" if c % 5 == 0 else "\n"))
    ctx = _lib.get_context()
    fa = _lib.Fasta(str(path))
    want_batch = _lib.Batch.from_fasta(ctx, fa, 4)
    want = want_batch.counts_u32()
    want_batch.close()
    idx, batch = _lib.Fasta.count_file(ctx, str(path), 4)
    try:
        assert batch.n == fa.n_records == len(sizes) and batch.total_bases == fa.total_bases == sum(sizes)
        assert np.array_equal(batch.counts_u32(), want)
        assert idx.titles() == fa.titles() and np.array_equal(idx.lengths(), fa.lengths())
        assert idx.phamers_ids().tolist() == fa.phamers_ids().tolist()
    finally:
        batch.close()
        idx.close()
        fa.close()
    with pytest.raises(IOError):
        _lib.Fasta.count_file(ctx, str(tmp_path / "missing.fasta"), 4)


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [0, 1, 3])
def test_device_delining_of_every_line_layout(tmp_path, threads):
    """phk_batch_from_fasta_file since round 5: the file's raw bytes go to the device and phk_deline_pack_kernel reads base i
    of a record at  begin + (i / lw) (lw + tl) + i % lw; records the index scan finds irregular are de-lined by the host into a
    side buffer.  Every layout the scan has to tell apart, against the host reader + phk_pack_kernel (phk_fasta_read +
    phk_batch_from_fasta): LF and CRLF line ends, trailing blanks / tabs (the same on every line: regular; different: not),
    a shorter last line, a last line without line end, one-line records, a blank line before / inside / after the sequence,
    lines of varying width, a wider line after a narrow one, ' ' and CR inside a line, '>' inside a line, title-only and
    empty records, a record whose only line is longer than any other -- with 1, 3 and all scan threads (slices cut records at
    other places)."""
    from phamers_amd import _lib, synth
    rng = np.random.default_rng(5)
    recs = []

    def seq(n):
        return "".join("ATGCNatgcRY"[i] for i in rng.integers(0, 11 if n % 3 == 0 else 4, size=n))

    def lines(s, w, end="\n", last_end=None):
        parts = [s[i:i + w] for i in range(0, len(s), w)] or [""]
        return end.join(parts) + (end if last_end is None else last_end)
    recs.append(lines(seq(1000), 70))                                   # the usual record
    recs.append(lines(seq(1400), 70, "\r\n"))                           # CRLF throughout
    recs.append(lines(seq(700), 60, " \t\n"))                           # the same trailing blanks on every line
    recs.append(lines(seq(350), 70))                                    # last line full
    recs.append(seq(5000) + "\n")                                       # one line
    recs.append("\n" + lines(seq(300), 50))                             # blank line before the sequence
    recs.append(lines(seq(200), 50) + "\n" + lines(seq(120), 50))       # blank line inside
    recs.append(lines(seq(260), 50) + "\n\n")                           # blank lines after
    recs.append(seq(40) + "\n" + seq(90) + "\n" + seq(40) + "\n")       # a wider line after a narrow one
    recs.append(seq(90) + "\n" + seq(40) + "\n" + seq(90) + "\n")       # bases after a short line
    s = seq(210)
    recs.append(s[:30] + " " + s[30:70] + "\n" + s[70:140] + "\r\n" + s[140:] + "\n")   # blank inside, mixed ends
    recs.append(seq(70) + "\n" + seq(35) + ">" + seq(34) + "\n" + seq(10) + "\n")       # '>' inside a line
    recs.append("")                                                     # title only
    recs.append("\n\n")                                                 # blank lines only
    recs.append(lines(seq(3), 70))
    recs.append(lines(seq(211), 1))                                     # one base per line
    recs.append(lines(seq(500), 70, "\n", last_end="\r\n"))             # only the last line ends differently
    recs.append(lines(seq(140), 70, "\r\n", last_end="\n"))
    recs.append(lines(seq(141), 70, "\n", last_end="  \n"))
    for _ in range(400):                                                # bulk: many records, so that several scan threads have work
        recs.append(lines(seq(int(rng.integers(0, 12000))), int(rng.integers(20, 120)), "\n" if rng.random() < 0.8 else "\r\n"))
    recs.append(lines(seq(333), 80, "\n", last_end=""))                 # the file ends without a line end
    path = tmp_path / "layouts.fasta"
    with open(path, "w", newline="") as f:
        f.write("junk before the first record\nmore junk\n")
        for c, body in enumerate(recs):
            f.write(">rec_%d_ID_%d words\n" % (c, c))
            f.write(body)
    assert os.path.getsize(path) > (2 << 20)                            # (> 1 MB per scan thread: the slices really are cut)
    ctx = _lib.get_context()
    fa = _lib.Fasta(str(path), threads=threads)
    want_batch = _lib.Batch.from_fasta(ctx, fa, 4)
    want = want_batch.counts_u32()
    want_batch.close()
    assert fa.n_records == len(recs)
    idx, batch = _lib.Fasta.count_file(ctx, str(path), 4, threads=threads)
    try:
        assert batch.n == fa.n_records and batch.total_bases == fa.total_bases
        assert np.array_equal(idx.lengths(), fa.lengths())
        got = batch.counts_u32()
        bad = np.flatnonzero((got != want).any(axis=1))
        assert bad.size == 0, (bad[:10], [repr(recs[int(i)][:80]) for i in bad[:3]])
        assert idx.titles() == fa.titles()
        # ... and one rank's share of the same file, for 3 ranks (phk_batch_from_fasta_part)
        rows = []
        for part in range(3):
            pidx, pbatch = _lib.Fasta.count_file(ctx, str(path), 4, threads=threads, part=(part, 3))
            rows.append(pbatch.counts_u32() if pbatch.n else np.zeros((0, 256), np.uint32))
            pbatch.close()
            pidx.close()
        assert np.array_equal(np.concatenate(rows), want)
    finally:
        batch.close()
        idx.close()
        fa.close()


def _bgzf(data, block=65280, eof=True, level=6):
    """`data` as a BGZF stream (bgzip's format: gzip members of at most 64 KiB, each with its compressed size in a 'BC'
    extra field), optionally with bgzip's empty end-of-file block."""
    import struct
    import zlib
    out = bytearray()
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof else [])
    for chunk in chunks:
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = c.compress(chunk) + c.flush()
        out += struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, len(comp) + 25)
        out += comp + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


def test_gzip_shapes_one_member_many_members_and_bgzf(tmp_path):
    """".gz" input (scripts/kmer.py:131-134): a one-member gzip file, a multi-member one (`cat a.gz b.gz`, members cut at
    awkward places), BGZF with and without its end-of-file block and with tiny blocks, a ".gz" that is not compressed at all
    (gzopen reads it as it is) -- all give the plain file's records; byte ranges and parts of the BGZF stream, which inflate
    only the blocks they need, give every record to exactly one range, as the plain file's do; a damaged member is an
    IOError, not a short read."""
    from phamers_amd import _lib, synth
    text = "junk\n"
    for c in range(300):
        s = synth.synth_contig(9, c, 200 + 913 * (c % 17), invalid_ppm=5000)
        w = 60 + c % 11
        text += ">" + synth.contig_header(c, len(s)) + " d%d\n" % c + "\n".join(s[i:i + w] for i in range(0, len(s), w)) + "\n"
    raw = text.encode()
    assert len(raw) > (1 << 20)
    plain = tmp_path / "x.fasta"
    plain.write_bytes(raw)

    def read(path, **kw):
        f = _lib.Fasta(str(path), **kw)
        out = (f.titles(), f.sequences())
        f.close()
        return out

    want = read(plain)
    assert len(want[0]) == 300
    shapes = {"one": gzip.compress(raw),
              "many": b"".join(gzip.compress(raw[a:b]) for a, b in ((0, 7), (7, 70001), (70001, 70002), (70002, len(raw)))),
              "bgzf": _bgzf(raw), "bgzf_noeof": _bgzf(raw, eof=False), "bgzf_tiny": _bgzf(raw, block=997, level=1),
              "stored": raw}
    for name, blob in shapes.items():
        p = tmp_path / (name + ".fasta.gz")
        p.write_bytes(blob)
        assert read(p) == want, name
        assert read(p, threads=3) == want, name
        for n in (2, 3, 7):
            parts = [read(p, part=(i, n)) for i in range(n)]
            assert sum((x[0] for x in parts), []) == want[0], (name, n)
            assert sum((x[1] for x in parts), []) == want[1], (name, n)
            if name.startswith("bgzf"):    # the same cuts as the plain file's: the fractions are of the uncompressed stream
                assert parts == [read(plain, part=(i, n)) for i in range(n)], (name, n)
        for cut in (0, 1, 5, 65279, 65280, 65281, 400000, len(raw) - 1, len(raw), len(raw) + 9):
            a, b = read(p, byte_range=(0, cut)), read(p, byte_range=(cut, None))
            assert a[0] + b[0] == want[0] and a[1] + b[1] == want[1], (name, cut)
    # damage: a flipped byte in the middle of a member, a truncated file
    for name in ("one", "bgzf"):
        blob = bytearray(shapes[name])
        blob[len(blob) // 2] ^= 0x5A
        p = tmp_path / ("bad_" + name + ".fasta.gz")
        p.write_bytes(bytes(blob))
        with pytest.raises(IOError):
            read(p)
        p.write_bytes(shapes[name][: len(shapes[name]) // 2])
        with pytest.raises(IOError):
            read(p)
    empty = tmp_path / "empty.fasta.gz"
    empty.write_bytes(b"")
    assert read(empty) == ([], [])
    empty.write_bytes(gzip.compress(b""))
    assert read(empty) == ([], [])
