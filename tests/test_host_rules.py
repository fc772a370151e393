"""Host-side rules either side of the path, pinned to reference-generated fixtures (tools/gen_golden.py):
FASTA header -> id (tests/golden/ids.json, scripts/id_parser.py:18-100) through the native scanner and the oracle,
and the transform_kmers index tables (tests/golden/transform.npz, scripts/transform_kmers.py:21-88).  CPU only:
the shared library is loaded, no device is touched."""
import os
import random

import numpy as np
import pytest

from tests import helpers


def _answer(fn, header):
    try:
        return fn(header), None
    except Exception as e:   # noqa: BLE001 -- the exception type is part of the contract
        return None, type(e).__name__


def test_id_rules_match_the_reference_fixture():
    from oracle import oracle
    from phamers_amd import id_parser
    cases = helpers.load_json("ids.json")["cases"]
    assert len(cases) >= 30
    for c in cases:
        want = (c["id"], c["error"])
        assert _answer(oracle.get_id, c["header"]) == want, c
        assert _answer(id_parser.get_id, c["header"]) == want, c


def test_native_id_scanner_agrees_with_the_oracle_on_random_headers():
    from oracle import oracle
    from phamers_amd import id_parser
    rnd = random.Random(5)
    pieces = ["_ID_", "ID", "_", "|", ">", " ", "\t", ".", "1", "23", "NC", "x", "-circular", "e5", "inf", "nan", "0x1p3",
              "SuperContig", "length", "gi", "ref", "A.1", "-", "+"]
    for _ in range(4000):
        h = "".join(rnd.choice(pieces) for _ in range(rnd.randint(0, 9)))
        assert _answer(id_parser.get_id, h) == _answer(oracle.get_id, h), repr(h)


def test_fasta_reader_ids(tmp_path):
    """The reader applies the id rules to record.id (first word of the title) on its worker threads; a header
    with none of the shapes raises IndexError like the reference's get_fasta_ids."""
    from phamers_amd import _lib
    good = [c for c in helpers.load_json("ids.json")["cases"] if c["error"] is None and c["header"].strip()
            and not any(ch.isspace() for ch in c["header"].strip())]
    path = tmp_path / "ids.fa"
    with open(path, "w") as f:
        for i, c in enumerate(good):
            f.write(">%s some description %d\nATGCATGC\n" % (c["header"].strip().lstrip(">"), i))
    fa = _lib.Fasta(str(path), threads=3)
    from oracle import oracle
    want = [oracle.get_id(c["header"].strip().lstrip(">")) for c in good]
    assert list(fa.phamers_ids()) == want
    fa.close()
    bad = tmp_path / "bad.fa"
    bad.write_text(">SuperContig_1_ID_1\nATGC\n>plain_contig_name\nATGC\n")
    fb = _lib.Fasta(str(bad))
    with pytest.raises(IndexError):
        fb.phamers_ids()
    fb.close()


def test_transform_tables_match_the_reference_fixture():
    from oracle import oracle
    from phamers_amd import transform_kmers as tk
    z = helpers.load_npz("transform.npz")
    for k in (2, 3, 4):
        for name, (rev, comp) in (("rev", (True, False)), ("comp", (False, True)), ("revcomp", (True, True))):
            want = z["%s_idx_k%d" % (name, k)]
            assert np.array_equal(tk.reference_indices(k, rev, comp), want), (name, k)
            assert np.array_equal(oracle.reference_transform_indices(k, rev, comp), want), (name, k)
            assert np.array_equal(oracle.transform_kmers(z["in_k%d" % k], rev, comp), z["%s_k%d" % (name, k)])
            perm = tk.exact_indices(k, rev, comp)
            assert sorted(perm) == list(range(4 ** k))
    # the reference's tables are not permutations (the documented deviation of exact=True)
    assert len(set(z["revcomp_idx_k4"])) == 64
    with pytest.raises(IndexError):
        oracle.transform_kmers(np.zeros((1, 4 ** 5), dtype=np.int64), True, True)


def test_command_line_accepts_the_reference_flag_set():
    """scripts/phamer.py:515-553: every option of the reference's command line parses (so run.sh / submit/*.sh style
    invocations do not die in argparse); -equal is the reference's short spelling of --equalize_reference; t-SNE and
    plotting are refused explicitly, before any work."""
    from phamers_amd import phamer
    a = phamer._parser().parse_args(["-in", "in_dir", "-data", "data", "-equal", "-m", "knn", "-eps", "3.5", "-mp", "4",
                                     "-pxty", "12", "-id", ".fa", "-k", "4", "-l", "5000", "-p", "p.fa", "-n", "neg_dir",
                                     "-tsne", "t.csv", "--debug"])
    assert a.equalize_reference and a.method == "knn" and a.eps == 3.5 and a.minPts == 4 and a.debug
    assert a.input_directory == "in_dir" and a.data_directory == "data" and a.file_identifier == ".fa"
    assert phamer._parser().parse_args(["-in", "x", "-e"]).equalize_reference          # the earlier spelling still works
    assert not phamer._parser().parse_args(["-in", "x"]).equalize_reference
    for flag in ("-do_tsne", "-plot"):
        with pytest.raises(NotImplementedError):
            phamer.main(["-in", "x", "-data", "y", flag])


def test_failed_features_cache_write_is_an_error_of_the_run(tmp_path):
    """The features cache is written on a thread beside the scoring; a write that fails (here: the directory does not
    exist) must surface from finish_io() as it would from the reference's sequential save_counts
    (scripts/phamer.py:132-134), and leave no partial file."""
    from phamers_amd import phamer
    sc = phamer.phamer_scorer()
    counts = np.arange(12, dtype=np.uint32).reshape(3, 4)
    ids = np.array(["a", "b", "c"])
    bad = str(tmp_path / "no_such_dir" / "x_features.csv")
    sc._write_cache_async(counts, ids, bad)
    with pytest.raises(Exception) as e:
        sc.finish_io()
    assert not isinstance(e.value, AssertionError)
    assert sc._pending_io == []
    good = str(tmp_path / "x_features.csv")
    sc._write_cache_async(counts, ids, good)
    sc.finish_io()
    assert os.path.exists(good) and not os.path.exists(good + ".part")
    sc.finish_io()   # nothing pending: a no-op


def test_kmer_command_line_parses_like_the_reference():
    """scripts/kmer.py:283-303: positional input / output, -k, -s, -sym, -id, -v, --debug, same defaults (the Namespace is
    what save_counts stamps into the features header: tests/golden/files.json holds one)."""
    from phamers_amd import kmer
    doc = helpers.load_json("files.json")["args"]
    a = kmer._parser().parse_args([doc["input_file"], doc["output_file"], "-k", str(doc["kmer_length"])] + (["-v"] if doc["verbose"] else []))
    for key in ("input_file", "output_file", "kmer_length", "symbols", "verbose", "debug", "sample", "file_identifier"):
        assert getattr(a, key) == doc[key], key
    b = kmer._parser().parse_args(["d", "o.csv", "-k", "5", "-s", "10", "-sym", "AUGC", "-id", ".fa", "--debug"])
    assert (b.kmer_length, b.sample, b.symbols, b.file_identifier, b.debug) == (5, 10, "AUGC", ".fa", True)
    with pytest.raises(SystemExit):
        kmer.main([str("/nonexistent/path/x.fasta"), "out.csv"])


def test_row_division_by_a_shared_reciprocal_is_the_ieee_quotient():
    """phk_div_row (csrc/phk_common.h) -- x / T from RN(1 / T) and two fused Newton steps on the quotient -- replayed in
    exact rational arithmetic: equal to the IEEE quotient for every (count, row sum) tried.  The device side of the same
    claim is the normalise kernel's bit-exact test against NumPy (tests/test_gpu_count.py)."""
    import random
    from fractions import Fraction

    def fma(a, b, c):
        return float(Fraction(a) * Fraction(b) + Fraction(c))

    def div_row(x, T, y):
        q0 = x * y
        q1 = fma(fma(-q0, T, x), y, q0)
        return fma(fma(-q1, T, x), y, q1)

    rng = random.Random(7)
    cases = [(x, T) for T in list(range(1, 120)) + [4996, 4997, 9995, 9996, 2 ** 16 - 1, 2 ** 24 + 1, 2 ** 32 - 1]
             for x in {0, 1, T // 3, T // 2, max(T - 1, 0), T}]
    cases += [(rng.randrange(0, T + 1), T) for T in (rng.randrange(1, 2 ** 32) for _ in range(4000))]
    cases += [(rng.randrange(0, min(T, 400) + 1), T) for T in (rng.randrange(1, 200000) for _ in range(4000))]
    for x, T in cases:
        xf, Tf = float(x), float(T)
        assert div_row(xf, Tf, 1.0 / Tf) == xf / Tf, (x, T)


def test_kmeans_plusplus_seeds_equal_scikit_learn():
    """learning.kmeans_plusplus_seeds, the NumPy statement of scikit-learn's k-means++ seeding that the product's k-means route
    runs on the host (scripts/learning.py:138 -> KMeans.fit -> _kmeans_plusplus): the same indices and centres as
    sklearn.cluster.kmeans_plusplus from the same RandomState, on the mean-centred reference matrices (the data the golden
    centroids were fitted on), on a fold-sized subset and on random matrices of other shapes and seeds."""
    from sklearn.cluster import kmeans_plusplus
    from oracle import oracle
    from phamers_amd import learning
    from tests import helpers
    ref = helpers.load_npz("ref_features.npz")
    pos = oracle.normalize_counts(ref["pos_counts"].astype(np.int64))
    neg = oracle.normalize_counts(ref["neg_counts"].astype(np.int64))
    rng = np.random.default_rng(3)
    cases = [(pos[:2255], 86, 10), (neg[:2255], 86, 10), (pos, 86, 10), (neg, 86, 10), (rng.random((700, 64)), 12, 3),
             (rng.standard_normal((3000, 40)), 50, 7), (pos[rng.permutation(len(pos))[:1800]], 86, 10), (rng.random((9, 5)), 9, 1)]
    for X, k, seed in cases:
        Xc = np.array(X, dtype=np.float64, order="C")
        Xc -= Xc.mean(axis=0)
        got_c, got_i = learning.kmeans_plusplus_seeds(Xc, k, np.random.RandomState(seed))
        want_c, want_i = kmeans_plusplus(Xc, k, random_state=np.random.RandomState(seed))
        assert np.array_equal(got_i, want_i), (X.shape, k, seed)
        assert np.array_equal(got_c, want_c), (X.shape, k, seed)
