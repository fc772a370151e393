"""GPU parity: counting / packing / normalising through the C ABI (host facade and device
API) against the golden vectors and the CPU oracle.  Bit-exact."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kmer():
    from phamers_amd import kmer
    return kmer


@pytest.fixture(scope="module")
def ctx():
    from phamers_amd import _lib
    return _lib.get_context()


def test_count_string_golden(kmer):
    golden = helpers.load_npz("counts.npz")
    seqs, doc = helpers.count_cases()
    n = 0
    for key, want in golden.items():
        name, _, ktag = key.rpartition("__k")
        if name not in seqs:
            continue
        got = kmer.count_string(seqs[name], int(ktag))
        assert got.dtype == np.int64 and got.shape == want.shape, key
        assert np.array_equal(got, want), key
        n += 1
    assert n > 100


def test_count_list_shapes_and_soft_failure(kmer):
    golden = helpers.load_npz("counts.npz")
    seqs, doc = helpers.count_cases()
    lst = [seqs[n] for n in doc["list5"]]
    got = kmer.count(lst, 4)
    assert got.shape == (5, 256) and np.array_equal(got, golden["list5__k4"])
    one = kmer.count([lst[0]], 4)
    assert one.shape == (256,) and np.array_equal(one, golden["list1__k4"])
    assert kmer.count(12345, 4) is None                       # scripts/kmer.py:108-110
    a = kmer.count_string(lst[0], 4, normalize=True)
    assert np.array_equal(a, golden["norm_mixed_invalid__k4"])
    z = kmer.count_string("N" * 40, 4, normalize=True)
    assert np.array_equal(z, golden["norm_all_N__k4"])
    with pytest.raises(NotImplementedError):
        kmer.count_string("ATGC", 2, symbols="RHKDESTNQCUGPAVILMFYW")
    from phamers_amd import _lib
    with pytest.raises(_lib.PhkError):
        kmer.count_string("ATGC", 9)


def test_count_other_alphabet_vs_oracle(kmer):
    from oracle import oracle
    s = "AUGCAUGGCCAUUAGCNAUGCaugc" * 7
    for k in (2, 4):
        assert np.array_equal(kmer.count_string(s, k, symbols="AUGC"), oracle.count_string(s, k, symbols="AUGC"))


def test_count_ragged_batch_vs_oracle(kmer):
    from oracle import oracle
    from phamers_amd import synth
    rng = np.random.default_rng(5)
    lens = [0, 1, 3, 4, 5, 15, 16, 17, 31, 32, 33, 63, 64, 65, 1023, 1024, 1025, 4999, 5000, 5001] + \
        [int(x) for x in rng.integers(0, 3000, 40)]
    seqs = [synth.synth_contig(9, i, L, invalid_ppm=(0 if i % 3 else 30000)) for i, L in enumerate(lens)]
    for k in (1, 3, 4, 5, 6, 7):
        got = kmer.count(seqs, k)
        want = oracle.count(seqs, k)
        assert np.array_equal(got, want), k


def test_normalize_bit_exact(kmer):
    g = helpers.load_npz("normalize.npz")
    got = kmer.normalize_counts(g["in2d"])
    assert got.dtype == np.float64
    assert np.array_equal(np.isnan(got), np.isnan(g["out2d"]))
    ok = ~np.isnan(got)
    assert np.array_equal(got[ok].view(np.uint64), g["out2d"][ok].view(np.uint64))
    assert np.isnan(got[3]).all()
    got1 = kmer.normalize_counts(g["in1d"])
    assert got1.shape == g["out1d"].shape and np.array_equal(got1, g["out1d"])
    # float input (renormalising frequencies): bit-exact with NumPy's own row division
    f = g["out2d"][[0, 1, 2, 4]] * 3.7
    want = f.copy()
    for i in range(want.shape[0]):
        want[i, :] /= np.sum(want[i, :])
    assert np.array_equal(kmer.normalize_counts(f).view(np.uint64), want.view(np.uint64))


def test_device_pack_and_count_match_host_statement(ctx):
    """Device packer vs the NumPy statement of the packed format; device counter on the packed
    stream (with and without mask) vs the oracle; nwin = row sums."""
    from oracle import oracle
    from phamers_amd import device, synth
    lens = [5000, 37, 0, 5000, 4, 3, 129, 70000, 5000]
    seqs = [synth.synth_contig(4, i, L, invalid_ppm=(20000 if i in (1, 3, 7) else 0)) for i, L in enumerate(lens)]
    T = sum(lens)
    offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    codes = np.concatenate([oracle.sequence_to_codes(s) for s in seqs])
    want_packed, want_mask = helpers.pack_codes(codes)
    raw = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    d_raw = device.DeviceArray.from_host(ctx, raw)
    d_packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
    d_mask = device.DeviceArray(ctx, device.mask_words(T), np.uint32)
    d_flag = device.DeviceArray(ctx, 1, np.uint32)
    device.pack_ascii(ctx, d_raw, T, d_packed, d_mask, d_flag)
    assert np.array_equal(d_packed.to_host(), want_packed)
    assert np.array_equal(d_mask.to_host(), want_mask)
    assert d_flag.to_host()[0] != 0
    d_off = device.DeviceArray.from_host(ctx, offsets)
    for k in (4, 5, 6):
        D = 4 ** k
        d_counts = device.DeviceArray(ctx, (len(lens), D), np.uint32)
        d_nwin = device.DeviceArray(ctx, len(lens), np.uint32)
        device.count(ctx, d_packed, d_mask, T, d_off, len(lens), k, d_counts, d_nwin)
        got = d_counts.to_host()
        want = oracle.count(seqs, k)
        assert np.array_equal(got.astype(np.int64), want), k
        assert np.array_equal(d_nwin.to_host().astype(np.int64), want.sum(axis=1))


def test_device_synth_matches_host_generator(ctx):
    from oracle import oracle
    from phamers_amd import device, synth
    for (n, L, ppm) in ((7, 5000, 0), (5, 333, 50000), (3, 32, 0), (4, 1, 0)):
        T = n * L
        d_packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
        d_mask = device.DeviceArray(ctx, device.mask_words(T), np.uint32)
        d_off = device.DeviceArray(ctx, n + 1, np.uint64)
        device.synth_packed(ctx, 3, 11, n, L, d_packed, d_off, d_mask, ppm)
        codes = np.concatenate([synth.synth_codes(3, 11 + c, L, ppm) for c in range(n)])
        want_packed, want_mask = helpers.pack_codes(codes)
        assert np.array_equal(d_packed.to_host(), want_packed), (n, L, ppm)
        assert np.array_equal(d_mask.to_host(), want_mask), (n, L, ppm)
        assert np.array_equal(d_off.to_host(), np.arange(n + 1, dtype=np.uint64) * L)


def test_transform_kmers_identity(kmer):
    """exact=True: counts(reverse / complement / reverse-complement of s) == transform_kmers(counts(s))."""
    from phamers_amd import synth, transform_kmers
    comp = str.maketrans("ATGC", "TACG")
    for k in (2, 3, 4, 5):
        seqs = [synth.synth_contig(4, i, 600 + 50 * i) for i in range(5)]
        c = kmer.count(seqs, k)
        rev = kmer.count([s[::-1] for s in seqs], k)
        cmp_ = kmer.count([s.translate(comp) for s in seqs], k)
        rc = kmer.count([s.translate(comp)[::-1] for s in seqs], k)
        assert np.array_equal(transform_kmers.transform_kmers(c, reverse=True, complement=False, exact=True), rev)
        assert np.array_equal(transform_kmers.transform_kmers(c, reverse=False, complement=True, exact=True), cmp_)
        assert np.array_equal(transform_kmers.transform_kmers(c, reverse=True, complement=True, exact=True), rc)
        assert transform_kmers.transform_kmers(c, reverse=False, complement=False) is c


@pytest.mark.gpu
def test_transform_kmers_matches_the_reference_outputs():
    """Default mode = the reference's own index tables (tests/golden/transform.npz: outputs of
    scripts/transform_kmers.py:68-88 run with Python 2 division), including its IndexError for k >= 5."""
    from phamers_amd import transform_kmers
    z = helpers.load_npz("transform.npz")
    errors = helpers.load_json("transform.json")["errors_by_k"]
    for k in (2, 3, 4):
        x = z["in_k%d" % k]
        assert np.array_equal(transform_kmers.transform_kmers(x, reverse=True, complement=False), z["rev_k%d" % k])
        assert np.array_equal(transform_kmers.transform_kmers(x, reverse=False, complement=True), z["comp_k%d" % k])
        assert np.array_equal(transform_kmers.transform_kmers(x, reverse=True, complement=True), z["revcomp_k%d" % k])
        assert np.array_equal(transform_kmers.transform_kmers(x, reverse=False, complement=False), z["none_k%d" % k])
    for k in (5, 6):
        assert errors[str(k)] == "IndexError"
        with pytest.raises(IndexError):
            transform_kmers.transform_kmers(np.zeros((2, 4 ** k), dtype=np.int64), reverse=True, complement=True)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [3, 4, 5])
def test_slot_count_kernel_ragged_and_handover(ctx, k):
    """The slot kernel (32 contigs per workgroup at k = 3 / 4, 16 at k = 5; all-valid input): ragged lengths around every
    edge (empty, shorter than k, exactly k, 63 / 64 / 65 / 127 / 128 / 129 windows per lane pair, the last
    contig ending on the last word of the stream), a contig count that is not a multiple of 32, and
    contigs far above the batch mean, which it must hand to the wave-per-contig kernel -- bit-exact
    against the oracle, with nwin = row sums; the same batch through the wave-per-contig kernel alone
    (option count_lanes=0) gives identical rows."""
    import os
    from oracle import oracle
    from phamers_amd import device, synth
    rng = np.random.default_rng(3)
    lens = [0, 1, k - 1, k, k + 1, 63 + k - 1, 64 + k - 1, 65 + k - 1, 126 + k, 127 + k, 128 + k, 129 + k, 255, 256, 257,
            1000, 4999, 5000, 5001, 40000, 90000]
    lens += [int(x) for x in rng.integers(0, 3000, 150)]
    lens += [16 * 7 + 5, 16 * 3]            # the stream ends word-aligned
    seqs = [synth.synth_contig(9, i, L) for i, L in enumerate(lens)]
    T = sum(lens)
    offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    raw = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    d_raw = device.DeviceArray.from_host(ctx, raw)
    d_packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
    d_mask = device.DeviceArray(ctx, device.mask_words(T), np.uint32)
    device.pack_ascii(ctx, d_raw, T, d_packed, d_mask, None)
    d_off = device.DeviceArray.from_host(ctx, offsets)
    want = oracle.count(seqs, k)
    D = 4 ** k
    rows = {}
    # slot kernel forced / chosen by the batch statistics / wave-per-contig kernel only; at k = 4 also the two-windows-per-add
    # kernel (5-mer pairs in 16-bit half bins, marginalised at the flush), forced and by the statistics, in both shapes
    # ("" = the library's own choice: those kernels where they apply; k = 5: "d" / "D" = the unstaged slot kernel, 512 / 1024 threads)
    for lanes in ("0", "", "f", "d") + (("q", "Q", "p", "P") if k == 4 else ()) + (("D",) if k == 5 else ()):
        ctx.set_option("count_lanes", lanes)
        try:
            d_counts = device.DeviceArray.from_host(ctx, np.full((len(lens), D), 0xABCD, np.uint32))
            d_nwin = device.DeviceArray.from_host(ctx, np.full(len(lens), 0xABCD, np.uint32))
            device.count(ctx, d_packed, None, T, d_off, len(lens), k, d_counts, d_nwin)
            rows[lanes] = d_counts.to_host()
            assert np.array_equal(rows[lanes].astype(np.int64), want), (k, lanes)
            assert np.array_equal(d_nwin.to_host().astype(np.int64), want.sum(axis=1)), (k, lanes)
        finally:
            ctx.set_option("count_lanes", "")
    for lanes in rows:
        assert np.array_equal(rows["0"], rows[lanes]), lanes
    # the same with invalid bases (validity mask): scattered single characters and long runs
    seqs_n = [synth.synth_contig(9, i, L, invalid_ppm=(30000 if i % 3 else 0)) for i, L in enumerate(lens)]
    seqs_n[16] = seqs_n[16][:300] + "N" * 200 + seqs_n[16][500:]
    seqs_n[18] = "N" * 70 + seqs_n[18][70:4000] + "n" * 5 + seqs_n[18][4005:]
    raw_n = np.frombuffer("".join(seqs_n).encode(), dtype=np.uint8)
    assert len(raw_n) == T
    d_raw_n = device.DeviceArray.from_host(ctx, raw_n)
    d_flag = device.DeviceArray(ctx, 1, np.uint32)
    device.pack_ascii(ctx, d_raw_n, T, d_packed, d_mask, d_flag)
    assert d_flag.to_host()[0] != 0
    want_n = oracle.count(seqs_n, k)
    for lanes in ("0", "", "f", "d"):   # (masked: the wave-per-contig kernel, the slot kernel by the statistics / forced / 512 threads)
        ctx.set_option("count_lanes", lanes)
        try:
            d_counts = device.DeviceArray.from_host(ctx, np.full((len(lens), D), 0xABCD, np.uint32))
            d_nwin = device.DeviceArray.from_host(ctx, np.full(len(lens), 0xABCD, np.uint32))
            device.count(ctx, d_packed, d_mask, T, d_off, len(lens), k, d_counts, d_nwin)
            assert np.array_equal(d_counts.to_host().astype(np.int64), want_n), (k, lanes, "masked")
            assert np.array_equal(d_nwin.to_host().astype(np.int64), want_n.sum(axis=1)), (k, lanes, "masked")
        finally:
            ctx.set_option("count_lanes", "")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_slot_count_kernel_random_batches(ctx, seed):
    """Randomised batches through the slot kernel (forced) and through the automatic choice: heavy-tailed
    lengths in arbitrary order, random k in {3, 4, 5}, with and without invalid characters, batch sizes that
    are not multiples of the workgroup's 32 / 16 contigs -- counts and window totals equal the oracle's."""
    import os
    from oracle import oracle
    from phamers_amd import device, synth
    rng = np.random.default_rng(100 + seed)
    k = int(rng.choice([3, 4, 5]))
    n = int(rng.integers(1, 140))
    lens = np.minimum((rng.pareto(1.2, n) * 300).astype(np.int64), 60000)
    lens[rng.integers(0, n, max(1, n // 10))] = 0
    masked = bool(seed % 2)
    seqs = [synth.synth_contig(50 + seed, i, int(L), invalid_ppm=(20000 if masked and i % 2 else 0)) for i, L in enumerate(lens)]
    T = int(lens.sum())
    if T < 2048:   # the slot kernel needs a stream of at least 64 words
        seqs.append(synth.synth_contig(50 + seed, n, 4096))
        lens = np.append(lens, 4096)
        T = int(lens.sum())
    offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    raw = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    d_raw = device.DeviceArray.from_host(ctx, raw)
    d_packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
    d_mask = device.DeviceArray(ctx, device.mask_words(T), np.uint32)
    d_flag = device.DeviceArray(ctx, 1, np.uint32)
    device.pack_ascii(ctx, d_raw, T, d_packed, d_mask, d_flag)
    d_off = device.DeviceArray.from_host(ctx, offsets)
    want = oracle.count(seqs, k)
    D = 4 ** k
    for lanes in ("0", "q", "Q", "p", "d", "f", ""):   # (q / Q / p: the two-windows-per-add kernel where it applies -- k = 4, no mask; d: k = 5)
        ctx.set_option("count_lanes", lanes)
        try:
            d_counts = device.DeviceArray.from_host(ctx, np.full((len(lens), D), 7, np.uint32))
            d_nwin = device.DeviceArray.from_host(ctx, np.full(len(lens), 7, np.uint32))
            device.count(ctx, d_packed, d_mask if masked else None, T, d_off, len(lens), k, d_counts, d_nwin)
            assert np.array_equal(d_counts.to_host().astype(np.int64), want), (seed, k, lanes)
            assert np.array_equal(d_nwin.to_host().astype(np.int64), want.sum(axis=1)), (seed, k, lanes)
        finally:
            ctx.set_option("count_lanes", "")


@pytest.mark.gpu
def test_two_windows_per_add_count_kernel_k4(ctx):
    """phk_count_pairs_kernel (k = 4, no mask): windows taken in pairs from the contig's first base, one add per pair into
    stride-2 5-mer bins (16-bit halves), 4-mer counts = the two marginals + the unpaired last window.  Edges: every start
    parity (contigs of odd and even lengths back to back), 0 .. 4 windows, pairs that straddle a chunk and a stage, a
    homopolymer whose half bin reaches exactly 65 535 and one a window longer (handed to the wave-per-contig kernel), a
    stream that ends word-aligned; both shapes of the kernel, forced and chosen by the batch statistics."""
    from oracle import oracle
    from phamers_amd import device, synth
    rng = np.random.default_rng(77)
    k, D = 4, 256
    batches = []
    batches.append([int(x) for x in rng.integers(0, 300, 333)] + [3, 4, 5, 6, 7, 8, 67, 68, 69, 131, 132, 133, 1027, 1028, 16 * 9])
    batches.append([int(x) for x in rng.integers(2000, 9000, 70)] + [300001, 4999, 5000, 5001, 5002] + [int(x) for x in rng.integers(1, 64, 31)])
    batches.append(None)   # the 16-bit limit
    for bi, lens in enumerate(batches):
        if lens is None:
            seqs = ["A" * 131073, "A" * 131074, "AT" * 65540, synth.synth_contig(5, 0, 120000), "C" * 131072, synth.synth_contig(5, 1, 140001),
                    synth.synth_contig(5, 2, 99999), "G" * 7]
            lens = [len(x) for x in seqs]
        else:
            seqs = [synth.synth_contig(40 + bi, i, L) for i, L in enumerate(lens)]
        T = sum(lens)
        offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum(lens)
        raw = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
        d_raw = device.DeviceArray.from_host(ctx, raw)
        d_packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
        d_mask = device.DeviceArray(ctx, device.mask_words(T), np.uint32)
        device.pack_ascii(ctx, d_raw, T, d_packed, d_mask, None)
        d_off = device.DeviceArray.from_host(ctx, offsets)
        want = oracle.count(seqs, k).reshape(len(lens), D)
        for lanes in ("q", "Q", "p", "P"):
            ctx.set_option("count_lanes", lanes)
            try:
                ctx.profile_reset()
                ctx.profile_enable(True)
                d_counts = device.DeviceArray.from_host(ctx, np.full((len(lens), D), 0xABCD, np.uint32))
                d_nwin = device.DeviceArray.from_host(ctx, np.full(len(lens), 0xABCD, np.uint32))
                device.count(ctx, d_packed, None, T, d_off, len(lens), k, d_counts, d_nwin)
                got = d_counts.to_host().astype(np.int64)
                ctx.profile_enable(False)
                assert "phk_count_pairs_kernel" in ctx.profile(), (bi, lanes)
                bad = np.flatnonzero((got != want).any(axis=1))
                assert bad.size == 0, (bi, lanes, bad[:8], [lens[i] for i in bad[:8]])
                assert np.array_equal(d_nwin.to_host().astype(np.int64), want.sum(axis=1)), (bi, lanes)
            finally:
                ctx.set_option("count_lanes", "")
        for a in (d_raw, d_packed, d_mask, d_off):
            a.free()


@pytest.mark.gpu
def test_count_directory_sums_per_file(kmer, tmp_path):
    """kmer.count_directory (scripts/kmer.py:143-181): one row per FASTA file = column sums over its records,
    id of the first record; files without the identifier or without countable sequence are skipped."""
    from oracle import oracle
    from phamers_amd import synth
    want = {}
    for fi, nrec in enumerate((1, 3, 5)):
        seqs = [synth.synth_contig(70 + fi, r, 900 + 37 * r) for r in range(nrec)]
        with open(tmp_path / ("genome_%d.fna" % fi), "w") as fh:
            for r, sq in enumerate(seqs):
                fh.write(">GB%03d%02d.1 some description\n" % (fi, r))
                for p0 in range(0, len(sq), 70):
                    fh.write(sq[p0:p0 + 70] + "\n")
        want["GB%03d00.1" % fi] = oracle.count(seqs, 4).reshape(nrec, 256).sum(axis=0)
    (tmp_path / "notes.txt").write_text("not a fasta file")
    (tmp_path / "empty.fna").write_text(">ZZ00000.1 only n\nNNNNNNNNNN\n")
    ids, counts = kmer.count_directory(str(tmp_path), 4)
    assert sorted(ids) == sorted(want) and counts.shape == (3, 256) and counts.dtype == np.float64
    for i, name in enumerate(ids):
        assert np.array_equal(counts[i], want[name].astype(np.float64)), name


@pytest.mark.gpu
def test_ragged_skewed_generator_matches_its_host_statement(ctx):
    """phk_synth_ragged_dev (heavy-tailed lengths, per-contig GC, invalid bases) against synth.synth_ragged_contig:
    the device batch is what the host re-derives, checked through the counts of every contig (k = 4 and 5, mask)."""
    from oracle import oracle
    from phamers_amd import device, synth
    n = 150
    lens = synth.ragged_lengths(7, n, lo=400, hi=30000)
    lens[5] = 0
    lens[9] = 3
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    T = int(offs[-1])
    d_off = device.DeviceArray.from_host(ctx, offs)
    d_packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
    d_mask = device.DeviceArray(ctx, device.mask_words(T), np.uint32)
    device.synth_ragged(ctx, 3, 17, n, d_off, T, d_packed, d_mask, gc_spread_permille=700, invalid_ppm=5000)
    seqs = [synth.synth_ragged_contig(3, 17 + c, int(lens[c]), 700, 5000) for c in range(n)]
    gc = [(s.count("G") + s.count("C")) / max(len(s), 1) for s in seqs if len(s) > 1000]
    assert len(gc) > 20 and max(gc) - min(gc) > 0.3  # the composition really is skewed per contig
    for k in (4, 5):
        d_counts = device.DeviceArray(ctx, (n, 4 ** k), np.uint32)
        device.count(ctx, d_packed, d_mask, T, d_off, n, k, d_counts)
        assert np.array_equal(d_counts.to_host().astype(np.int64), oracle.count(seqs, k).reshape(n, -1)), k


@pytest.mark.gpu
@pytest.mark.parametrize("k,masked", [(4, False), (4, True), (5, True), (3, False)])
def test_ragged_batch_sorted_slots_and_pieces(ctx, k, masked):
    """A heavy-tailed batch in arbitrary order: the slot kernel walks it in length-bucketed order and contigs far
    above the mean are counted in 32768-window pieces by several waves (atomic adds onto the zeroed row).  Counts and
    window totals equal the oracle's; the legacy stand-down path (count_sort off) and the wave-per-contig kernel
    agree bit for bit."""
    from oracle import oracle
    from phamers_amd import device, synth
    n = 700
    lens = synth.ragged_lengths(11 + k, n, lo=300, hi=400000, shape=0.9)
    lens[3] = 0
    lens[10] = k - 1
    lens[11] = k
    lens[20] = 399999
    lens[30], lens[40] = 150000, 98304 + k - 1        # three pieces exactly
    lens[50], lens[51] = 65536 + k, 65536 + k - 1     # just above / exactly at the hand-over threshold (in windows)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    T = int(offs[-1])
    d_off = device.DeviceArray.from_host(ctx, offs)
    d_packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
    d_mask = device.DeviceArray(ctx, device.mask_words(T), np.uint32)
    ppm = 3000 if masked else 0
    device.synth_ragged(ctx, 5, 0, n, d_off, T, d_packed, d_mask, gc_spread_permille=600, invalid_ppm=ppm)
    seqs = [synth.synth_ragged_contig(5, c, int(lens[c]), 600, ppm) for c in range(n)]
    want = oracle.count(seqs, k).reshape(n, -1)
    assert (lens > 70000).sum() >= 4          # some contigs really are cut into pieces
    D = 4 ** k
    got = {}
    for name, opts in (("sorted", {}), ("standdown", {"count_sort": "0"}),
                       ("wave", {"count_lanes": "0"})):
        for key, val in opts.items():
            ctx.set_option(key, val)
        d_counts = device.DeviceArray.from_host(ctx, np.full((n, D), 0xABCD, np.uint32))
        d_nwin = device.DeviceArray.from_host(ctx, np.full(n, 0xABCD, np.uint32))
        device.count(ctx, d_packed, d_mask if masked else None, T, d_off, n, k, d_counts, d_nwin)
        got[name] = d_counts.to_host()
        assert np.array_equal(got[name].astype(np.int64), want), (name, k, masked)
        assert np.array_equal(d_nwin.to_host().astype(np.int64), want.sum(axis=1)), (name, k, masked)
        ctx.set_option("count_sort", "1")
        ctx.set_option("count_lanes", "")
    for name in got:
        assert np.array_equal(got["sorted"], got[name]), name


@pytest.mark.gpu
def test_kmer_command_line_end_to_end(tmp_path):
    """`python -m phamers_amd.kmer` (scripts/kmer.py:283-334): a FASTA file -> one row per record; a directory of genome
    files -> one row per file (column sums); the CSV reads back through read_feature_file and equals the oracle; the
    '#' header carries the Namespace as the reference's files do."""
    from oracle import oracle
    from phamers_amd import fileIO, kmer as pk, synth
    seqs = [synth.synth_contig(90, r, 700 + 53 * r, 500 if r % 3 == 0 else 0) for r in range(9)]
    fasta = tmp_path / "contigs.fasta"
    with open(fasta, "w") as fh:
        for r, sq in enumerate(seqs):
            fh.write(">SuperContig_%d_length_%d_ID_%d\n%s\n" % (r, len(sq), r, sq))
    out = tmp_path / "features.csv"
    pk.main([str(fasta), str(out), "-k", "4"])
    ids, counts = fileIO.read_feature_file(str(out))
    assert [str(x) for x in ids] == [str(r) for r in range(9)]
    assert np.array_equal(counts, oracle.count(seqs, 4))
    head = [ln for ln in open(out).read().splitlines() if ln.startswith("#")]
    assert any("kmer_length:\t4" in ln for ln in head) and any("symbols:\t'ATGC'" in ln for ln in head)
    gdir = tmp_path / "genomes"
    gdir.mkdir()
    want = {}
    for fi in range(3):
        recs = [synth.synth_contig(91 + fi, r, 1500 + 11 * r) for r in range(fi + 1)]
        with open(gdir / ("g%d.fna" % fi), "w") as fh:
            for r, sq in enumerate(recs):
                fh.write(">GB%03d%02d.1 genome\n%s\n" % (fi, r, sq))
        want["GB%03d00.1" % fi] = oracle.count(recs, 5).reshape(len(recs), -1).sum(axis=0)
    out2 = tmp_path / "ref_features.csv"
    pk.main([str(gdir), str(out2), "-k", "5", "-id", ".fna"])
    ids2, counts2 = fileIO.read_feature_file(str(out2))
    assert sorted(str(x) for x in ids2) == sorted(want)
    for i, name in enumerate(ids2):
        assert np.array_equal(counts2[i], want[str(name)])


@pytest.mark.gpu
def test_multi_chunk_upload_through_the_staging_buffers(ctx):
    """phk_batch_from_ascii on 150 MB of bases: three 64 MB upload chunks through the context's two pinned staging buffers
    (filled by host threads while the previous chunk is on the bus), contigs that straddle the chunk cuts, a stretch of
    non-symbols across one cut.  Counts = the single-copy device path (DeviceArray.from_host + pack + count) on the same
    bytes, bit for bit, and = the oracle on a sample of contigs including those on the cuts."""
    import ctypes
    from oracle import oracle
    from phamers_amd import _lib, device
    rng = np.random.default_rng(2024)
    T = 150 * (1 << 20) + 12345
    bases = np.frombuffer(b"ATGC", dtype=np.uint8)[rng.integers(0, 4, size=T, dtype=np.uint8)]
    chunk = 64 << 20
    bases[chunk - 40:chunk + 40] = ord("N")               # non-symbols across the first cut
    bases[rng.integers(0, T, size=2000)] = ord("n")        # and sprinkled (lower case is not a symbol)
    lens = rng.integers(1000, 9000, size=40000)
    cuts = np.cumsum(lens)
    cuts = cuts[cuts < T]
    offsets = np.concatenate(([0], cuts, [T])).astype(np.uint64)
    n = len(offsets) - 1
    k = 4
    h = ctypes.c_void_p()
    _lib.check(ctx.lib.phk_batch_from_ascii(ctx.handle, _lib.ptr(bases), _lib.ptr(offsets), n, k, b"ATGC", ctypes.byref(h)))
    batch = _lib.Batch(ctx, h)
    try:
        assert batch.n == n and batch.total_bases == T and batch.any_invalid
        got = batch.counts_u32()
    finally:
        batch.close()
    d_raw = device.DeviceArray.from_host(ctx, bases)
    d_packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
    d_mask = device.DeviceArray(ctx, device.mask_words(T), np.uint32)
    d_flag = device.DeviceArray(ctx, 1, np.uint32)
    device.pack_ascii(ctx, d_raw, T, d_packed, d_mask, d_flag)
    d_off = device.DeviceArray.from_host(ctx, offsets)
    d_counts = device.DeviceArray(ctx, (n, 4 ** k), np.uint32)
    d_nwin = device.DeviceArray(ctx, n, np.uint32)
    device.count(ctx, d_packed, d_mask, T, d_off, n, k, d_counts, d_nwin)
    assert np.array_equal(got, d_counts.to_host())
    on_cuts = [int(np.searchsorted(offsets, c, side="right") - 1) for c in (chunk, 2 * chunk)]
    sample = sorted(set(on_cuts + [0, n - 1] + rng.integers(0, n, size=60).tolist()))
    seqs = [bases[int(offsets[c]):int(offsets[c + 1])].tobytes().decode() for c in sample]
    assert np.array_equal(got[sample].astype(np.int64), oracle.count(seqs, k))


@pytest.mark.gpu
def test_large_copies_through_the_staging_buffers_round_trip(ctx):
    """phk_memcpy_h2d / _d2h above 128 MB go through the two pinned 64 MB staging buffers in chunks (host threads on one
    side, the bus on the other): sizes that end inside a chunk, on a chunk's last byte and one past it come back as sent."""
    from phamers_amd import device
    rng = np.random.default_rng(99)
    chunk = 64 << 20
    for nbytes in (2 * chunk, 2 * chunk + 1, 3 * chunk - 1, 4 * chunk + 12345):
        src = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
        d = device.DeviceArray.from_host(ctx, src)
        back = d.to_host()
        assert back.shape == src.shape and np.array_equal(back, src), nbytes


@pytest.mark.gpu
def test_staged_copies_from_two_host_threads_at_once():
    """Two host threads, each with a context (and so staging buffers) of its own, copy 200 MB to the device and back five
    times each at the same time; every round trip is exact.  (A persistent pool for the copies' host-side loops was measured
    against threads started per chunk: no difference, not kept.)"""
    import threading
    from phamers_amd import _lib, device
    errors = []

    def work(seed):
        try:
            ctx = _lib.Context(_lib.default_device())
            rng = np.random.default_rng(seed)
            src = rng.integers(0, 256, size=(200 << 20) + seed, dtype=np.uint8)
            for _ in range(5):
                d = device.DeviceArray.from_host(ctx, src)
                if not np.array_equal(d.to_host(), src):
                    errors.append("mismatch in thread %d" % seed)
                del d
            ctx.close()
        except Exception as e:   # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))
    threads = [threading.Thread(target=work, args=(s,)) for s in (1, 2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a copy thread hangs"
    assert not errors, errors
