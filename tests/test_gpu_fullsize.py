"""BASELINE configs[1] at full size (k = 4, 1M x 5 kb contigs on one GPU) through size-independent
properties: the two count kernels agree bit for bit, every row sums to L - k + 1, scores do not depend on
where in a batch (which tile, which workgroup, which launch) a contig sits, combo = knn + kmeans, a
random sample of rows matches the oracle, and ALL 10^6 scores equal those of the float64 brute-force path."""
import os

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu
N, L, K = 1000000, 5000, 4


def _model(ctx):
    from phamers_amd import _lib
    g = helpers.load_npz("scoring_k4.npz")
    with np.load(os.path.join(helpers.GOLDEN, "ref_features.npz")) as z:
        pos = z["pos_counts"].astype(np.float64)
        neg = z["neg_counts"].astype(np.float64)
    pos /= pos.sum(axis=1, keepdims=True)
    neg /= neg.sum(axis=1, keepdims=True)
    n = min(len(pos), len(neg))
    return _lib.Model(ctx, pos[:n], neg[:n], g["cpos_eq"], g["cneg_eq"], 3), pos[:n], neg[:n], g


def test_full_size_batch_properties(monkeypatch):
    from oracle import oracle
    from phamers_amd import _lib, device, synth
    ctx = _lib.get_context()
    model, pos, neg, g = _model(ctx)
    T = N * L
    d_packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
    d_off = device.DeviceArray(ctx, N + 1, np.uint64)
    device.synth_packed(ctx, 0, 0, N, L, d_packed, d_off)
    d_counts = device.DeviceArray(ctx, (N, 256), np.uint32)
    d_nwin = device.DeviceArray(ctx, N, np.uint32)
    d_scores = device.DeviceArray(ctx, N, np.float64)
    d_status = device.DeviceArray.from_host(ctx, np.zeros(1, np.uint32))

    # counting: slot kernel (default) vs wave-per-contig kernel, bit for bit; row sums
    device.count(ctx, d_packed, None, T, d_off, N, K, d_counts, d_nwin)
    counts = d_counts.to_host()
    assert np.all(d_nwin.to_host() == L - K + 1)
    assert np.all(counts.sum(axis=1, dtype=np.uint64) == L - K + 1)
    ctx.set_option("count_lanes", "0")
    device.count(ctx, d_packed, None, T, d_off, N, K, d_counts, d_nwin)
    ctx.set_option("count_lanes", "")
    assert np.array_equal(d_counts.to_host(), counts)
    # linearity: the column sums of the batch are the counts of the concatenated contigs' windows that do not
    # straddle a contig boundary -- checked against the oracle on a sample instead: rows of a random sample
    rng = np.random.default_rng(0)
    sample = np.sort(rng.choice(N, 96, replace=False))
    want_counts = oracle.count([synth.synth_contig(0, int(c), L) for c in sample], K)
    assert np.array_equal(counts[sample].astype(np.int64), want_counts)

    # scoring the whole batch
    out = {}
    for method in ("knn", "kmeans", "combo"):
        device.score_counts(ctx, model, d_counts, N, method, d_scores, d_status)
        out[method] = d_scores.to_host()
        assert d_status.to_host()[0] == 0
    n_fallback, n_exact = ctx.score_stats()
    assert n_fallback < 100 and n_exact < N // 10
    assert set(np.unique(out["knn"])) <= {-1.0, 1.0}
    assert np.all(np.abs(out["kmeans"]) < 1.0)
    assert np.allclose(out["combo"], out["knn"] + out["kmeans"], rtol=0, atol=1e-15)
    # oracle on the sample
    q = oracle.normalize_counts(want_counts)
    want = oracle.knn_score_points(q, pos, neg, 3) + oracle.centroid_score_points_fast(q, g["cpos_eq"], g["cneg_eq"])
    assert helpers.rel_err(out["combo"][sample], want) < 1e-6

    # the whole batch through the float64 brute-force path (no MFMA proposal, no certification logic): every vote
    # identical, every metric equal to rounding -- 10^6 queries, not a sample
    ctx.set_option("force_exact", "1")
    exact = {}
    for method in ("knn", "kmeans"):
        device.score_counts(ctx, model, d_counts, N, method, d_scores, d_status)
        exact[method] = d_scores.to_host()
    ctx.set_option("force_exact", "0")
    assert np.array_equal(out["knn"], exact["knn"])
    assert helpers.rel_err(out["kmeans"], exact["kmeans"]) < 1e-9

    # position independence: a permuted sub-batch (different tiles, workgroups and launch size) scores identically
    perm = rng.permutation(N)[:300001]
    d_sub = device.DeviceArray.from_host(ctx, counts[perm])
    d_sub_scores = device.DeviceArray(ctx, len(perm), np.float64)
    device.score_counts(ctx, model, d_sub, len(perm), "combo", d_sub_scores, d_status)
    assert np.array_equal(d_sub_scores.to_host(), out["combo"][perm])
    model.close()


def test_config2_k5_10M_x_10kb_counting_and_scoring():
    """BASELINE configs[2] at full size: k = 5 (1024 bins), 10 M x 10 kb contigs on one GPU (100 Gbases; 25 GB packed,
    41 GB of counts per copy).  Every row sums to L - k + 1, the slot kernel and the wave-per-contig kernel agree bit
    for bit (both checked on the device), 96 sampled rows equal the oracle's counts, and ALL 10 M rows are scored against
    2255 + 2255 synthetic reference genomes in one call -- ten scoring batches of 2^20 rows, the per-row hand-over queues
    (three-digit re-sweep, f16 sweep, brute force) carried per batch -- with the oracle on a sample spread over all ten
    batches, and a slice re-scored alone (other batch split, other tiles) equal bit for bit."""
    from oracle import oracle
    from phamers_amd import _lib, device, synth, workloads
    cfg = workloads.CONFIGS[2]
    n, L2, k = cfg["contigs"], cfg["length"], cfg["k"]
    D = 4 ** k
    T = n * L2
    ctx = _lib.get_context()
    packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
    offsets = device.DeviceArray(ctx, n + 1, np.uint64)
    counts = device.DeviceArray(ctx, (n, D), np.uint32)
    counts_w = device.DeviceArray(ctx, (n, D), np.uint32)
    nwin = device.DeviceArray(ctx, n, np.uint32)
    device.synth_packed(ctx, 0, 0, n, L2, packed, offsets)
    device.count(ctx, packed, None, T, offsets, n, k, counts, nwin)
    ctx.set_option("count_lanes", "0")
    device.count(ctx, packed, None, T, offsets, n, k, counts_w, None)
    ctx.set_option("count_lanes", "")
    ctx.sync()
    assert np.all(nwin.to_host() == L2 - k + 1)
    assert nwin.to_host().shape == (10000000,) and counts.nbytes == 10000000 * 1024 * 4
    bad_rows, bad_words = device.check_counts(ctx, counts, counts_w, n, D, L2 - k + 1)
    assert bad_rows == 0 and bad_words == 0
    counts_w.free()
    rng = np.random.default_rng(2)
    sample = np.sort(rng.choice(n, 96, replace=False))
    got = device.read_rows(ctx, counts, sample, D).astype(np.int64)
    want = oracle.count([synth.synth_contig(0, int(c), L2) for c in sample], k)
    assert np.array_equal(got, want)

    # scoring a slice of the batch (general-D count-exact MFMA path), sample against the oracle
    pos, neg, cpos, cneg = workloads.synthetic_reference(ctx, k, cfg["refs"], cfg["ref_length"])
    model = _lib.Model(ctx, pos, neg, cpos, cneg, 3)
    scores = device.DeviceArray(ctx, n, np.float64)
    status = device.DeviceArray(ctx, 1, np.uint32)
    device.score_counts(ctx, model, counts, n, "combo", scores, status)
    assert status.to_host()[0] == 0
    n_fallback, _ = ctx.score_stats()
    assert n_fallback < n // 1000, n_fallback
    all_scores = scores.to_host()
    assert np.isfinite(all_scores).all() and np.all(np.abs(all_scores) < 2.0)
    # six rows of each of the ten scoring batches (2^20 rows each; the last one holds 562 816)
    pick = np.sort(np.concatenate([b * (1 << 20) + rng.choice(min(1 << 20, n - b * (1 << 20)), 6, replace=False) for b in range(10)]))
    assert pick[-1] >= 9 * (1 << 20)
    q = oracle.normalize_counts(device.read_rows(ctx, counts, pick, D).astype(np.int64))
    want_s = oracle.knn_score_points(q, pos, neg, 3) + oracle.centroid_score_points_fast(q, cpos, cneg)
    assert helpers.rel_err(all_scores[pick], want_s) < 1e-6
    # a slice that straddles two batches of the full call, scored alone: route independence at full size
    lo, ns = 3 * (1 << 20) - 70000, 262144
    sub_scores = device.DeviceArray(ctx, ns, np.float64)
    device.score_counts(ctx, model, counts.ptr + lo * D * 4, ns, "combo", sub_scores, status)   # (a device pointer into the matrix)
    assert np.array_equal(sub_scores.to_host(), all_scores[lo:lo + ns])
    model.close()
    for a in (packed, offsets, counts, nwin, scores, status, sub_scores):
        a.free()


def test_config4_k6_50k_row_reference_131072_queries():
    """BASELINE configs[4] at full size on one GPU: k = 6 (D = 4096), 50 000-row reference matrix, 131 072 queries of
    10 kb, counted and scored on the device.  Sampled queries against oracle.score_points (float64 direct-difference
    brute force over all 50 000 rows), the brute-force fallback share bounded, combo = knn + kmeans, and scores
    independent of a query's position in the batch."""
    from oracle import oracle
    from phamers_amd import _lib, device, synth, workloads
    cfg = workloads.CONFIGS[4]
    n, L4, k = cfg["contigs"], cfg["length"], cfg["k"]
    D = 4 ** k
    T = n * L4
    ctx = _lib.get_context()
    pos, neg, cpos, cneg = workloads.synthetic_reference(ctx, k, cfg["refs"], cfg["ref_length"])
    assert pos.shape == (25000, D) and neg.shape == (25000, D)
    model = _lib.Model(ctx, pos, neg, cpos, cneg, 3)
    packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
    offsets = device.DeviceArray(ctx, n + 1, np.uint64)
    counts = device.DeviceArray(ctx, (n, D), np.uint32)
    scores = device.DeviceArray(ctx, n, np.float64)
    status = device.DeviceArray(ctx, 1, np.uint32)
    device.synth_packed(ctx, 0, 0, n, L4, packed, offsets)
    out = {}
    for method in ("combo", "knn", "kmeans"):
        if method == "combo":   # the whole path: count -> normalise -> score
            device.count_score(ctx, model, packed, None, T, offsets, n, k, method, counts, scores, status)
        else:
            device.score_counts(ctx, model, counts, n, method, scores, status)
        out[method] = scores.to_host()
        assert status.to_host()[0] == 0
        n_fallback, _ = ctx.score_stats()
        assert n_fallback < n // 200, (method, n_fallback)
    assert set(np.unique(out["knn"])) <= {-1.0, 1.0}
    assert np.allclose(out["combo"], out["knn"] + out["kmeans"], rtol=0, atol=1e-15)
    assert device.check_counts(ctx, counts, None, n, D, L4 - k + 1) == (0, 0)
    # the oracle on a sample: counts bit-exact, scores within 1e-6 relative
    rng = np.random.default_rng(4)
    pick = np.sort(rng.choice(n, 24, replace=False))
    want_counts = oracle.count([synth.synth_contig(0, int(c), L4) for c in pick], k)
    host_counts = counts.to_host()
    assert np.array_equal(host_counts[pick].astype(np.int64), want_counts)
    q = oracle.normalize_counts(want_counts)
    want = oracle.knn_score_points(q, pos, neg, 3) + oracle.centroid_score_points_fast(q, cpos, cneg, chunk=8)
    assert helpers.rel_err(out["combo"][pick], want) < 1e-6
    # position independence: a permuted sub-batch of another size (other tiles, workgroups, batch split)
    perm = rng.permutation(n)[:40001]
    sub = device.DeviceArray.from_host(ctx, host_counts[perm])
    sub_scores = device.DeviceArray(ctx, len(perm), np.float64)
    device.score_counts(ctx, model, sub, len(perm), "combo", sub_scores, status)
    assert np.array_equal(sub_scores.to_host(), out["combo"][perm])
    model.close()
    for a in (packed, offsets, counts, scores, status, sub, sub_scores):
        a.free()


def test_config3_per_rank_share_12_5M_contigs():
    """BASELINE configs[3] (k = 4, 100 M x 5 kb contigs over 8 GPUs) as ONE rank sees it: rank 3's share of 12.5 M
    contigs (62.5 Gbases; 15.6 GB packed, 12.8 GB of counts) through phk_count_score_dev in one call, i.e. 12 scoring
    batches of 2^20 queries.  Checked through size-independent properties: every row sums to L - k + 1 (on the device),
    sampled rows equal the oracle's counts and scores, a slice equals the float64 brute-force path, and a contig's score
    does not depend on the shard it is scored in (the same contigs as a 1 M-contig batch of their own: bit-identical,
    which is the invariant SURVEY 8(e) asks of the sharding)."""
    from oracle import oracle
    from phamers_amd import _lib, device, synth, workloads
    cfg = workloads.CONFIGS[3]
    world, rank = 8, 3
    n, L3, k = cfg["contigs_total"] // world, cfg["length"], cfg["k"]
    assert n == 12500000
    first = rank * n
    D = 4 ** k
    T = n * L3
    ctx = _lib.get_context()
    model, pos, neg, g = _model(ctx)
    packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
    offsets = device.DeviceArray(ctx, n + 1, np.uint64)
    counts = device.DeviceArray(ctx, (n, D), np.uint32)
    scores = device.DeviceArray(ctx, n, np.float64)
    status = device.DeviceArray.from_host(ctx, np.zeros(1, np.uint32))
    device.synth_packed(ctx, 0, first, n, L3, packed, offsets)
    device.count_score(ctx, model, packed, None, T, offsets, n, k, "combo", counts, scores, status)
    got = scores.to_host()
    assert status.to_host()[0] == 0
    n_fallback, n_exact = ctx.score_stats()
    assert n_fallback < 1000 and n_exact < n // 10
    assert device.check_counts(ctx, counts, None, n, D, L3 - k + 1) == (0, 0)
    assert np.all(np.isfinite(got)) and np.all(np.abs(got) < 1.7616)
    # the oracle on a sample spread over all 12 scoring batches (incl. the first and the last contig of the shard)
    rng = np.random.default_rng(3)
    sample = np.unique(np.concatenate(([0, n - 1], rng.choice(n, 70, replace=False))))
    want_counts = oracle.count([synth.synth_contig(0, first + int(c), L3) for c in sample], k)
    assert np.array_equal(device.read_rows(ctx, counts, sample, D).astype(np.int64), want_counts)
    q = oracle.normalize_counts(want_counts)
    want = oracle.knn_score_points(q, pos, neg, 3) + oracle.centroid_score_points_fast(q, g["cpos_eq"], g["cneg_eq"])
    assert helpers.rel_err(got[sample], want) < 1e-6
    # a slice that straddles a scoring-batch boundary through the float64 brute-force path
    lo, m = 3 * (1 << 20) - 100000, 200000
    sl_counts = counts.ptr + lo * D * 4
    sl = device.DeviceArray(ctx, m, np.float64)
    ctx.set_option("force_exact", "1")
    device.score_counts(ctx, model, sl_counts, m, "knn", sl, status)
    knn = sl.to_host()
    device.score_counts(ctx, model, sl_counts, m, "kmeans", sl, status)
    kme = sl.to_host()
    ctx.set_option("force_exact", "0")
    assert np.array_equal(np.trunc(got[lo:lo + m] + np.where(got[lo:lo + m] > 0, 0.5, -0.5)), knn)   # the vote is the integer part
    assert helpers.rel_err(got[lo:lo + m] - knn, kme) < 1e-9
    # shard invariance: the same contigs scored as a batch of their own (other batch split, other queues)
    device.score_counts(ctx, model, sl_counts, m, "combo", sl, status)
    assert np.array_equal(sl.to_host(), got[lo:lo + m])
    model.close()
    for a in (packed, offsets, counts, scores, status, sl):
        a.free()


def test_batch_beyond_two_pow_32_grid_threads():
    """30M x 5 kb contigs in one call: 2^37.1 bases, so every one-thread-per-32-bases launch (the synthetic generator, the
    packer) needs more than 2^32 threads and has to go out in slices -- a grid that large is otherwise launched short
    without any error (round 5: rows beyond the first 2.5M were never generated).  Rows from the far end of the batch match
    the oracle, every row sums to L - k + 1, and the run certifies as the 1M batch does (no flood of brute-forced rows)."""
    from oracle import oracle
    from phamers_amd import _lib, device, synth
    ctx = _lib.get_context()
    model, pos, neg, g = _model(ctx)
    n = 30000000
    T = n * L
    assert (T + 31) // 32 > 1 << 32
    d_packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
    d_off = device.DeviceArray(ctx, n + 1, np.uint64)
    device.synth_packed(ctx, 0, 0, n, L, d_packed, d_off)
    d_counts = device.DeviceArray(ctx, (n, 256), np.uint32)
    d_scores = device.DeviceArray(ctx, n, np.float64)
    d_status = device.DeviceArray.from_host(ctx, np.zeros(1, np.uint32))
    device.count_score(ctx, model, d_packed, None, T, d_off, n, K, "combo", d_counts, d_scores, d_status)
    assert d_status.to_host()[0] == 0
    assert device.check_counts(ctx, d_counts, None, n, 256, L - K + 1) == (0, 0)
    n_fallback, n_exact = ctx.score_stats()
    assert n_fallback < 3000 and n_exact < n // 10
    rng = np.random.default_rng(5)
    sample = np.sort(np.concatenate([rng.choice(n, 40, replace=False), np.arange(n - 8, n)]))
    rows = np.concatenate([d_counts.rows_to_host(int(r), 1) for r in sample])
    want_counts = oracle.count([synth.synth_contig(0, int(c), L) for c in sample], K)
    assert np.array_equal(rows.astype(np.int64), want_counts)
    q = oracle.normalize_counts(want_counts)
    want = oracle.knn_score_points(q, pos, neg, 3) + oracle.centroid_score_points_fast(q, g["cpos_eq"], g["cneg_eq"])
    got = d_scores.to_host()[sample]
    assert helpers.rel_err(got, want) < 1e-6
    for a in (d_packed, d_off, d_counts, d_scores, d_status):
        a.free()
