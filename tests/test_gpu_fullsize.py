"""BASELINE configs[1] at full size (k = 4, 1M x 5 kb contigs on one GPU) through size-independent
properties: the two count kernels agree bit for bit, every row sums to L - k + 1, scores do not depend on
where in a batch (which tile, which workgroup, which launch) a contig sits, combo = knn + kmeans, and a
random sample of rows matches the oracle."""
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

    # position independence: a permuted sub-batch (different tiles, workgroups and launch size) scores identically
    perm = rng.permutation(N)[:300001]
    d_sub = device.DeviceArray.from_host(ctx, counts[perm])
    d_sub_scores = device.DeviceArray(ctx, len(perm), np.float64)
    device.score_counts(ctx, model, d_sub, len(perm), "combo", d_sub_scores, d_status)
    assert np.array_equal(d_sub_scores.to_host(), out["combo"][perm])
    model.close()
