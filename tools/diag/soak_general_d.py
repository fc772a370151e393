#!/usr/bin/env python3
"""One-off soak on the GPU box for the general-D scoring path (k = 5, 6): ragged, GC-skewed, masked contigs scored through the
default two-digit int8 sweep, the three-digit one and the float64 brute-force path of the same library.  Every vote must agree
with the float64 path, every metric to 1e-7, and the two int8 sweeps must agree bit for bit.  Lengths reach far beyond the int8
operand's range (bins more than 127 from the row's centre), so batches are declined to the f16 sweep and rows are queued."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phamers_amd import _lib, device, synth

ctx = _lib.get_context()
for k, n_ref, cases in ((5, 3000, ((11, 150000, 2000, 120000), (12, 200000, 300, 20000))),
                        (6, 2000, ((13, 40000, 3000, 400000), (14, 60000, 1000, 30000)))):
    D = 4 ** k
    # reference: counts of synthetic genomes, half of them skewed
    L = 40000
    T = n_ref * L
    d_packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
    d_off = device.DeviceArray(ctx, n_ref + 1, np.uint64)
    device.synth_packed(ctx, 500 + k, 0, n_ref, L, d_packed, d_off)
    d_rc = device.DeviceArray(ctx, (n_ref, D), np.uint32)
    device.count(ctx, d_packed, None, T, d_off, n_ref, k, d_rc)
    rc = d_rc.to_host().astype(np.float64)
    rc[: n_ref // 2] *= 1.0 + 0.3 * np.sin(np.arange(D) * 0.37)
    ref = rc / rc.sum(axis=1, keepdims=True)
    pos, neg = ref[: n_ref // 2], ref[n_ref // 2:]
    cpos = np.stack([pos[i::20].mean(axis=0) for i in range(20)])
    cneg = np.stack([neg[i::20].mean(axis=0) for i in range(20)])
    model = _lib.Model(ctx, pos, neg, cpos, cneg, 3)
    for x in (d_packed, d_off, d_rc):
        x.free()
    for seed, n, lo, hi in cases:
        lens = synth.ragged_lengths(seed, n, lo=lo, hi=hi)
        offs = np.zeros(n + 1, dtype=np.uint64); offs[1:] = np.cumsum(lens); T = int(offs[-1])
        d_off = device.DeviceArray.from_host(ctx, offs)
        d_packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
        d_mask = device.DeviceArray(ctx, device.mask_words(T), np.uint32)
        device.synth_ragged(ctx, seed, 0, n, d_off, T, d_packed, d_mask, gc_spread_permille=500, invalid_ppm=1500)
        d_counts = device.DeviceArray(ctx, (n, D), np.uint32)
        device.count(ctx, d_packed, d_mask, T, d_off, n, k, d_counts)
        d_scores = device.DeviceArray(ctx, n, np.float64)
        d_status = device.DeviceArray.from_host(ctx, np.zeros(1, np.uint32))
        out = {}
        for name in ("i8", "i83", "exact"):
            ctx.set_option("force_exact", "1" if name == "exact" else "0")
            ctx.set_option("proposal", "i83" if name == "i83" else "")
            ctx.set_option("score_batch", "65536")     # several batches: some declined (long contigs), some not
            res = {}
            for method in ("knn", "kmeans"):
                device.score_counts(ctx, model, d_counts, n, method, d_scores, d_status)
                res[method] = d_scores.to_host()
            out[name] = res
            if name != "exact":
                print("  k=%d seed %d %s stats %s" % (k, seed, name, ctx.score_stats_ex()))
        ctx.set_option("force_exact", "0"); ctx.set_option("proposal", ""); ctx.set_option("score_batch", "0")
        ok = ~np.isnan(out["exact"]["knn"])
        a, c, b = out["i8"], out["i83"], out["exact"]
        for m in ("knn", "kmeans"):
            assert np.array_equal(a[m][ok], c[m][ok]), (m, int((a[m][ok] != c[m][ok]).sum()))
        assert np.array_equal(np.isnan(a["knn"]), ~ok)
        assert np.array_equal(a["knn"][ok], b["knn"][ok]), int((a["knn"][ok] != b["knn"][ok]).sum())
        rel = np.max(np.abs(a["kmeans"][ok] - b["kmeans"][ok]) / np.maximum(np.abs(b["kmeans"][ok]), 1e-300))
        assert rel < 1e-7, rel
        print("k=%d seed %d: %d contigs (%d zero-count), mean %d bases: votes identical, int8 sweeps bit-equal, metric rel err %.1e"
              % (k, seed, n, int((~ok).sum()), T // n, rel))
        for x in (d_off, d_packed, d_mask, d_counts, d_scores, d_status):
            x.free()
    model.close()
print("soak OK")
