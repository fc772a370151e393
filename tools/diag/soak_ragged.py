#!/usr/bin/env python3
"""One-off soak on the GPU box: the ragged, GC-skewed, masked workload (and a short-contig one) scored by the MFMA path
and by the float64 brute-force path of the same library; every vote must agree, every metric to 1e-9."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phamers_amd import _lib, device, synth, workloads

ctx = _lib.get_context()
pos, neg, cpos, cneg = workloads.phamers_reference()
model = _lib.Model(ctx, pos, neg, cpos, cneg, 3)
for seed, n, lo, hi in ((1000, 200000, 5000, 500000), (7, 300000, 40, 3000), (9, 100000, 300, 60000)):
    lens = synth.ragged_lengths(seed, n, lo=lo, hi=hi)
    offs = np.zeros(n + 1, dtype=np.uint64); offs[1:] = np.cumsum(lens); T = int(offs[-1])
    d_off = device.DeviceArray.from_host(ctx, offs)
    d_packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
    d_mask = device.DeviceArray(ctx, device.mask_words(T), np.uint32)
    device.synth_ragged(ctx, seed, 0, n, d_off, T, d_packed, d_mask, gc_spread_permille=600, invalid_ppm=2000)
    d_counts = device.DeviceArray(ctx, (n, 256), np.uint32)
    d_scores = device.DeviceArray(ctx, n, np.float64)
    d_status = device.DeviceArray.from_host(ctx, np.zeros(1, np.uint32))
    out = {}
    for name in ("fast", "exact"):
        ctx.set_option("force_exact", "1" if name == "exact" else "0")
        res = {}
        for method in ("knn", "kmeans"):
            if name == "fast" and method == "knn":
                device.count_score(ctx, model, d_packed, d_mask, T, d_off, n, 4, method, d_counts, d_scores, d_status)
            else:
                device.score_counts(ctx, model, d_counts, n, method, d_scores, d_status)
            res[method] = d_scores.to_host()
        out[name] = res
        if name == "fast":
            print("  stats", ctx.score_stats_ex())
    ctx.set_option("force_exact", "0")
    ok = ~np.isnan(out["exact"]["knn"])
    a, b = out["fast"], out["exact"]
    assert np.array_equal(np.isnan(a["knn"]), ~ok)
    assert np.array_equal(a["knn"][ok], b["knn"][ok]), int((a["knn"][ok] != b["knn"][ok]).sum())
    rel = np.max(np.abs(a["kmeans"][ok] - b["kmeans"][ok]) / np.maximum(np.abs(b["kmeans"][ok]), 1e-300))
    assert rel < 1e-7, rel   # (two float64 summation orders; scores near zero amplify the relative measure)
    print("seed %d: %d contigs (%d zero-count), mean %d bases: votes identical, metric rel err %.1e" % (seed, n, int((~ok).sum()), T // n, rel))
    for x in (d_off, d_packed, d_mask, d_counts, d_scores, d_status):
        x.free()
print("soak OK")
