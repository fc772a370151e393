#!/usr/bin/env python3
"""Count-kernel microbenchmark (device-resident synthetic batch): ms per launch and algorithmic GB/s
(packed input + offsets + uint32 counts).  --lanes picks the count kernel (count_lanes): "" the library's choice, d / D / f the
one-window-per-add slot kernel (512 / 1024 threads, forced), p / P / q / Q the two-windows-per-add kernels, 0 the wave-per-contig
kernel; --invalid-ppm adds a validity mask."""
import argparse, json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phamers_amd import _lib, device

ap = argparse.ArgumentParser()
ap.add_argument("--contigs", type=int, default=1000000)
ap.add_argument("--length", type=int, default=5000)
ap.add_argument("--k", type=int, default=4)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--invalid-ppm", type=int, default=0)
ap.add_argument("--check", type=int, default=64)
ap.add_argument("--lanes", default="", help="count_lanes knob: '' default, p / P two-windows-per-add kernel (256x1 / 512x2), 2 slot kernel forced, 0 wave kernel")
a = ap.parse_args()
ctx = _lib.Context(0)
ctx.set_option("count_lanes", a.lanes)
n, L, k = a.contigs, a.length, a.k
T, D = n * L, 4 ** k
packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
mask = device.DeviceArray(ctx, device.mask_words(T), np.uint32) if a.invalid_ppm else None
off = device.DeviceArray(ctx, n + 1, np.uint64)
counts = device.DeviceArray(ctx, (n, D), np.uint32)
device.synth_packed(ctx, 0, 0, n, L, packed, off, mask, a.invalid_ppm)
device.count(ctx, packed, mask, T, off, n, k, counts)
ctx.sync()
ctx.profile_reset(); ctx.profile_enable(True)
for _ in range(a.iters):
    device.count(ctx, packed, mask, T, off, n, k, counts)
ctx.sync()
prof = ctx.profile()
ms = sum(prof[name][0] for name in ("phk_count_kernel", "phk_count_pairs_kernel", "phk_count_pairs2_kernel", "phk_count_direct_kernel") if name in prof) / a.iters
alg = n * ((L + 3) // 4 + 8 + 4 * D) + (n * ((L + 7) // 8) if mask else 0)
ok = None
if a.check:
    from oracle import oracle
    from phamers_amd import synth
    m = min(a.check, n)
    want = oracle.count(synth.synth_contigs(0, m, L, a.invalid_ppm), k).reshape(m, D)
    got = counts.to_host()[:m].astype(np.int64)
    ok = bool(np.array_equal(got, want))
print(json.dumps({"lanes": a.lanes, "invalid_ppm": a.invalid_ppm, "per_kernel_ms": {kn: v[0] / a.iters for kn, v in prof.items()}, "kernels": sorted(prof), "k": k, "contigs": n, "length": L,
                  "ms": ms, "GBps_algorithmic": alg / ms / 1e6, "Gbases_per_s": T / ms / 1e6,
                  "frac_hbm_peak": alg / ms / 1e6 / 8000.0, "bit_exact_vs_oracle": ok}))
