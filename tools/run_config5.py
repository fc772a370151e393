#!/usr/bin/env python3
"""BASELINE.json configs[4]: k=6 (4096-dim) scoring against a 50 000-row synthetic reference matrix
(rows = normalised 6-mer counts of seeded random 50 kb genomes, half labelled positive), N=131072
queries, one MI355X.  Reports the scoring time / rates of the general-D split-f16 MFMA path and checks
a sample of queries against the float64 brute-force GPU path (size-independent parity property)."""
import argparse, json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phamers_amd import _lib, device

ap = argparse.ArgumentParser()
ap.add_argument("--refs", type=int, default=50000)
ap.add_argument("--ref-length", type=int, default=50000)
ap.add_argument("--queries", type=int, default=131072)
ap.add_argument("--query-length", type=int, default=10000)
ap.add_argument("--k", type=int, default=6)
ap.add_argument("--sample", type=int, default=256)
ap.add_argument("--iters", type=int, default=2)
a = ap.parse_args()
ctx = _lib.Context(0)
k, D = a.k, 4 ** a.k


def device_counts(seed, n, L, chunk=8192):
    """k-mer counts of n seeded synthetic contigs, produced on the device in chunks."""
    outs = []
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        T = m * L
        packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
        off = device.DeviceArray(ctx, m + 1, np.uint64)
        device.synth_packed(ctx, seed, s, m, L, packed, off)
        cnt = device.DeviceArray(ctx, (m, D), np.uint32)
        device.count(ctx, packed, None, T, off, m, k, cnt)
        outs.append(cnt)
    return outs


t0 = time.time()
ref = np.concatenate([c.to_host() for c in device_counts(1000, a.refs, a.ref_length)]).astype(np.float64)
w = 1.0 + 0.25 * np.sin(np.arange(D) * 0.37)          # skew one class so that labels are learnable
ref[: a.refs // 2] *= w
ref /= ref.sum(axis=1, keepdims=True)
pos, neg = ref[: a.refs // 2], ref[a.refs // 2:]
cpos = np.stack([pos[i::86].mean(axis=0) for i in range(86)])
cneg = np.stack([neg[i::86].mean(axis=0) for i in range(86)])
t_ref = time.time() - t0
t0 = time.time()
model = _lib.Model(ctx, pos, neg, cpos, cneg, 3)
t_model = time.time() - t0

qparts = device_counts(2000, a.queries, a.query_length, chunk=a.queries)
d_q = qparts[0]
N = a.queries
d_scores = device.DeviceArray(ctx, N, np.float64)
d_status = device.DeviceArray(ctx, 1, np.uint32)
os.environ["PHK_FORCE_EXACT"] = "0"
device.score_counts(ctx, model, d_q, N, "combo", d_scores, d_status)   # warm-up (workspaces)
ctx.sync()
ctx.profile_reset(); ctx.profile_enable(True)
t0 = time.time()
for _ in range(a.iters):
    device.score_counts(ctx, model, d_q, N, "combo", d_scores, d_status)
ctx.sync()
wall = (time.time() - t0) / a.iters
prof = {kname: ms / cnt for kname, (ms, cnt) in ctx.profile().items()}
ctx.profile_enable(False)
n_fallback, n_exact = ctx.score_stats()
got = d_scores.to_host()

# parity on a sample through the float64 brute-force GPU path
ns = min(a.sample, N)
sample = device.DeviceArray.from_host(ctx, d_q.to_host()[:ns])
d_s2 = device.DeviceArray(ctx, ns, np.float64)
os.environ["PHK_FORCE_EXACT"] = "1"
device.score_counts(ctx, model, sample, ns, "combo", d_s2, d_status)
want = d_s2.to_host()
os.environ["PHK_FORCE_EXACT"] = "0"
M, C = a.refs, 172
flops = 2.0 * D * (M + C) * N
ms = prof.get("phk_knn_f16_general_kernel", 0.0)
print(json.dumps({
    "config": "k=%d (D=%d), %d-row reference (%d kb genomes), %d queries x %d bases, combo" % (k, D, a.refs, a.ref_length // 1000, N, a.query_length),
    "score_wall_ms": wall * 1e3, "kernel_ms": prof,
    "algorithmic_TFLOP": flops / 1e12,
    "proposal_kernel_TFLOPs_algorithmic": flops / (ms / 1e3) / 1e12 if ms else None,
    "proposal_kernel_frac_of_2.5PF_f16_peak": flops / (ms / 1e3) / 2.5e15 if ms else None,
    "proposal_kernel_mfma_issue_frac": (3 if os.environ.get("PHK_PROPOSAL", "cx").startswith("f1") else 2) * flops / (ms / 1e3) / 2.5e15 if ms else None,
    "queries_per_s": N / wall,
    "fallback_queries": n_fallback, "orderings_decided_by_exact_distances": n_exact,
    "sample_checked": ns, "sample_knn_votes_equal": bool(np.array_equal(np.sign(got[:ns]), np.sign(want))),
    "sample_max_rel_err_vs_f64_path": float(np.max(np.abs(got[:ns] - want) / np.abs(want))),
    "setup_s": {"reference_counts": t_ref, "model_build": t_model},
}))
