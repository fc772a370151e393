#!/usr/bin/env python3
"""
bench.py -- Gbases/s of the k-mer count + phage-score hot path on MI355X.

A "step" is one pass of the whole device-resident path over one synthetic batch:
2-bit packed contigs in HBM -> per-contig 4^k counts (materialised, uint32) -> normalise ->
3-NN vote + nearest-centroid proximity metric ("combo") -> float64 scores in HBM.
Workload at N=1 = BASELINE.json configs[1]: k=4, 1M x 5 kb contigs on one MI355X; for N>1
every rank processes its own 1M-contig shard (weak scaling, no data-path collective) and the
step ends with one RCCL all-gather of the score vectors.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
"roofline" (dominant kernel, algorithmic work / HIP-event time measured in this run) and
"cpu_baseline" (the CPU oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # fp32-input MFMA dense peak
F64_PEAK_TF = 78.6             # fp64 vector / matrix peak
MFMA_F16_PEAK_TF = 2500.0      # f16/bf16 dense MFMA peak (the count-exact kernel issues 2 MFMA flops per algorithmic flop, the split-query one 3)


def load_model_inputs(D):
    """Reference matrix + centroids: the real PhaMers 4-mer matrix (equalised, 2255 + 2255 rows)
    with the golden k-means centroids when the fixtures are present, else a seeded synthetic
    matrix of the same shape."""
    ref = os.path.join(REPO, "tests", "golden", "ref_features.npz")
    sco = os.path.join(REPO, "tests", "golden", "scoring_k4.npz")
    if D == 256 and os.path.exists(ref) and os.path.exists(sco):
        with np.load(ref) as z:
            pos = z["pos_counts"].astype(np.float64)
            neg = z["neg_counts"].astype(np.float64)
        pos /= pos.sum(axis=1, keepdims=True)
        neg /= neg.sum(axis=1, keepdims=True)
        n = min(len(pos), len(neg))
        with np.load(sco) as z:
            return pos[:n], neg[:n], z["cpos_eq"], z["cneg_eq"], "PhaMers reference_features (equalised)"
    rng = np.random.default_rng(1)
    pos = rng.gamma(2.0, 1.0, (2255, D))
    neg = rng.gamma(2.0, 1.0, (2255, D)) * np.linspace(0.7, 1.3, D)
    pos /= pos.sum(axis=1, keepdims=True)
    neg /= neg.sum(axis=1, keepdims=True)
    cpos = np.stack([pos[i::86].mean(axis=0) for i in range(86)])
    cneg = np.stack([neg[i::86].mean(axis=0) for i in range(86)])
    return pos, neg, cpos, cneg, "synthetic gamma rows"


def cpu_baseline(k, L, pos, neg, cpos, cneg, budget_s=12.0):
    """The oracle's literal window loop + NumPy normalise + brute k-NN + centroid loop on ONE host
    core, on a bounded sample of the same synthetic workload."""
    from oracle import oracle
    from phamers_amd import synth
    t0 = time.perf_counter()
    n = 0
    rows = []
    while True:
        rows.append(oracle.count_string_literal(synth.synth_contig(0, n, L), k))
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 4096:
            break
    counts = np.array(rows)
    q = oracle.normalize_counts(counts)
    oracle.score_points(q, pos, neg, "combo", 3, cpos, cneg)
    dt = time.perf_counter() - t0
    return {"value": n * L / dt / 1e9, "unit": "Gbases/s", "cores": 1, "kind": "port",
            "sample": "%d of the run's synthetic %d-base contigs: oracle literal window loop (kmer.py:47-50 "
                      "restated) + normalise + brute 3-NN + centroid loop, %.1f s" % (n, L, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--contigs", type=int, default=1000000, help="contigs per GPU")
    ap.add_argument("--length", type=int, default=5000)
    ap.add_argument("--k", type=int, default=4)
    ap.add_argument("--method", default="combo")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-contigs", type=int, default=256)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    ndev = torch.cuda.device_count()
    local_dev = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if args.gpus > 1 or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL over xGMI in production; PHK_BENCH_BACKEND=gloo only to rehearse the multi-rank code path
        # with several ranks sharing one GPU (RCCL refuses duplicate devices)
        backend = os.environ.get("PHK_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)

    from phamers_amd import _lib, device
    ctx = _lib.Context(local_dev, stream.cuda_stream)
    n, L, k = args.contigs, args.length, args.k
    D = 4 ** k
    T = n * L
    pos, neg, cpos, cneg, ref_name = load_model_inputs(D)
    model = _lib.Model(ctx, pos, neg, cpos, cneg, k_neighbors=3)
    M, C = pos.shape[0] + neg.shape[0], cpos.shape[0] + cneg.shape[0]

    packed = torch.empty(device.packed_words(T), dtype=torch.int32, device=dev)
    offsets = torch.empty(n + 1, dtype=torch.int64, device=dev)
    counts = torch.empty((n, D), dtype=torch.int32, device=dev)
    scores = torch.empty(n, dtype=torch.float64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    gathered = torch.empty(world * n, dtype=torch.float64, device=dev) if dist else None
    device.synth_packed(ctx, 0, rank * n, n, L, packed.data_ptr(), offsets.data_ptr())

    def step():
        device.count_score(ctx, model, packed.data_ptr(), None, T, offsets.data_ptr(), n, k, args.method,
                           counts.data_ptr(), scores.data_ptr(), status.data_ptr())
        if dist:
            if dist.get_backend() == "nccl":
                dist.all_gather_into_tensor(gathered, scores)   # the only collective: final score gather
            else:
                host = scores.cpu()
                parts = [torch.empty_like(host) for _ in range(world)]
                dist.all_gather(parts, host)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    ctx.profile_reset()
    ctx.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    prof = ctx.profile()
    n_fallback, n_exact = ctx.score_stats()

    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    # ---- parity spot check against the oracle (outside the timed region) ----
    from oracle import oracle
    from phamers_amd import synth
    npar = min(args.parity_contigs, n)
    seqs = synth.synth_contigs(0, npar, L)
    want_counts = oracle.count(seqs, k).reshape(npar, D)
    got_counts = counts[:npar].cpu().numpy().view(np.uint32).astype(np.int64)
    q = oracle.normalize_counts(want_counts)
    want = oracle.knn_score_points(q, pos, neg, 3) + oracle.centroid_score_points_fast(q, cpos, cneg) \
        if args.method == "combo" else oracle.score_points(q, pos, neg, args.method, 3, cpos, cneg)
    got = scores[:npar].cpu().numpy()
    parity = {"contigs_checked": npar, "counts_bit_exact": bool(np.array_equal(got_counts, want_counts)),
              "max_rel_score_err": float(np.max(np.abs(got - want) / np.abs(want))),
              "nan_rows": int(status.item()), "fallback_queries": n_fallback,
              "orderings_decided_by_exact_distances": n_exact}

    # ---- roofline of the dominant kernel (algorithmic work / HIP-event time in this run) ----
    alg = {  # kernel -> (bound, unit, peak, algorithmic work per step on this rank)
        "phk_count_kernel": ("hbm", "GB/s", HBM_PEAK_GBS, n * ((L + 3) // 4 + 8 + 4 * D + 8) / 1e9),
        "phk_count_slots_kernel": ("hbm", "GB/s", HBM_PEAK_GBS, n * ((L + 3) // 4 + 8 + 4 * D + 8) / 1e9),
        "phk_knn_mfma_kernel": ("mfma", "TFLOP/s", MFMA_F32_PEAK_TF, n * 2.0 * D * (M + C) / 1e12),
        "phk_knn_f16_kernel": ("mfma", "TFLOP/s", MFMA_F16_PEAK_TF, n * 2.0 * D * (M + C) / 1e12),
        "phk_knn_f16c_kernel": ("mfma", "TFLOP/s", MFMA_F16_PEAK_TF, n * 2.0 * D * (M + C) / 1e12),
        "phk_dist2_f64_kernel": ("mfma", "TFLOP/s", F64_PEAK_TF, n * 2.0 * D * (M + C) / 1e12),
    }
    if "phk_count_slots_kernel" in prof:   # the wave-per-contig kernel then only serves the hand-over list
        alg.pop("phk_count_kernel", None)
    kernels = {}
    for name, (ms, launches) in prof.items():
        kernels[name] = {"ms_per_step": ms / args.steps, "launches_per_step": launches / args.steps}
        if name in alg and ms > 0:
            bound, unit, peak, work = alg[name]
            ach = work * args.steps / (ms / 1e3)
            kernels[name].update({"bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak})
    dom = max((kname for kname in kernels if kname in alg), key=lambda kname: kernels[kname]["ms_per_step"])
    # HBM traffic of the dominant kernel: PMC measurement of this same command (FETCH_SIZE / WRITE_SIZE
    # in separate rocprofv3 passes, corrected as MI355X_MICROARCH.md prescribes), stored per contig
    traffic = None
    tfile = os.path.join(REPO, "profiles", "r01", "traffic.json")
    if os.path.exists(tfile):
        tk = json.load(open(tfile)).get("kernels", {}).get(dom)
        if tk:
            traffic = tk["hbm_bytes_per_contig"] * n
    roofline = {"kernel": dom, "bound": kernels[dom]["bound"], "achieved": kernels[dom]["achieved"],
                "peak": kernels[dom]["peak"], "unit": kernels[dom]["unit"], "frac": kernels[dom]["frac"],
                "traffic": traffic}
    if dom == "phk_knn_f16_kernel":
        # the split-f16 kernel issues 3 MFMA flops per algorithmic flop (hi.hi + hi.lo + lo.hi)
        roofline["mfma_issue_frac"] = 3.0 * kernels[dom]["frac"]
    if dom == "phk_knn_f16c_kernel":
        # the count-exact kernel issues 2 MFMA flops per algorithmic flop (c.r_hi + c.r_lo)
        roofline["mfma_issue_frac"] = 2.0 * kernels[dom]["frac"]

    out = {
        "metric": "Gbases/s k-mer-count+score, k=%d, %d kb contigs" % (k, L // 1000),
        "value": world * n * L * args.steps / elapsed / 1e9,
        "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32 counts + f16 MFMA proposal (exact integer counts x split-f16 reference, f32 accumulate) + f64 decision",
        "data": "synthetic (seeded uniform ATGC contigs, device-generated); reference matrix: " + ref_name,
        "config": {"workload": "k=%d, %d x %d-base contigs per GPU, count+normalise+%s score, "
                               "%d reference rows + %d centroids" % (k, n, L, args.method, M, C),
                   "contigs_per_gpu": n, "contig_length": L, "k": k, "method": args.method,
                   "parallelism": "contig shards, %d rank(s), final all-gather of scores" % world},
        "roofline": roofline,
        "kernels": kernels,
        "parity": parity,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(k, L, pos, neg, cpos, cneg)
    print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
