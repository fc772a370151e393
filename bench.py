#!/usr/bin/env python3
"""
bench.py -- Gbases/s of the k-mer count + phage-score hot path on MI355X.

A "step" is one pass of the whole device-resident path over one synthetic batch:
2-bit packed contigs in HBM -> per-contig 4^k counts (materialised, uint32) -> normalise ->
3-NN vote + nearest-centroid proximity metric ("combo") -> float64 scores in HBM.

Workload (--config, phamers_amd/workloads.py; BASELINE.json `configs`):
  1 (default, the configuration the metric is quoted on)  k=4, 1M x 5 kb contigs per GPU, real PhaMers matrix
  2  k=5, 10M x 10 kb contigs per GPU, 2255+2255 synthetic reference genomes
  3  k=4, 100M x 5 kb contigs split over the ranks (12.5M per GPU at 8), real PhaMers matrix
  4  k=6, 131072 x 10 kb queries per GPU against 50 000 synthetic reference genomes (replicated on every rank)
For N>1 every rank processes its own shard of contigs / queries against a replicated reference (no data-path
collective) and every step issues one RCCL all-gather of the score vectors; the gather is asynchronous (it travels while
the next step computes, two score buffers) and all of them have completed when the timed region closes.
`python bench.py --gpus N` started plainly launches its own N ranks (one child process per GPU); under
torch.distributed.run it is one of the ranks.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
"roofline" (dominant kernel, algorithmic work / HIP-event time measured in this run) and
"cpu_baseline" (the CPU oracle timed on this box's host cores on a bounded sample; 1 core, with the
all-core vectorised variant beside it under "all_cores").
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F64_PEAK_TF = 78.6             # fp64 vector / matrix peak
MFMA_F16_PEAK_TF = 2500.0      # f16/bf16 dense MFMA peak
MFMA_I8_PEAK_TF = 5000.0       # int8 dense MFMA peak (2 x bf16 per clock: MI355X_MICROARCH.md, MFMA table)


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


_G = {}   # per-process state of the all-core baseline's pool workers


def _make_scorer(pos, neg, cpos, cneg):
    """The reference's own scoring route on the host: scikit-learn brute 3-NN (scripts/learning.py:127-128) + the
    nearest-centroid proximity metric; fitted once."""
    from oracle import oracle
    try:
        from sklearn.neighbors import KNeighborsClassifier
        clf = KNeighborsClassifier(n_neighbors=3, algorithm="brute").fit(
            np.vstack((pos, neg)), np.append(np.ones(len(pos)), np.zeros(len(neg))))
        knn = lambda q: 2 * (clf.predict(q) - 0.5)   # noqa: E731
    except ImportError:
        knn = lambda q: oracle.knn_score_points(q, pos, neg, 3)   # noqa: E731
    return lambda q: knn(q) + oracle.centroid_score_points_fast(q, cpos, cneg)


def _pool_worker_init():
    try:   # one BLAS / OpenMP thread per worker process: the pool is the parallelism
        from threadpoolctl import threadpool_limits
        _G["limits"] = threadpool_limits(limits=1)
    except ImportError:
        pass


def _pool_task(blob):
    """One unit of the all-core baseline, entirely inside one worker: vectorised count of the shared sample ->
    normalise -> score.  A worker's first task loads the sample and the model inputs from `blob` (an .npz written by
    the parent) and fits the classifier."""
    from oracle import oracle
    if blob is not None and _G.get("blob") != blob:
        with np.load(blob) as z:
            _G.update(blob=blob, seqs=[str(x) for x in z["seqs"]], k=int(z["k"]),
                      score=_make_scorer(z["pos"], z["neg"], z["cpos"], z["cneg"]))
        _pool_worker_init()   # again, now that scikit-learn's OpenMP runtime is loaded in this process
    seqs, k = _G["seqs"], _G["k"]
    q = oracle.normalize_counts(oracle.count(list(seqs), k).reshape(len(seqs), -1))
    return float(np.sum(_G["score"](q)))


def cpu_baseline(k, L, pos, neg, cpos, cneg, budget_s=12.0, pool=None):
    """(i) 1 core: the oracle's literal window loop (scripts/kmer.py:47-50 restated) + normalise + brute 3-NN +
    centroid metric; (ii) all cores: a process per core, each running the oracle's vectorised counter + the same
    scoring on its share.  Both on bounded samples of the run's synthetic contigs."""
    from oracle import oracle
    from phamers_amd import synth
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        import contextlib
        threadpool_limits = lambda limits: contextlib.nullcontext()   # noqa: E731
    cores = os.cpu_count() or 1
    score = _make_scorer(pos, neg, cpos, cneg)
    nmax = max(64, int(4096 * 5000 / L))
    seqs = synth.synth_contigs(0, nmax, L)           # sample preparation is not timed
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        rows = []
        for s in seqs:
            rows.append(oracle.count_string_literal(s, k))
            if time.perf_counter() - t0 > budget_s * 0.8:
                break
        n = len(rows)
        score(oracle.normalize_counts(np.array(rows)))
        dt = time.perf_counter() - t0
    out = {"value": n * L / dt / 1e9, "unit": "Gbases/s", "cores": 1, "kind": "port",
           "sample": "%d of the run's synthetic %d-base contigs: oracle literal window loop (kmer.py:47-50 restated) "
                     "+ normalise + scikit-learn brute 3-NN + centroid metric, %.1f s" % (n, L, dt),
           "cpu": cpu_model_name(), "host_cores": cores}
    # (ii) every core of this job's CPU share.  The pool was forked BEFORE this process touched the GPU (its workers
    # hold no device state); the sample and the model inputs reach the workers through a file.
    if pool is None:
        return out
    import tempfile
    blob = None
    try:
        per_task = max(8, min(256, nmax))
        workers = pool._processes
        if (pos.nbytes + neg.nbytes) > (256 << 20):
            raise RuntimeError("reference matrix too large for a per-process copy in %d workers" % workers)
        _G.update(seqs=seqs[:per_task], k=k, score=score)
        with threadpool_limits(limits=1):
            t1 = time.perf_counter()
            _pool_task(None)
            one = time.perf_counter() - t1
        fd, blob = tempfile.mkstemp(suffix=".npz", prefix="phk_cpu_baseline_")
        os.close(fd)
        np.savez(blob, seqs=np.array(seqs[:per_task]), k=k, pos=pos, neg=neg, cpos=cpos, cneg=cneg)
        ntasks = max(workers, int(budget_s / max(one, 1e-3)) * workers)
        pool.map(_pool_task, [blob] * (4 * workers), chunksize=1)       # workers load + fit outside the timed region
        t1 = time.perf_counter()
        pool.map(_pool_task, [blob] * ntasks, chunksize=1)
        dta = time.perf_counter() - t1
        out["all_cores"] = {"value": ntasks * per_task * L / dta / 1e9, "unit": "Gbases/s", "cores": workers,
                            "kind": "port",
                            "sample": "%d contigs (%d tasks of %d) on a %d-process pool, each process single-threaded: "
                                      "oracle vectorised counter + normalise + scikit-learn brute 3-NN + centroid "
                                      "metric, %.1f s" % (ntasks * per_task, ntasks, per_task, workers, dta)}
    except Exception as e:   # noqa: BLE001 -- the 1-core baseline is the contract; this one is best effort
        out["all_cores"] = {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        if blob and os.path.exists(blob):
            os.unlink(blob)
    return out


def latest_traffic(kernel, config=1):
    """HBM bytes per contig of `kernel` from the newest profiles/r*/traffic*.json of this configuration that has it
    (PMC measurement of this same command, FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes, corrected as
    MI355X_MICROARCH.md prescribes).  Returns (bytes per contig, source file, {kernel: bytes per contig} of that
    file)."""
    name = "traffic.json" if config == 1 else "traffic_config%d.json" % config
    for tfile in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", name)), reverse=True):
        try:
            ks = json.load(open(tfile)).get("kernels", {})
        except (OSError, ValueError):
            continue
        tk = ks.get(kernel)
        if tk:
            per = {kn: v["hbm_bytes_per_contig"] for kn, v in ks.items() if "synth" not in kn}
            return tk["hbm_bytes_per_contig"], os.path.relpath(tfile, REPO), per
    return None, None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4], help="BASELINE.json workload (see module doc)")
    ap.add_argument("--contigs", type=int, default=None, help="contigs per GPU (overrides the configuration)")
    ap.add_argument("--length", type=int, default=None)
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--refs", type=int, default=None, help="synthetic reference genomes (configs 2, 4)")
    ap.add_argument("--method", default="combo")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-contigs", type=int, default=None)
    ap.add_argument("--workload", default="uniform", choices=["uniform", "ragged"],
                    help="uniform: the BASELINE configuration (fixed-length, uniform i.i.d. contigs; the headline). "
                         "ragged: heavy-tailed lengths 5-500 kb in arbitrary order, per-contig GC 0.3-0.7, 0.1 %% N -- "
                         "a second, separately labelled workload; --contigs defaults to 200000 there")
    ap.add_argument("--min-seconds", type=float, default=8.0,
                    help="time at least this long: the timed region is max(--steps, enough steps to fill it), so that a "
                         "millisecond-scale step is measured at the sustained clock (and is visible to an outside "
                         "GPU-activity sampler); 0 = exactly --steps")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, form the process group, all-gather the rank ids and print the line's "
                         "n_gpus / ranks_seen; no GPU work (what the CPU test of the launcher runs, over gloo)")
    args = ap.parse_args()

    # `python bench.py --gpus N` started plainly (no RANK in the environment): this process becomes the launcher.  It
    # starts the N ranks as CHILD processes before anything here touches the GPU (the devices are counted in sysfs, not through the runtime), hands
    # through rank 0's JSON line and exits with the ranks' status.  Under torch.distributed.run RANK is set and this
    # branch is not taken.  With fewer than N devices visible nothing is measured (exit code 2); PHK_BENCH_BACKEND=gloo
    # lifts that check to rehearse the multi-rank code path with several ranks on one device (or, with
    # --rendezvous-only, on none).
    backend = os.environ.get("PHK_BENCH_BACKEND", "nccl")
    if args.gpus > 1 and "RANK" not in os.environ:
        from phamers_amd import dist as pdist
        sys.exit(pdist.launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                    require_gpus=(backend == "nccl")))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_env != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world_env))
        sys.exit(2)
    if args.rendezvous_only:
        import torch.distributed as tdist
        from phamers_amd import dist as pdist
        rank = int(os.environ.get("RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        tdist.init_process_group("gloo", rank=rank, world_size=world_env)
        seen = pdist.ranks_seen()
        tdist.barrier()
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": world_env, "ranks_seen": seen, "backend": "gloo",
                              "launched_by": os.environ.get("PHK_LAUNCHED_BY", "torch.distributed.run")}))
        tdist.destroy_process_group()
        return

    # worker processes of the all-core CPU baseline: forked now, before anything initialises the GPU in this process
    pool = None
    if world_env == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        import multiprocessing as mp
        # the job's CPU share (a 1-GPU box allots 16 of the host's cores), not the host's core count
        try:
            share = len(os.sched_getaffinity(0))
        except AttributeError:
            share = os.cpu_count() or 1
        pool = mp.get_context("fork").Pool(max(1, min(share, 16)), initializer=_pool_worker_init)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    ndev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if backend == "nccl" and ndev < local_world:
        sys.stderr.write("bench.py: %d rank(s) on this node but %d GPU(s) visible -- not measuring\n" % (local_world, ndev))
        sys.exit(2)
    local_dev = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if args.gpus > 1 or world > 1 or os.environ.get("PHK_BENCH_FORCE_DIST") == "1":   # (the last: a 1-rank group, to run
        import torch.distributed as dist                                                 #  the collective's code path on one GPU)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL over xGMI in production; PHK_BENCH_BACKEND=gloo only to rehearse the multi-rank code path
        # with several ranks sharing one GPU (RCCL refuses duplicate devices)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    seen = 1
    if dist:
        from phamers_amd import dist as pdist
        seen = pdist.ranks_seen(device=dev if backend == "nccl" else None)
        if seen != world:
            sys.stderr.write("bench.py: the all-gather of rank ids saw %d distinct rank(s), world size %d\n" % (seen, world))
            sys.exit(3)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)

    from phamers_amd import _lib, device, workloads
    ctx = _lib.Context(local_dev, stream.cuda_stream)
    cfg = dict(workloads.CONFIGS[args.config])
    if args.k is not None and args.k != cfg["k"]:
        cfg.update(k=args.k, reference="synthetic", refs=args.refs or 4510, ref_length=50000)
    k = cfg["k"]
    L = args.length or cfg["length"]
    ragged = args.workload == "ragged"
    if args.contigs is not None:
        n = args.contigs
    elif ragged:
        n = 200000
    elif "contigs_total" in cfg:
        n = cfg["contigs_total"] // world          # config 3: a fixed total, split over the ranks
    else:
        n = cfg["contigs"]
    scaling = "strong" if ("contigs_total" in cfg and args.contigs is None) else "weak"
    D = 4 ** k
    lengths = None
    if ragged:
        from phamers_amd import synth as _synth
        lengths = _synth.ragged_lengths(1000 + rank, n)
        T = int(lengths.sum())
    else:
        T = n * L
    pos, neg, cpos, cneg, ref_name = workloads.reference_for(ctx, cfg, args.refs)
    model = _lib.Model(ctx, pos, neg, cpos, cneg, k_neighbors=3)
    M, C = pos.shape[0] + neg.shape[0], cpos.shape[0] + cneg.shape[0]

    packed = torch.empty(device.packed_words(T), dtype=torch.int32, device=dev)
    offsets = torch.empty(n + 1, dtype=torch.int64, device=dev)
    counts = torch.empty((n, D), dtype=torch.int32, device=dev)
    scores_buf = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(2)]
    scores = scores_buf[0]
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    gathered_buf = [torch.full((world * n,), float("nan"), dtype=torch.float64, device=dev) for _ in range(2)] if dist else None
    first = rank * n
    mask = None
    if ragged:
        offs = np.zeros(n + 1, dtype=np.int64)
        offs[1:] = np.cumsum(lengths)
        offsets.copy_(torch.from_numpy(offs))
        mask = torch.empty(device.mask_words(T), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ragged_ppm = int(os.environ.get("PHK_BENCH_RAGGED_PPM", "1000"))   # (diagnostic runs: 0 = the ragged batch without invalid bases)
        device.synth_ragged(ctx, 0, first, n, offsets.data_ptr(), T, packed.data_ptr(), mask.data_ptr(),
                            gc_spread_permille=400, invalid_ppm=ragged_ppm)
        if ragged_ppm == 0:
            mask = None
    else:
        device.synth_packed(ctx, 0, first, n, L, packed.data_ptr(), offsets.data_ptr())
    mask_ptr = mask.data_ptr() if mask is not None else None

    # The only collective of the path: the final gather of the score vectors (RCCL all-gather over xGMI).  It is issued
    # asynchronously, so the gather of step i travels while step i + 1 computes (two score / gather buffers; a buffer's
    # pending gather is waited for before the buffer is written again), and every gather has completed before the timed
    # region closes.
    pending = [None, None]
    nstep = [0]

    def drain():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    def step():
        b = nstep[0] & 1
        nstep[0] += 1
        if pending[b] is not None:
            pending[b].wait()       # (stream-level: the compute stream waits for the gather that still reads scores_buf[b])
            pending[b] = None
        device.count_score(ctx, model, packed.data_ptr(), mask_ptr, T, offsets.data_ptr(), n, k, args.method,
                           counts.data_ptr(), scores_buf[b].data_ptr(), status.data_ptr())
        if dist:
            if dist.get_backend() == "nccl":
                pending[b] = dist.all_gather_into_tensor(gathered_buf[b], scores_buf[b], async_op=True)
            else:
                host = scores_buf[b].cpu()
                parts = [torch.empty_like(host) for _ in range(world)]
                dist.all_gather(parts, host)

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    ctx.profile_reset()
    ctx.profile_enable(True)
    steps = args.steps
    if args.min_seconds > 0:
        # sustained run: size the timed region from one probe step so that every rank times the same number of steps
        t0 = time.perf_counter()
        step()
        drain()
        torch.cuda.synchronize()
        one = time.perf_counter() - t0
        if dist:
            tmax = torch.tensor([one], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            one = float(tmax.item())
        steps = max(steps, int(args.min_seconds / max(one, 1e-6)) + 1)
        ctx.profile_reset()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    drain()
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
    stats_ex = ctx.score_stats_ex()
    # the collective, checked after the timed region: this rank's slice of the last step's gathered vector is its score
    # vector bit for bit, and every rank's slice arrived (a score is never NaN for these synthetic contigs; the gather
    # buffers start out as NaN)
    gather_info = None
    if dist:
        gather_info = {"backend": dist.get_backend(), "async": dist.get_backend() == "nccl",
                       "bytes_per_rank_per_step": n * 8}
        if gathered_buf is not None and dist.get_backend() == "nccl":
            b = (nstep[0] - 1) & 1
            mine = gathered_buf[b][rank * n:(rank + 1) * n]
            gather_info["gathered_equals_scores"] = bool(torch.equal(mine, scores_buf[b]))
            gather_info["slices_arrived"] = int(sum(
                bool(torch.isfinite(gathered_buf[b][r * n:(r + 1) * n]).all().item()) for r in range(world)))
    scores = scores_buf[(nstep[0] - 1) & 1]

    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    # ---- parity spot check against the oracle (outside the timed region) ----
    from oracle import oracle
    from phamers_amd import synth
    npar = args.parity_contigs if args.parity_contigs is not None else (256 if D * M <= 256 * 5000 else 24)
    npar = min(npar, n)
    if ragged:
        npar = min(npar, 64)
    # the sample: contigs drawn at random over the rank's whole range (seeded; always with the first and the last one), not
    # its first tile -- every scoring batch, tile position and workgroup of the step has a chance of being looked at
    if npar >= n:
        sample = np.arange(n)
    else:
        sample = np.unique(np.concatenate(([0, n - 1], np.random.default_rng(20241005 + rank).choice(n, max(npar - 2, 0), replace=False))))
        npar = len(sample)
    if ragged:
        seqs = [synth.synth_ragged_contig(0, first + int(c), int(lengths[int(c)]), 400, int(os.environ.get("PHK_BENCH_RAGGED_PPM", "1000"))) for c in sample]
    else:
        seqs = [synth.synth_contig(0, first + int(c), L) for c in sample]
    want_counts = oracle.count(seqs, k).reshape(npar, D)
    sample_t = torch.from_numpy(sample.astype(np.int64)).to(counts.device)
    got_counts = counts[sample_t].cpu().numpy().view(np.uint32).astype(np.int64)
    q = oracle.normalize_counts(want_counts)
    want = oracle.knn_score_points(q, pos, neg, 3) + oracle.centroid_score_points_fast(q, cpos, cneg) \
        if args.method == "combo" else oracle.score_points(q, pos, neg, args.method, 3, cpos, cneg)
    got = scores[sample_t].cpu().numpy()
    parity = {"contigs_checked": npar, "sample": "random over the rank's contigs (seeded), first and last included",
              "counts_bit_exact": bool(np.array_equal(got_counts, want_counts)),
              "max_rel_score_err": float(np.max(np.abs(got - want) / np.abs(want))),
              "nan_rows": int(status.item()), "fallback_queries": n_fallback,
              "orderings_decided_by_exact_distances": n_exact, "decision_stats": stats_ex}

    # ---- roofline of the dominant kernel (algorithmic work / HIP-event time in this run) ----
    # packed + offset + counts + score (+ the validity mask when one is supplied)
    count_bytes = ((T + 3) // 4 + (T // 8 if ragged else 0) + n * (8 + 4 * D + 8)) / 1e9
    score_tflop = n * 2.0 * D * (M + C) / 1e12                     # 2 D (M + C) per contig
    alg = {  # kernel -> (bound, unit, peak, algorithmic work per step on this rank)
        "phk_count_kernel": ("hbm", "GB/s", HBM_PEAK_GBS, count_bytes),
        "phk_count_pairs_kernel": ("hbm", "GB/s", HBM_PEAK_GBS, count_bytes),
        "phk_count_direct_kernel": ("hbm", "GB/s", HBM_PEAK_GBS, count_bytes),
        "phk_knn_f16_kernel": ("mfma", "TFLOP/s", MFMA_F16_PEAK_TF, score_tflop),
        "phk_knn_f16h_kernel": ("mfma", "TFLOP/s", MFMA_F16_PEAK_TF, score_tflop),
        "phk_knn_f16_general_kernel": ("mfma", "TFLOP/s", MFMA_F16_PEAK_TF, score_tflop),
        "phk_knn_i8_general_kernel": ("mfma", "TFLOP/s", MFMA_I8_PEAK_TF, score_tflop),
        "phk_dist2_f64_kernel": ("mfma", "TFLOP/s", F64_PEAK_TF, score_tflop),
    }
    if "phk_count_pairs_kernel" in prof or "phk_count_direct_kernel" in prof:
        alg.pop("phk_count_kernel", None)   # the wave-per-contig kernel then only serves the hand-over list
    slotk = [kname for kname in ("phk_count_pairs_kernel", "phk_count_direct_kernel") if kname in prof]
    if len(slotk) > 1:
        # two of them are launched and decide on the device which one counts the batch (the other returns at once):
        # the bytes are credited to the one that did the work
        busy = max(slotk, key=lambda kname: prof[kname][0])
        for kname in slotk:
            if kname != busy:
                alg.pop(kname, None)
    # the split-query kernel is the whole sweep only as a first pass (float64 rows / proposal=f16); as the second
    # chance of a count-exact first pass it sees the queued rows alone: credit it with those (stats_ex[2] is the
    # last step's queue on this rank), never with the batch
    first_pass = [kname for kname in ("phk_knn_f16h_kernel", "phk_knn_f16_general_kernel", "phk_knn_i8_general_kernel") if kname in prof]
    if first_pass and "phk_knn_f16_kernel" in prof:
        bound, unit, peak, _ = alg["phk_knn_f16_kernel"]
        alg["phk_knn_f16_kernel"] = (bound, unit, peak, score_tflop * (stats_ex["second_chance"] / float(n) if n else 0.0))
    kernels = {}
    for name, (ms, launches) in prof.items():
        kernels[name] = {"ms_per_step": ms / steps, "launches_per_step": launches / steps}
        if name in alg and ms > 0:
            bound, unit, peak, work = alg[name]
            ach = work * steps / (ms / 1e3)
            kernels[name].update({"bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak})
    # (the second chance is never the dominant kernel: in a multi-batch call it runs on the library's second stream beside the
    # next batch's sweep, and its event time then spans that sweep)
    second = {"phk_knn_f16_kernel"} if first_pass else set()
    dom = max((kname for kname in kernels if kname in alg and kname not in second), key=lambda kname: kernels[kname]["ms_per_step"])
    as_configured = not ragged and L == cfg["length"] and k == cfg["k"] and args.refs is None
    per_contig, tsrc, per_all = latest_traffic(dom, args.config) if as_configured else (None, None, None)
    roofline = {"kernel": dom, "bound": kernels[dom]["bound"], "achieved": kernels[dom]["achieved"],
                "peak": kernels[dom]["peak"], "unit": kernels[dom]["unit"], "frac": kernels[dom]["frac"],
                "traffic": per_contig * n if per_contig else None}
    if tsrc:
        # the PMC passes are separate rocprofv3 runs of this same command (counters cannot be read from inside it):
        # the figure is quoted from the newest committed measurement, not measured by the run that prints this line
        roofline["traffic_source"] = tsrc
        roofline["traffic_measured_in_this_run"] = False
    # the whole step against both roofs: every algorithmic flop / byte of the step over the step's wall time, and the
    # HBM bytes all of its kernels moved (PMC, same source) beside the algorithmic bytes
    step_s = elapsed / steps
    roofline["whole_step"] = {
        "mfma_frac": score_tflop / step_s / MFMA_F16_PEAK_TF, "hbm_frac": count_bytes / step_s / HBM_PEAK_GBS,
        "algorithmic_bytes_per_contig": count_bytes * 1e9 / n,
        "hbm_bytes_per_contig_all_kernels": round(sum(per_all.values()), 1) if per_all else None}
    # MFMA flops ISSUED per algorithmic flop: 3 (split-query f16: hi.hi + hi.lo + lo.hi), 2 (count-exact: c.r_hi +
    # c.r_lo at general D), 17/16 (k = 4, high parts only: the low parts are applied to the few candidates by the decision stage;
    # the 17th step carries the bias),
    # int8 (against the int8 peak): the reference column is a 24-bit fixed-point value in three int8 parts; the default sweep
    # issues the upper two (the third is applied to the window's candidates by the decision stage), proposal=i83 all three
    i8_parts = 3.0 if os.environ.get("PHK_PROPOSAL", "") == "i83" else 2.0
    issue = {"phk_knn_f16_kernel": 3.0, "phk_knn_f16h_kernel": 17.0 / 16.0,
             "phk_knn_f16_general_kernel": 2.0, "phk_knn_i8_general_kernel": i8_parts}.get(dom)
    if issue:
        roofline["mfma_issue_frac"] = issue * kernels[dom]["frac"]

    out = {
        "metric": ("Gbases/s k-mer-count+score, k=%d, ragged 5-500 kb contigs" % k) if ragged else
                  "Gbases/s k-mer-count+score, k=%d, %d kb contigs" % (k, L // 1000),
        "value": world * T * steps / elapsed / 1e9,
        "unit": "Gbases/s", "n_gpus": world, "ranks_seen": seen, "steps": steps, "steps_requested": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None,
        "dtype": ("u32 counts + int8 MFMA proposal (exact integer counts x 24-bit fixed-point reference columns in three int8 "
                  "parts, exact int32 accumulate; the upper two parts in the sweep, the third added as an exact integer product "
                  "to the window's candidates) + f64 decision") if "phk_knn_i8_general_kernel" in prof else
                 "u32 counts + f16 MFMA proposal (exact integer counts x fp16 reference parts, f32 accumulate; k=4: high parts in the sweep, low parts added in f64 to the window's candidates) + f64 decision",
        "data": ("synthetic (seeded, device-generated: heavy-tailed lengths 5-500 kb in arbitrary order, per-contig GC "
                 "0.3-0.7, 0.1 % invalid bases); reference matrix: " if ragged else
                 "synthetic (seeded uniform ATGC contigs, device-generated); reference matrix: ") + ref_name,
        "config": {"workload": ("RAGGED (not the BASELINE configuration): k=%d, %d contigs per GPU, %.2f Gbases, mean %d / max %d "
                                "bases, validity mask, count+normalise+%s score, %d reference rows + %d centroids"
                                % (k, n, T / 1e9, T // n, int(lengths.max()), args.method, M, C)) if ragged else
                               "BASELINE configs[%d]: k=%d, %d x %d-base contigs per GPU, count+normalise+%s score, "
                               "%d reference rows + %d centroids" % (args.config, k, n, L, args.method, M, C),
                   "contigs_per_gpu": n, "contig_length": L, "k": k, "method": args.method,
                   "parallelism": "contig shards, %d rank(s), replicated reference, final all-gather of scores" % world},
        "roofline": roofline,
        "kernels": kernels,
        # per-kernel times are HIP events around each launch on the stream it runs on: in a call of several scoring batches
        # (more than 2^20 contigs at k = 4) a batch's tail runs on a second stream beside the next batch's sweep, so those
        # kernels' times overlap the sweep's and the column does not add up to the step
        "kernel_times_overlap": bool(k == 4 and n > (1 << 20)),
        "parity": parity,
        "timed_region_s": elapsed,
    }
    if gather_info:
        out["gather"] = gather_info
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(k, L if not ragged else int(T // n), pos, neg, cpos, cneg, pool=pool)
    if pool is not None:
        pool.close()
        pool.join()
    print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
