"""
dist.py -- contig sharding across the GPUs of one node and the single collective of the path:
the final gather of the per-rank score vectors (RCCL all-gather over xGMI; gloo on CPU for tests).

Contigs are independent (scripts/kmer.py:102-105, scripts/phamer.py:251-255), so the batch is cut
into contiguous contig ranges balanced in bases, every rank counts + scores its own range against
a replicated reference matrix, and nothing is exchanged until the scores are gathered.  One
process per GPU, launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE) or by
launch_ranks() below, which starts the rank processes itself.
"""
import os
import socket
import subprocess
import sys

import numpy as np


def _parse_visible(value, have):
    """Entries of a *_VISIBLE_DEVICES list that name one of `have` devices by index (the runtime stops at the first
    entry it cannot resolve; UUID entries are taken as present)."""
    n = 0
    for item in value.split(","):
        item = item.strip()
        if not item:
            break
        if item.lstrip("-").isdigit():
            if not 0 <= int(item) < have:
                break
        n += 1
    return n


def visible_gpus(topology="/sys/class/kfd/kfd/topology/nodes"):
    """Number of GPUs this process may use, counted WITHOUT the HIP / HSA runtime: the KFD topology in sysfs lists
    one node per agent, and a GPU is a node with a non-zero simd_count (CPU nodes have 0); ROCR_VISIBLE_DEVICES /
    HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES narrow the count as they would narrow the runtime's.  A launcher may
    therefore call this and still start its ranks as children that are the first to touch the GPU."""
    import glob
    have = 0
    for props in glob.glob(os.path.join(topology, "*", "properties")):
        try:
            with open(props) as f:
                for line in f:
                    if line.startswith("simd_count"):
                        have += int(line.split()[1]) > 0
                        break
        except (OSError, ValueError, IndexError):
            continue
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            have = min(have, _parse_visible(v, have))
    return have


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n_ranks, argv, require_gpus=True, timeout=None, extra_env=None):
    """Start ``n_ranks`` copies of the command ``argv`` as CHILD processes, one per GPU, with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (the variables torch.distributed.run would set), forward rank 0's
    standard output and wait for all of them.  The caller must not have initialised the GPU: the ranks are children
    of a process that holds no device state, and nothing is exec'ed over a process that does.  Returns the first
    non-zero exit code of a rank, or 0; when a rank fails the others are terminated.  ``require_gpus``: refuse
    (exit code 2, nothing started) when fewer than ``n_ranks`` devices are visible, instead of putting several ranks
    on one device."""
    n_ranks = int(n_ranks)
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    if require_gpus:
        have = visible_gpus()
        if have < n_ranks:
            sys.stderr.write("launch_ranks: %d rank(s) asked for, %d GPU(s) visible -- not started\n" % (n_ranks, have))
            return 2
    port = free_port()
    procs = []
    for rank in range(n_ranks):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n_ranks),
                   LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PHK_LAUNCHED_BY="launch_ranks")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_ranks)))
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=None if rank == 0 else subprocess.DEVNULL))
    import time
    t0 = time.monotonic()
    rc = 0
    live = set(range(n_ranks))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                sys.stderr.write("launch_ranks: rank %d exited with code %d; stopping the others\n" % (r, code))
                for o in live:
                    procs[o].terminate()
        if live:
            if timeout is not None and time.monotonic() - t0 > timeout:
                sys.stderr.write("launch_ranks: timeout after %.0f s; stopping %d rank(s)\n" % (timeout, len(live)))
                for o in live:
                    procs[o].kill()
                rc = rc or 124
                timeout = None
            time.sleep(0.05)
    return rc


def ranks_seen(group=None, device=None):
    """How many distinct ranks answer an all-gather of the rank ids (the launcher's self-check: equals the world
    size when every rank is its own process in one group)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    mine = torch.tensor([dist.get_rank(group)], dtype=torch.int64, device=device or "cpu")
    out = torch.empty(world, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return int(torch.unique(out.cpu()).numel())


def shard_bounds(lengths, world_size):
    """Contiguous contig ranges [lo, hi) per rank, balanced on the prefix sum of bases.
    Every contig belongs to exactly one rank; ranks may get empty ranges when there are fewer
    contigs than ranks."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = lengths.shape[0]
    world_size = int(world_size)
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    csum = np.concatenate(([0], np.cumsum(lengths)))
    total = int(csum[-1])
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        # first contig boundary at or after the target, never before the previous bound
        cut = int(np.searchsorted(csum, target, side="left"))
        cut = min(max(cut, bounds[-1]), n)
        bounds.append(cut)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def gather_variable(local, group=None):
    """All-gather of 1-D float64 tensors whose lengths differ per rank -> concatenation in rank
    order on every rank.  One size exchange + one padded all-gather."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n_local = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    width = max(max(sizes), 1)
    padded = torch.zeros(width, dtype=local.dtype, device=local.device)
    padded[: local.numel()] = local
    out = torch.empty(world * width, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * width: r * width + sizes[r]] for r in range(world)])


def gather_rows_to_root(local, group=None, root=0):
    """Rows of 2-D tensors whose row counts differ per rank -> their concatenation in rank order on ``root`` (None
    elsewhere).  One size exchange + one padded gather: only the root holds the whole matrix."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    height = max(max(sizes), 1)
    padded = torch.zeros((height,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)] if rank == root else None
    dist.gather(padded, parts, dst=root, group=group)
    if rank != root:
        return None
    return torch.cat([parts[r][: sizes[r]] for r in range(world)])


def rank_device():
    """This rank's GPU, chosen ONCE: LOCAL_RANK (or PHAMERS_HIP_DEVICE) modulo the number of visible devices.
    The phk context, torch's current device and the collective's tensors must all sit on it: RCCL refuses two
    ranks on one device and hangs when a rank's tensor lives on another rank's GPU."""
    import torch
    from . import _lib
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise RuntimeError("no GPU visible to this rank")
    dev = _lib.default_device() % ndev
    torch.cuda.set_device(dev)
    return dev


def default_scorer(device_index=None):
    """Per-rank scorer on the GPU path, device resident: (sequences, k, method, model inputs) -> scores.  The shard's
    bases go up once, counts and row sums stay in HBM (_lib.Batch), only the scores come back."""
    from . import _lib

    def score(sequences, kmer_length, method, positive, negative, cpos, cneg, k_neighbors):
        if len(sequences) == 0:
            return np.zeros(0)
        ctx = _lib.get_context(device_index)
        batch = _lib.Batch.from_sequences(ctx, list(sequences), kmer_length)
        model = _lib.Model(ctx, positive, negative, cpos if method != "knn" else None,
                           cneg if method != "knn" else None, k_neighbors)
        try:
            return batch.score(model, method)
        finally:
            model.close()
            batch.close()
    return score


def score_contigs_distributed(sequences, positive, negative, positive_centroids=None, negative_centroids=None,
                              kmer_length=4, method="combo", k_neighbors=3, group=None, scorer=None):
    """Count -> normalise -> score ``sequences`` with the work sharded over the ranks of ``group``;
    returns the full score vector (contig order) on every rank.  ``scorer`` defaults to the GPU
    path on this rank's device; the CPU tests pass the oracle here to exercise the sharding and
    the collective without a GPU."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run)")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_bounds([len(s) for s in sequences], world)[rank]
    use_cuda = dist.get_backend(group) == "nccl"
    gpu = rank_device() if (use_cuda or scorer is None) else None
    scorer = scorer or default_scorer(device_index=gpu)
    local = np.asarray(scorer(sequences[lo:hi], kmer_length, method, positive, negative,
                              positive_centroids, negative_centroids, k_neighbors), dtype=np.float64)
    dev = torch.device("cuda", gpu) if use_cuda else torch.device("cpu")
    full = gather_variable(torch.from_numpy(local).to(dev), group=group)
    return full.cpu().numpy()


def score_fasta_distributed(path, positive, negative, positive_centroids=None, negative_centroids=None,
                            kmer_length=4, method="combo", k_neighbors=3, group=None, length_requirement=None,
                            threads=0):
    """The sharded form of phamer_scorer.load_data + score_points on a FASTA file (scripts/kmer.py:124-140,
    scripts/phamer.py:131-142, 177-195): every rank parses ONLY the records that begin in its byte range of the file
    (phk_fasta_read_part: no rank ever holds another rank's shard), counts and scores them device resident, and the
    ranks exchange nothing but their ids and score vectors at the end.  ``length_requirement``: keep contigs of at least
    that many bases (screen_by_length, scripts/phamer.py:144-157; the reference's fixed 5000 when the screen is on).
    Returns (ids, scores) in file order on every rank."""
    import torch
    import torch.distributed as dist
    from . import _lib
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run or launch_ranks)")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    use_cuda = dist.get_backend(group) == "nccl"
    gpu = rank_device()
    ctx = _lib.get_context(gpu)
    fasta = _lib.Fasta(path, threads=threads, part=(rank, world))
    try:
        ids = fasta.phamers_ids()
        lengths = fasta.lengths()
        batch = _lib.Batch.from_fasta(ctx, fasta, kmer_length) if fasta.n_records else None
    finally:
        fasta.close()
    local = np.zeros(0)
    if batch is not None:
        model = _lib.Model(ctx, positive, negative, positive_centroids if method != "knn" else None,
                           negative_centroids if method != "knn" else None, k_neighbors)
        try:
            if length_requirement:
                keep = np.flatnonzero(lengths >= int(length_requirement))
                ids = ids[keep]
                sub = batch.select(keep) if len(keep) else None
                batch.close()
                batch = sub
            if batch is not None:
                local = batch.score(model, method)
        finally:
            model.close()
            if batch is not None:
                batch.close()
    dev = torch.device("cuda", gpu) if use_cuda else torch.device("cpu")
    scores = gather_variable(torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64)).to(dev), group=group)
    all_ids = [None] * world
    dist.all_gather_object(all_ids, [str(x) for x in ids], group=group)
    return np.array([x for part in all_ids for x in part]), scores.cpu().numpy()
