"""
dist.py -- contig sharding across the GPUs of one node and the single collective of the path:
the final gather of the per-rank score vectors (RCCL all-gather over xGMI; gloo on CPU for tests).

Contigs are independent (scripts/kmer.py:102-105, scripts/phamer.py:251-255), so the batch is cut
into contiguous contig ranges balanced in bases, every rank counts + scores its own range against
a replicated reference matrix, and nothing is exchanged until the scores are gathered.  One
process per GPU, launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE).
"""
import numpy as np


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
