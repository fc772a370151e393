"""One rank of `python -m phamers_amd.phamer ... --gpus N` for the CPU tests (tests/test_dist.py): the command line's own
rank code (phamer._run_rank: process group, replicated reference, byte-range shards, the global length screen, the gathers,
rank 0's files) with the per-rank GPU work -- phamer._rank_count_and_score -- replaced by the native FASTA reader + the
oracle.  The product has no CPU path; this file is test infrastructure."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ["PHK_REPO"])
from oracle import oracle                     # noqa: E402
from phamers_amd import _lib, phamer          # noqa: E402


def oracle_rank(fasta_file, part, kmer_length, method, positive, negative, cpos, cneg, k_neighbors, keep_of, gpu):
    fa = _lib.Fasta(fasta_file, part=part)
    try:
        ids, lengths, seqs = fa.phamers_ids(), fa.lengths(), fa.sequences()
    finally:
        fa.close()
    D = 4 ** int(kmer_length)
    counts = oracle.count(seqs, kmer_length).reshape(len(seqs), D).astype(np.uint32) if seqs else np.zeros((0, D), np.uint32)
    keep = keep_of(ids, lengths)
    scores = np.zeros(0)
    if keep.any():
        q = oracle.normalize_counts(counts[keep].astype(np.int64))
        scores = oracle.score_points(q, positive, negative, method, k_neighbors, cpos, cneg)
    return ids, counts, keep, scores


phamer._rank_count_and_score = oracle_rank
from phamers_amd import kmer                  # noqa: E402
kmer.normalize_counts = lambda counts: oracle.normalize_counts(np.asarray(counts))   # (the reference matrices' rows: bit-equal to the device's)
phamer.main(sys.argv[1:])
print("rank", os.environ.get("RANK"), "ok")
