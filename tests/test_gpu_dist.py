"""The sharded path on the GPU: shard-count invariance of the device scorer (scores bit-identical for
1 / 2 / 4 / 8 shards) and the product-level sharded entry points with their default (GPU) scorer under a 2-rank gloo
group on one device (RCCL needs a device per rank), and -- as 1-rank `nccl` groups in fresh child processes -- the RCCL
code path itself: init_process_group(device_id=...), the asynchronous double-buffered all_gather_into_tensor of
bench.py against the library's stream, gather_variable on device tensors, destroy_process_group."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers
from tests.helpers import REPO

pytestmark = pytest.mark.gpu


def _contigs():
    """A ragged, awkward batch: heavy-tailed lengths, N runs, lower case, a few long low-complexity contigs whose counts
    exceed what the fp16 count operand carries (second chance), and near-duplicates (near-ties between neighbours)."""
    from phamers_amd import synth
    rng = np.random.default_rng(11)
    lens = np.concatenate((rng.integers(300, 9000, 1500), rng.integers(20000, 90000, 40), [5000] * 500))
    rng.shuffle(lens)
    seqs = [synth.synth_contig(5, i, int(L), 300 if i % 7 == 0 else 0) for i, L in enumerate(lens)]
    seqs[3] = "ATATATATAT" * 3000                     # counts far above 2048 in a few bins
    seqs[10] = "ATGC" * 2500 + "GGGGCCCC" * 400
    seqs[17] = seqs[16]                               # exact duplicate
    seqs[18] = seqs[16][:-1] + ("A" if seqs[16][-1] != "A" else "T")
    seqs[25] = seqs[24].lower()[:2000] + seqs[24][2000:]
    return seqs


def test_shard_count_invariance_on_the_device():
    from oracle import oracle
    from phamers_amd import _lib, dist as pdist
    g = helpers.load_npz("scoring_k4.npz")
    ref = helpers.load_npz("ref_features.npz")
    pos = oracle.normalize_counts(ref["pos_counts"].astype(np.int64))[:2255]
    neg = oracle.normalize_counts(ref["neg_counts"].astype(np.int64))[:2255]
    cp, cn = g["cpos_eq"], g["cneg_eq"]
    seqs = _contigs()
    scorer = pdist.default_scorer()
    out = {}
    for world in (1, 2, 4, 8):
        parts = [scorer(seqs[lo:hi], 4, "combo", pos, neg, cp, cn, 3)
                 for lo, hi in pdist.shard_bounds([len(s) for s in seqs], world)]
        out[world] = np.concatenate(parts)
        assert out[world].shape == (len(seqs),)
    for world in (2, 4, 8):
        assert np.array_equal(out[1], out[world]), world
    # and against the oracle on a sample (incl. the awkward rows)
    pick = [0, 3, 10, 16, 17, 18, 24, 25] + list(range(100, 140))
    q = oracle.normalize_counts(oracle.count([seqs[i] for i in pick], 4))
    want = oracle.knn_score_points(q, pos, neg, 3) + oracle.centroid_score_points_fast(q, cp, cn)
    assert helpers.rel_err(out[1][pick], want) < 1e-6
    # the scoring-batch split inside one shard does not matter either
    ctx = _lib.get_context()
    ctx.set_option("score_batch", "256")
    again = scorer(seqs, 4, "combo", pos, neg, cp, cn, 3)
    ctx.set_option("score_batch", "0")
    assert np.array_equal(again, out[1])


WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["PHK_REPO"])
import torch.distributed as dist
from phamers_amd import dist as pdist
from tests import test_gpu_dist as T, helpers
from oracle import oracle

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
g = helpers.load_npz("scoring_k4.npz")
ref = helpers.load_npz("ref_features.npz")
pos = oracle.normalize_counts(ref["pos_counts"].astype(np.int64))[:2255]
neg = oracle.normalize_counts(ref["neg_counts"].astype(np.int64))[:2255]
seqs = T._contigs()
full = pdist.score_contigs_distributed(seqs, pos, neg, g["cpos_eq"], g["cneg_eq"], 4, "combo", 3)   # default: GPU scorer
fasta = os.environ["PHK_FASTA"]
full_f = pdist.score_fasta_distributed(fasta, pos, neg, g["cpos_eq"], g["cneg_eq"], 4, "combo", 3)
if dist.get_rank() == 0:
    np.save(os.environ["PHK_OUT"], np.stack([full, full_f[1]]))
    json.dump([str(x) for x in full_f[0][:5]], open(os.environ["PHK_OUT"] + ".ids.json", "w"))
dist.barrier()
dist.destroy_process_group()
'''


def test_sharded_entry_points_with_the_gpu_scorer_two_ranks(tmp_path):
    """dist.score_contigs_distributed / score_fasta_distributed with their default scorer (device resident, this rank's
    GPU), two ranks started by dist.launch_ranks sharing the one device over gloo: the gathered vector equals the
    single-process device result bit for bit."""
    from oracle import oracle
    from phamers_amd import dist as pdist
    seqs = _contigs()
    fasta = tmp_path / "contigs.fasta"
    with open(fasta, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">SuperContig_%d_length_%d_ID_%d\n" % (i, len(s), i))
            for j in range(0, len(s), 70):
                f.write(s[j:j + 70] + "\n")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    outp = str(tmp_path / "out.npy")
    rc = pdist.launch_ranks(2, [sys.executable, str(script)], require_gpus=False, timeout=900,
                            extra_env={"PHK_REPO": REPO, "PHK_OUT": outp, "PHK_FASTA": str(fasta), "OMP_NUM_THREADS": "4"})
    assert rc == 0
    got = np.load(outp)
    g = helpers.load_npz("scoring_k4.npz")
    ref = helpers.load_npz("ref_features.npz")
    pos = oracle.normalize_counts(ref["pos_counts"].astype(np.int64))[:2255]
    neg = oracle.normalize_counts(ref["neg_counts"].astype(np.int64))[:2255]
    want = pdist.default_scorer()(seqs, 4, "combo", pos, neg, g["cpos_eq"], g["cneg_eq"], 3)
    assert np.array_equal(got[0], want)
    assert np.array_equal(got[1], want)
    assert json.load(open(outp + ".ids.json")) == ["0", "1", "2", "3", "4"]


def _rccl_env(**extra):
    from phamers_amd import dist as pdist
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(pdist.free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0",
               OMP_NUM_THREADS="4")
    env.update(extra)
    return env


def test_bench_runs_the_rccl_gather_as_a_one_rank_group():
    """bench.py with PHK_BENCH_FORCE_DIST=1 in a fresh child process: a 1-rank RCCL group on the one GPU of the box --
    init_process_group("nccl", device_id=...), the rank-id all-gather, the per-step asynchronous
    all_gather_into_tensor on two buffers beside the library's stream, the closing barrier, destroy_process_group.
    The line must report one rank seen, green parity, and a gathered vector equal to the score vector."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--config", "1", "--contigs", "65536",
                        "--no-cpu-baseline", "--min-seconds", "0", "--steps", "3", "--warmup", "1"],
                       env=_rccl_env(PHK_BENCH_FORCE_DIST="1", PHK_BENCH_BACKEND="nccl"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["steps"] == 3 and line["steps_requested"] == 3
    g = line["gather"]
    assert g["backend"] == "nccl" and g["async"] is True
    assert g["gathered_equals_scores"] is True and g["slices_arrived"] == 1
    par = line["parity"]
    assert par["counts_bit_exact"] is True and par["max_rel_score_err"] < 1e-6 and par["nan_rows"] == 0
    assert line["value"] > 0 and line["config"]["contigs_per_gpu"] == 65536


RCCL_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PHK_REPO"])
import torch
import torch.distributed as dist
from phamers_amd import dist as pdist
from tests import test_gpu_dist as T, helpers
from oracle import oracle

dev = pdist.rank_device()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", dev))
assert pdist.ranks_seen(device=torch.device("cuda", dev)) == 1
g = helpers.load_npz("scoring_k4.npz")
ref = helpers.load_npz("ref_features.npz")
pos = oracle.normalize_counts(ref["pos_counts"].astype(np.int64))[:2255]
neg = oracle.normalize_counts(ref["neg_counts"].astype(np.int64))[:2255]
seqs = T._contigs()
full = pdist.score_contigs_distributed(seqs, pos, neg, g["cpos_eq"], g["cneg_eq"], 4, "combo", 3)
ids, full_f = pdist.score_fasta_distributed(os.environ["PHK_FASTA"], pos, neg, g["cpos_eq"], g["cneg_eq"], 4, "combo", 3)
# gather_variable on device tensors of different lengths is what N > 1 ranks run; with one rank: identity
t = torch.arange(7, dtype=torch.float64, device="cuda:%d" % dev)
assert torch.equal(pdist.gather_variable(t), t)
np.save(os.environ["PHK_OUT"], np.stack([full, full_f]))
dist.barrier()
dist.destroy_process_group()
'''


def test_sharded_entry_points_under_a_one_rank_rccl_group(tmp_path):
    """dist.score_contigs_distributed / score_fasta_distributed under the `nccl` backend (the scores travel as DEVICE
    tensors through RCCL's all-gather) in a fresh child process: equal to the single-process device result."""
    from oracle import oracle
    from phamers_amd import dist as pdist
    seqs = _contigs()
    fasta = tmp_path / "contigs.fasta"
    with open(fasta, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">SuperContig_%d_length_%d_ID_%d\n" % (i, len(s), i))
            for j in range(0, len(s), 70):
                f.write(s[j:j + 70] + "\n")
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    outp = str(tmp_path / "out.npy")
    r = subprocess.run([sys.executable, str(script)], env=_rccl_env(PHK_REPO=REPO, PHK_OUT=outp, PHK_FASTA=str(fasta)),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    got = np.load(outp)
    g = helpers.load_npz("scoring_k4.npz")
    ref = helpers.load_npz("ref_features.npz")
    pos = oracle.normalize_counts(ref["pos_counts"].astype(np.int64))[:2255]
    neg = oracle.normalize_counts(ref["neg_counts"].astype(np.int64))[:2255]
    want = pdist.default_scorer()(seqs, 4, "combo", pos, neg, g["cpos_eq"], g["cneg_eq"], 3)
    assert np.array_equal(got[0], want)
    assert np.array_equal(got[1], want)


def test_command_line_rank_path_under_a_one_rank_rccl_group(tmp_path):
    """`python -m phamers_amd.phamer -in <dir> -data <dir> -e` twice in fresh child processes: as the one-GPU run (phamer._run)
    and as a rank of `--gpus N` (phamer._run_rank: PHAMERS_FORCE_RANK_PATH=1 makes it a 1-rank `nccl` group on the box's
    one GPU -- init_process_group(device_id=...), the centroid broadcast, the ids / lengths exchange for the length screen,
    phk_batch_from_fasta_part, the score gather and the count gather to rank 0 as DEVICE tensors through RCCL).  Both must
    write the same phamer_scores.csv (below the header block, which lists the arguments) and the same features cache,
    byte for byte.  The 2- and 3-rank forms of the same code run over gloo in tests/test_dist.py."""
    from phamers_amd import fileIO, synth
    ref = helpers.load_npz("ref_features.npz")
    data = tmp_path / "data" / "reference_features"
    data.mkdir(parents=True)
    fileIO.save_counts(ref["pos_counts"], ref["pos_ids"], str(data / "positive_features.csv"))
    fileIO.save_counts(ref["neg_counts"], ref["neg_ids"], str(data / "negative_features.csv"))
    lens = [5000] * 40 + [300, 5000, 4999, 7000, 20, 6100, 900, 5600, 8000, 70] + [5000] * 30
    outs = {}
    for name, extra in (("plain", {}), ("rank", {"PHAMERS_FORCE_RANK_PATH": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})):
        indir = tmp_path / ("in_" + name)
        indir.mkdir()
        with open(indir / "contigs.fasta", "w") as f:
            for c, L in enumerate(lens):
                s = synth.synth_contig(3, c, L)
                f.write(">SuperContig_%d_length_%d_ID_%d\n" % (c, L, c))
                f.write("\n".join(s[i:i + 70] for i in range(0, L, 70)) + "\n")
        r = subprocess.run([sys.executable, "-m", "phamers_amd.phamer", "-in", str(indir), "-data", str(tmp_path / "data"), "-e"],
                           cwd=REPO, env=_rccl_env(**extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (name, r.stderr[-4000:])
        sc = (indir / "phamer_output" / "phamer_scores.csv").read_bytes()
        outs[name] = (b"\n".join(ln for ln in sc.split(b"\n") if not ln.startswith(b"#")), (indir / "contigs_features.csv").read_bytes())
    assert outs["plain"][0].count(b"\n") >= 70
    assert outs["rank"] == outs["plain"]
