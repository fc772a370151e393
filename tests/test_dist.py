"""Multi-rank path on CPU: sharding arithmetic (single process) and the sharded driver with the
final score gather over gloo, world_size 2 and 3 (the oracle stands in for the per-rank GPU
scorer, which lets the collective + ordering be checked without a GPU)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers
from tests.helpers import REPO

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PHK_REPO"])
import torch.distributed as dist
from oracle import oracle
from phamers_amd import dist as pdist, synth
from tests import helpers

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
g = helpers.load_npz("scoring_k4.npz")
ref = helpers.load_npz("ref_features.npz")
pos = oracle.normalize_counts(ref["pos_counts"][:300].astype(np.int64))
neg = oracle.normalize_counts(ref["neg_counts"][:300].astype(np.int64))
cp, cn = g["cpos_full"][:10], g["cneg_full"][:10]
lens = [5000, 0, 3, 1200, 5000, 800, 6000, 40, 5000, 2500, 5000]
seqs = [synth.synth_contig(2, i, L) for i, L in enumerate(lens)]
seqs[1] = "ATGCATGCAT"   # keep every row non-zero (zero rows -> NaN, compared separately)
seqs[2] = "GGATCCAATT"

def oracle_scorer(sequences, k, method, p, n, cpos, cneg, kn):
    if len(sequences) == 0:
        return np.zeros(0)
    q = oracle.normalize_counts(oracle.count(list(sequences), k).reshape(len(sequences), -1))
    return oracle.score_points(q, p, n, method, kn, cpos, cneg)

full = pdist.score_contigs_distributed(seqs, pos, neg, cp, cn, 4, "combo", 3, scorer=oracle_scorer)
want = oracle_scorer(seqs, 4, "combo", pos, neg, cp, cn, 3)
assert full.shape == want.shape, (full.shape, want.shape)
assert np.array_equal(full, want), np.abs(full - want).max()
# the raw collective with ragged (and empty) shards
import torch
r = dist.get_rank()
mine = torch.arange(r * 10, r * 10 + (0 if r == 1 else r + 2), dtype=torch.float64)
got = pdist.gather_variable(mine)
exp = torch.cat([torch.arange(q * 10, q * 10 + (0 if q == 1 else q + 2), dtype=torch.float64)
                 for q in range(dist.get_world_size())])
assert torch.equal(got, exp)
dist.barrier()
dist.destroy_process_group()
print("rank", r, "ok")
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_driver_and_gather_over_gloo(world, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PHK_REPO=REPO, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out)
        assert "ok" in out


def test_shard_bounds_cover_and_balance():
    from phamers_amd import dist as pdist
    rng = np.random.default_rng(3)
    for n, world in ((0, 4), (1, 8), (5, 8), (100, 1), (1000, 8), (17, 3)):
        lens = rng.integers(0, 20000, n)
        b = pdist.shard_bounds(lens, world)
        assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
        for (lo, hi), (lo2, hi2) in zip(b[:-1], b[1:]):
            assert lo <= hi == lo2 <= hi2
        if n >= 100:
            per = np.array([lens[lo:hi].sum() for lo, hi in b], dtype=float)
            assert per.max() - per.min() <= 2 * lens.max()
    # uniform contigs split evenly
    assert pdist.shard_bounds([5000] * 800, 8) == [(i * 100, (i + 1) * 100) for i in range(8)]


def test_shard_count_invariance_single_process():
    """Scores are identical for 1/2/4/8 shards: run the sharding logic with a fake gather
    (concatenate) -- no cross-contig arithmetic exists on the path."""
    from oracle import oracle
    from phamers_amd import dist as pdist, synth
    g = helpers.load_npz("scoring_k4.npz")
    ref = helpers.load_npz("ref_features.npz")
    pos = oracle.normalize_counts(ref["pos_counts"][:200].astype(np.int64))
    neg = oracle.normalize_counts(ref["neg_counts"][:200].astype(np.int64))
    seqs = [synth.synth_contig(8, i, 700 + 37 * i) for i in range(24)]
    q = oracle.normalize_counts(oracle.count(seqs, 4))
    want = oracle.score_points(q, pos, neg, "combo", 3, g["cpos_full"][:6], g["cneg_full"][:6])
    for world in (1, 2, 4, 8):
        parts = []
        for lo, hi in pdist.shard_bounds([len(s) for s in seqs], world):
            if hi > lo:
                qq = oracle.normalize_counts(oracle.count(seqs[lo:hi], 4).reshape(hi - lo, -1))
                parts.append(oracle.score_points(qq, pos, neg, "combo", 3, g["cpos_full"][:6], g["cneg_full"][:6]))
        assert np.array_equal(np.concatenate(parts), want)


def test_rank_device_follows_local_rank(monkeypatch):
    """The phk context, torch's current device and the gather tensor of a rank all use ONE device index:
    LOCAL_RANK modulo the visible device count (device functions mocked: no GPU here)."""
    import torch
    from phamers_amd import dist as pdist
    chosen = []
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: chosen.append(d))
    monkeypatch.delenv("PHAMERS_HIP_DEVICE", raising=False)
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert pdist.rank_device() == 5 and chosen == [5]
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    assert pdist.rank_device() == 1 and chosen == [5, 1]
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 0)
    with pytest.raises(RuntimeError):
        pdist.rank_device()


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


@pytest.mark.parametrize("world", [2, 3])
def test_bench_starts_its_own_ranks(world):
    """`python bench.py --gpus N` with no WORLD_SIZE / RANK in the environment (as the driver may run it) becomes the
    launcher: N child ranks, one process group, rank 0's line says n_gpus = N and the all-gather of rank ids saw N
    distinct ranks.  --rendezvous-only stops before any GPU work, so this runs here over gloo."""
    import json
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(world), "--rendezvous-only"],
                       env=_clean_env(PHK_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # exactly one JSON line: rank 0's
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["ranks_seen"] == world and line["launched_by"] == "launch_ranks"


def test_bench_refuses_to_measure_with_too_few_gpus():
    """RCCL needs a device per rank: with fewer than N visible (none here) nothing is started and the exit code is
    non-zero -- never a 1-rank measurement labelled as N."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=_clean_env(PHK_BENCH_BACKEND="nccl"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    assert "not started" in r.stderr and "{" not in r.stdout


def test_launch_ranks_env_and_failure(tmp_path):
    """Each child gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; a failing rank's exit code is returned and the
    other ranks are stopped."""
    from phamers_amd import dist as pdist
    script = tmp_path / "child.py"
    script.write_text(
        "import os, sys, time\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "assert int(os.environ['MASTER_PORT']) > 0\n"
        "open(os.path.join(sys.argv[1], 'rank%d_of_%d' % (r, w)), 'w').close()\n"
        "if sys.argv[2] == 'fail':\n"
        "    if r == 1: sys.exit(7)\n"
        "    time.sleep(60)\n")
    assert pdist.launch_ranks(3, [sys.executable, str(script), str(tmp_path), "ok"], require_gpus=False, timeout=120) == 0
    assert sorted(p.name for p in tmp_path.glob("rank*")) == ["rank0_of_3", "rank1_of_3", "rank2_of_3"]
    import time
    t0 = time.monotonic()
    assert pdist.launch_ranks(2, [sys.executable, str(script), str(tmp_path), "fail"], require_gpus=False, timeout=120) == 7
    assert time.monotonic() - t0 < 30        # rank 0 was terminated, not waited for


def test_visible_gpus_counts_kfd_nodes_without_the_runtime(tmp_path, monkeypatch):
    """The launcher counts GPUs from the KFD topology in sysfs (a node with simd_count > 0), narrowed by the
    *_VISIBLE_DEVICES variables -- no HIP / HSA call, so its children are the first to touch the device."""
    from phamers_amd import dist as pdist
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):       # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if not simd else 0, simd))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert pdist.visible_gpus(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert pdist.visible_gpus(str(tmp_path)) == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,7,0")        # the runtime stops at the entry it cannot resolve
    assert pdist.visible_gpus(str(tmp_path)) == 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert pdist.visible_gpus(str(tmp_path)) == 0
    assert pdist.visible_gpus(str(tmp_path / "absent")) == 0


def _cli_fasta(path):
    """Contigs either side of the 5000-base screen, titles with descriptions, lower case and N; 23 records so that 2 and 3
    byte ranges cut in the middle of records.  (Ids are unique: for a short contig that shares its id with a long one the
    reference keeps the row and drops the id, scripts/phamer.py:153-155, and cannot write its output.)"""
    from phamers_amd import synth
    lens = [5200, 300, 5000, 4999, 7000, 5100, 20, 6100, 5000, 5050, 900, 5600, 5000, 8000, 5001, 4000, 5300, 5000, 5000, 6500, 70, 5900, 5000]
    with open(path, "w") as f:
        for i, L in enumerate(lens):
            name = "SuperContig_%d_ID_%d" % (i, i)
            seq = synth.synth_contig(5, i, L)
            if i == 7:
                seq = seq[:1000] + "NNNNNNNNNN" + seq[1010:3000].lower() + seq[3000:]
            f.write(">%s some description %d\n" % (name, i))
            for a in range(0, len(seq), 70):
                f.write(seq[a:a + 70] + "\n")
    return lens


def _run_cli_ranks(world, indir, outdir, refdir, tmp_path):
    from phamers_amd import dist as pdist
    env = {"PHAMERS_DIST_BACKEND": "gloo", "PHAMERS_KMEANS": "sklearn", "PHK_REPO": REPO, "OMP_NUM_THREADS": "1",
           "PHAMERS_FORCE_RANK_PATH": "1"}
    argv = [sys.executable, os.path.join(REPO, "tests", "dist_cli_worker.py"), "-in", str(indir), "-out", str(outdir),
            "-pf", os.path.join(refdir, "pos.csv"), "-nf", os.path.join(refdir, "neg.csv"), "--gpus", str(world)]
    rc = pdist.launch_ranks(world, argv, require_gpus=False, timeout=600, extra_env=env)
    assert rc == 0
    # (the '#' header block of the score file lists the run's arguments -- --gpus, -out -- : compared without it)
    scores = b"\n".join(ln for ln in open(os.path.join(outdir, "phamer_scores.csv"), "rb").read().split(b"\n") if not ln.startswith(b"#"))
    cache = [f for f in os.listdir(indir) if f.endswith("_features.csv")]
    assert len(cache) == 1
    feats = open(os.path.join(indir, cache[0]), "rb").read()
    os.unlink(os.path.join(indir, cache[0]))      # (the next run must count the FASTA file again, not read the cache)
    return scores, feats


def test_command_line_ranks_write_the_one_rank_files_byte_for_byte(tmp_path):
    """`python -m phamers_amd.phamer -in <dir> ... --gpus N` (phamer._run_rank) over gloo with 1, 2 and 3
    ranks: rank 0's phamer_scores.csv and <fasta>_features.csv are the same bytes whatever the number of ranks, and equal
    the files built here from the oracle over the whole FASTA file with the reference's length screen
    (scripts/phamer.py:144-157).  The per-rank GPU work is replaced by the oracle
    (tests/dist_cli_worker.py); everything else is the command line's own code."""
    from oracle import oracle
    from phamers_amd import _lib, fileIO, learning
    ref = helpers.load_npz("ref_features.npz")
    refdir = tmp_path / "ref"
    refdir.mkdir()
    npos, nneg = 260, 240
    fileIO.save_counts(ref["pos_counts"][:npos].astype(np.int64), ["p%d" % i for i in range(npos)], str(refdir / "pos.csv"))
    fileIO.save_counts(ref["neg_counts"][:nneg].astype(np.int64), ["n%d" % i for i in range(nneg)], str(refdir / "neg.csv"))
    indir = tmp_path / "in"
    indir.mkdir()
    lens = _cli_fasta(str(indir / "contigs.fasta"))
    outs = {}
    for world in (1, 2, 3):
        out = tmp_path / ("out%d" % world)
        outs[world] = _run_cli_ranks(world, indir, out, str(refdir), tmp_path)
    assert outs[2] == outs[1] and outs[3] == outs[1]
    # ... and what those bytes must be
    fa = _lib.Fasta(str(indir / "contigs.fasta"))
    ids, seqs = fa.phamers_ids(), fa.sequences()
    fa.close()
    assert [len(x) for x in seqs] == lens
    counts = oracle.count(seqs, 4).reshape(len(seqs), 256)
    is_long = np.array(lens) >= 5000
    keep = is_long
    assert int(keep.sum()) == 17 and not keep[1]
    pos = oracle.normalize_counts(ref["pos_counts"][:npos].astype(np.int64))
    neg = oracle.normalize_counts(ref["neg_counts"][:nneg].astype(np.int64))
    os.environ["PHAMERS_KMEANS"] = "sklearn"
    try:
        cpos = learning.get_centroids(pos, learning.kmeans(pos, 86))
        cneg = learning.get_centroids(neg, learning.kmeans(neg, 86))
    finally:
        del os.environ["PHAMERS_KMEANS"]
    want = oracle.score_points(oracle.normalize_counts(counts[keep]), pos, neg, "combo", 3, cpos, cneg)
    want_scores = tmp_path / "want_scores.csv"
    fileIO.save_phamer_scores(ids[keep], want, str(want_scores))
    strip = lambda b: b"\n".join(ln for ln in b.split(b"\n") if not ln.startswith(b"#"))
    assert outs[2][0] == strip(open(want_scores, "rb").read())
    want_feats = tmp_path / "want_features.csv"
    fileIO.save_counts(counts.astype(np.uint32), ids, str(want_feats))
    assert outs[2][1] == open(want_feats, "rb").read()


def test_command_line_gpus_flag_starts_ranks_as_children(monkeypatch):
    """--gpus N in a process that is not a rank: N children of `python -m phamers_amd.phamer <same arguments>`, nothing on
    the GPU in the launcher itself; a failing rank's exit code becomes the command's."""
    from phamers_amd import dist as pdist, phamer
    seen = {}

    def fake_launch(n, argv, require_gpus=True, **kw):
        seen.update(n=n, argv=list(argv), require_gpus=require_gpus)
        return seen.get("rc", 0)
    monkeypatch.setattr(pdist, "launch_ranks", fake_launch)
    for var in ("WORLD_SIZE", "RANK", "PHAMERS_FORCE_RANK_PATH", "PHAMERS_DIST_BACKEND"):
        monkeypatch.delenv(var, raising=False)
    args = ["-in", "/nonexistent/in", "-data", "/nonexistent/data", "--gpus", "4", "-e"]
    assert phamer.main(args) is None
    assert seen["n"] == 4 and seen["require_gpus"] is True
    assert seen["argv"][1:3] == ["-m", "phamers_amd.phamer"] and seen["argv"][3:] == args
    seen["rc"] = 7
    with pytest.raises(SystemExit) as e:
        phamer.main(args)
    assert e.value.code == 7
