#!/usr/bin/env python3
"""End-to-end line of the drop-in facade: FASTA on disk -> phamer_scores.csv (what a PhaMers user runs:
scripts/phamer.py:566-581), with the ingest / upload+count / cache / model / score / write split.

Runs ON THE GPU BOX.  Writes a synthetic N-contig FASTA (device-generated bases, reference-style headers) and the
real reference matrix into a scratch directory, then
  (a) times the stages the facade goes through, one by one, with the facade's own calls, and
  (b) times `phamers_amd.phamer.main([...])` as a whole, cold cache (FASTA counted on the GPU, features cache written)
      and warm cache (cache read back through np.loadtxt, as the reference does).
Prints one JSON object."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phamers_amd import _lib, device, fileIO, phamer, synth, workloads  # noqa: E402


def write_fasta(ctx, path, n, L, width=70, chunk=20000):
    """n seeded synthetic contigs (the bench's generator) as a FASTA file with `width`-column sequence lines."""
    lut = np.frombuffer(b"ATGC", dtype=np.uint8)
    shifts = np.arange(30, -2, -2, dtype=np.uint32)
    nl = (L + width - 1) // width
    with open(path, "wb") as f:
        for s in range(0, n, chunk):
            m = min(chunk, n - s)
            T = m * L
            packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
            off = device.DeviceArray(ctx, m + 1, np.uint64)
            device.synth_packed(ctx, 0, s, m, L, packed, off)
            words = packed.to_host()
            packed.free(); off.free()
            codes = ((words[:, None] >> shifts[None, :]) & 3).astype(np.uint8).reshape(-1)[:T]
            seqs = lut[codes].reshape(m, L)
            # sequence lines of `width` columns, each followed by '\n'
            pad = np.full((m, nl * width), ord("\n"), dtype=np.uint8)
            pad[:, :L] = seqs
            lines = np.concatenate([pad.reshape(m, nl, width), np.full((m, nl, 1), ord("\n"), np.uint8)], axis=2)
            body = lines.reshape(m, nl * (width + 1))
            tail = nl * width - L    # filler newlines of the last line are dropped
            for i in range(m):
                f.write((">%s\n" % synth.contig_header(s + i, L)).encode())
                f.write(body[i, : nl * (width + 1) - tail - 1].tobytes())
                f.write(b"\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contigs", type=int, default=1000000)
    ap.add_argument("--length", type=int, default=5000)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--skip-cli", action="store_true")
    a = ap.parse_args()
    ctx = _lib.get_context()
    root = a.dir or tempfile.mkdtemp(prefix="phk_e2e_")
    indir, data = os.path.join(root, "input"), os.path.join(root, "data", "reference_features")
    os.makedirs(indir, exist_ok=True)
    os.makedirs(data, exist_ok=True)
    with np.load(os.path.join(REPO, "tests", "golden", "ref_features.npz")) as z:
        fileIO.save_counts(z["pos_counts"], z["pos_ids"], os.path.join(data, "positive_features.csv"))
        fileIO.save_counts(z["neg_counts"], z["neg_ids"], os.path.join(data, "negative_features.csv"))
    fasta_path = os.path.join(indir, "contigs.fasta")
    t0 = time.perf_counter()
    write_fasta(ctx, fasta_path, a.contigs, a.length)
    t_gen = time.perf_counter() - t0
    fasta_bytes = os.path.getsize(fasta_path)
    bases = a.contigs * a.length
    out = {"contigs": a.contigs, "contig_length": a.length, "bases": bases, "fasta_bytes": fasta_bytes,
           "fasta_generation_s": t_gen, "stages_s": {}}
    st = out["stages_s"]

    # ---- (a) the facade's stages, one by one ----
    def timed(name, fn):
        t = time.perf_counter()
        r = fn()
        ctx.sync()
        st[name] = time.perf_counter() - t
        return r

    fasta = timed("fasta_read_parse_ids(native, threads)", lambda: _lib.Fasta(fasta_path))
    ids = timed("ids_to_numpy", fasta.phamers_ids)
    lengths = fasta.lengths()
    batch = timed("upload_pack_count(device)", lambda: _lib.Batch.from_fasta(ctx, fasta, 4))
    fasta.close()
    cache = os.path.join(root, "cache_features.csv")
    cu32 = timed("counts_download_u32", batch.counts_u32)
    timed("features_cache_write(native)", lambda: fileIO.save_counts(cu32, ids, cache))
    out["features_cache_bytes"] = os.path.getsize(cache)
    del cu32
    os.unlink(cache)
    pos, neg = timed("reference_csv_load(np.loadtxt)+normalise", lambda: tuple(
        fileIO.read_feature_file(os.path.join(data, f), normalize=True)[1] for f in ("positive_features.csv", "negative_features.csv")))
    m = min(len(pos), len(neg))
    pos, neg = pos[:m], neg[:m]
    sc = phamer.phamer_scorer()
    sc.positive_data, sc.negative_data = pos, neg
    timed("kmeans_fit(scikit-learn, 2 x 86 clusters)", sc._fit_centroids)
    model = timed("model_build_upload", lambda: _lib.Model(ctx, pos, neg, sc.positive_centroids, sc.negative_centroids, 3))
    keep = np.flatnonzero(lengths >= 5000)
    sel = timed("length_screen_device_gather", lambda: batch.select(keep))
    scores = timed("score(device)+scores_download", lambda: sel.score(model, "combo"))
    timed("scores_csv_write(native)", lambda: fileIO.save_phamer_scores(ids[keep], scores, os.path.join(root, "scores.csv")))
    out["staged_total_s"] = sum(st.values())
    out["staged_gbases_per_s"] = bases / out["staged_total_s"] / 1e9
    out["gpu_only_gbases_per_s(upload_pack_count+gather+score)"] = bases / (
        st["upload_pack_count(device)"] + st["length_screen_device_gather"] + st["score(device)+scores_download"]) / 1e9
    out["score_sample"] = scores[:3].tolist()
    sel.close(); batch.close(); model.close()

    # ---- (b) the command line as a user runs it ----
    if not a.skip_cli:
        argv = ["-in", indir, "-data", os.path.join(root, "data"), "--equalize_reference"] + (["--debug"] if os.environ.get("PHK_E2E_DEBUG") else [])
        prof = None
        if os.environ.get("PHK_E2E_PROFILE"):      # where the command line's time goes: cProfile, top of the cumulative list on stderr
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t = time.perf_counter()
        s1 = phamer.main(argv)
        out["cli_cold_s"] = time.perf_counter() - t
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
        out["cli_cold_gbases_per_s"] = bases / out["cli_cold_s"] / 1e9
        assert np.array_equal(s1.scores, scores)
        # the warm run reads <fasta>_features.csv (native reader; np.loadtxt needed minutes at 1 M contigs) and still
        # parses the FASTA for the length screen, as the reference does
        t = time.perf_counter()
        s2 = phamer.main(argv)
        out["cli_warm_cache_s"] = time.perf_counter() - t
        # (the cached path scores float64 normalised rows, the cold path the integer counts: equal to rounding)
        assert np.max(np.abs(s2.scores - scores) / np.maximum(np.abs(scores), 1e-300)) < 1e-12
    print(json.dumps(out))
    if not a.keep and not a.dir:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
