#!/usr/bin/env python3
"""ON THE GPU BOX: ".gz" ingest -- one gzip member (zlib's serial inflate bounds it) against BGZF (blocks inflated in
parallel; a rank inflates only its byte range) against the plain file: file -> counts on the device, and the command line
end to end.  Prints one JSON object.  usage: tools/diag/gz_ingest.py [--contigs 200000]"""
import argparse
import json
import os
import struct
import sys
import tempfile
import time
import zlib

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
from phamers_amd import _lib, fileIO, phamer  # noqa: E402
import e2e_fasta  # noqa: E402


def bgzf_file(src, dst, block=65280):
    with open(src, "rb") as f, open(dst, "wb") as g:
        while True:
            chunk = f.read(block)
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            comp = c.compress(chunk) + c.flush()
            g.write(struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, len(comp) + 25))
            g.write(comp + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
            if not chunk:
                break


def gzip_file(src, dst):
    c = zlib.compressobj(1, zlib.DEFLATED, 31)
    with open(src, "rb") as f, open(dst, "wb") as g:
        while True:
            chunk = f.read(16 << 20)
            if not chunk:
                break
            g.write(c.compress(chunk))
        g.write(c.flush())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contigs", type=int, default=200000)
    a = ap.parse_args()
    ctx = _lib.get_context()
    root = tempfile.mkdtemp(prefix="phk_gz_")
    data = os.path.join(root, "data", "reference_features")
    os.makedirs(data)
    with np.load(os.path.join(REPO, "tests", "golden", "ref_features.npz")) as z:
        fileIO.save_counts(z["pos_counts"], z["pos_ids"], os.path.join(data, "positive_features.csv"))
        fileIO.save_counts(z["neg_counts"], z["neg_ids"], os.path.join(data, "negative_features.csv"))
    plain = os.path.join(root, "contigs.fasta")
    e2e_fasta.write_fasta(ctx, plain, a.contigs, 5000)
    out = {"contigs": a.contigs, "fasta_bytes": os.path.getsize(plain), "host_cores": os.cpu_count()}
    shapes = {"plain": plain, "gzip_one_member": os.path.join(root, "g", "contigs.fasta.gz"), "bgzf": os.path.join(root, "b", "contigs.fasta.gz")}
    os.makedirs(os.path.join(root, "g"))
    os.makedirs(os.path.join(root, "b"))
    gzip_file(plain, shapes["gzip_one_member"])
    bgzf_file(plain, shapes["bgzf"])
    want = None
    for name, path in shapes.items():
        r = {"bytes": os.path.getsize(path)}
        t = time.perf_counter()
        idx = _lib.Fasta(path, index_only=True)
        r["index_only_s"] = time.perf_counter() - t
        idx.close()
        t = time.perf_counter()
        idx, batch = _lib.Fasta.count_file(ctx, path, 4)
        r["file_to_counts_on_device_s"] = time.perf_counter() - t
        rows = batch.counts_u32()
        if want is None:
            want = rows
        r["counts_equal_plain"] = bool(np.array_equal(rows, want))
        batch.close(); idx.close()
        t = time.perf_counter()
        parts = []
        for i in range(8):       # what each rank of an 8-GPU run loads (here one after the other)
            t1 = time.perf_counter()
            pidx, pb = _lib.Fasta.count_file(ctx, path, 4, part=(i, 8), threads=max(1, (os.cpu_count() or 8) // 8))
            parts.append(time.perf_counter() - t1)
            pb.close(); pidx.close()
        r["one_of_8_parts_s_max"] = max(parts)
        if name != "plain":      # the command line on the compressed file (its own input directory)
            indir = os.path.dirname(path)
            t = time.perf_counter()
            phamer.main(["-in", indir, "-fasta", path, "-data", os.path.join(root, "data"), "-e", "-out", os.path.join(indir, "out")])
            r["cli_cold_s"] = time.perf_counter() - t
        out[name] = r
    print(json.dumps(out))


if __name__ == "__main__":
    main()
