"""Shared test helpers: golden loaders and a NumPy statement of the packed-base
format (include/phamers_hip.h, "Packed base stream") used to cross-check the
device packer.  Test-only code."""
import json
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def load_json(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def count_cases():
    """name -> sequence string for every golden count case."""
    from phamers_amd import synth
    doc = load_json("count_cases.json")
    out = {}
    for name, spec in doc["cases"].items():
        if "seq" in spec:
            out[name] = spec["seq"]
        else:
            seed, c, L, ppm = spec["synth"]
            out[name] = synth.synth_contig(seed, c, L, ppm)
    return out, doc


def pack_codes(codes):
    """codes (int8, -1 invalid) of the whole concatenated stream -> (packed u32 words,
    validity-mask u32 words).  Base g sits in word g//16 at bits [31-2*(g%16)-1, 31-2*(g%16)]
    (first base in the most significant bits); validity bit of base g is bit 31-(g%32) of
    mask word g//32.  One zero pad word is appended to each."""
    codes = np.asarray(codes, dtype=np.int64)
    T = codes.shape[0]
    nw = (T + 15) // 16
    c = np.zeros(nw * 16, dtype=np.uint64)
    c[:T] = np.where(codes < 0, 0, codes)
    sh = (30 - 2 * (np.arange(nw * 16) % 16)).astype(np.uint64)
    packed = (c << sh).reshape(nw, 16).sum(axis=1).astype(np.uint32)
    nm = (T + 31) // 32
    v = np.zeros(nm * 32, dtype=np.uint64)
    v[:T] = codes >= 0
    shm = (31 - (np.arange(nm * 32) % 32)).astype(np.uint64)
    mask = (v << shm).reshape(nm, 32).sum(axis=1).astype(np.uint32)
    return np.append(packed, np.uint32(0)), np.append(mask, np.uint32(0))


def rel_err(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)) if a.size else 0.0
