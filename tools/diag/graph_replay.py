"""ON THE GPU BOX: does replaying phk_count_score_dev's launch chain from a hipGraph beat enqueueing it call by call?
  Two consecutive calls -- the control blocks ping-pong, so a pair is the chain's period -- are captured
from the context's own stream (after warm-up: no allocation, no memset of a first use is left in the chain) and the graph is
replayed; the same number of steps is timed eagerly on the same box, interleaved, with HIP events on that stream.
usage: python tools/diag/graph_replay.py [--config 1] [--pairs 200] [--rounds 3]"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from phamers_amd import _lib, device, workloads  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: hipError %d" % (what, rc))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--contigs", type=int, default=None)
    ap.add_argument("--pairs", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    cfg = dict(workloads.CONFIGS[args.config])
    k, L = cfg["k"], cfg["length"]
    n = args.contigs or cfg.get("contigs") or 1000000
    stream = ctypes.c_void_p()
    chk(hip.hipSetDevice(0), "hipSetDevice")
    chk(hip.hipStreamCreate(ctypes.byref(stream)), "hipStreamCreate")
    ctx = _lib.Context(0, stream=stream.value)
    pos, neg, cpos, cneg, _ = workloads.reference_for(ctx, cfg, None)
    model = _lib.Model(ctx, pos, neg, cpos, cneg, k_neighbors=3)
    D, T = 4 ** k, n * L
    packed = device.DeviceArray(ctx, device.packed_words(T), np.uint32)
    offsets = device.DeviceArray(ctx, n + 1, np.uint64)
    counts = device.DeviceArray(ctx, (n, D), np.uint32)
    scores = [device.DeviceArray(ctx, n, np.float64) for _ in range(2)]
    status = device.DeviceArray.from_host(ctx, np.zeros(1, np.uint32))
    device.synth_packed(ctx, 0, 0, n, L, packed, offsets)

    def step(b):
        device.count_score(ctx, model, packed, None, T, offsets, n, k, "combo", counts, scores[b], status)

    for i in range(6):
        step(i & 1)
    ctx.sync()
    want = [s.to_host() for s in scores]

    graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
    chk(hip.hipStreamBeginCapture(stream, 2), "hipStreamBeginCapture")   # 2 = relaxed
    try:
        step(0)
        step(1)
    finally:
        rc = hip.hipStreamEndCapture(stream, ctypes.byref(graph))
    chk(rc, "hipStreamEndCapture")
    nnodes = ctypes.c_size_t(0)
    chk(hip.hipGraphGetNodes(graph, None, ctypes.byref(nnodes)), "hipGraphGetNodes")
    chk(hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, ctypes.c_size_t(0)), "hipGraphInstantiate")
    ev = [ctypes.c_void_p() for _ in range(2)]
    for e in ev:
        chk(hip.hipEventCreate(ctypes.byref(e)), "hipEventCreate")

    def timed(fn):
        chk(hip.hipEventRecord(ev[0], stream), "record")
        fn()
        chk(hip.hipEventRecord(ev[1], stream), "record")
        chk(hip.hipEventSynchronize(ev[1]), "sync")
        ms = ctypes.c_float()
        chk(hip.hipEventElapsedTime(ctypes.byref(ms), ev[0], ev[1]), "elapsed")
        return ms.value / (2 * args.pairs)

    def eager():
        for _ in range(args.pairs):
            step(0)
            step(1)

    def replay():
        for _ in range(args.pairs):
            chk(hip.hipGraphLaunch(gexec, stream), "hipGraphLaunch")

    print("config %d: %d contigs x %d bases, k = %d; graph of two steps: %d nodes" % (args.config, n, L, k, nnodes.value))
    for r in range(args.rounds):
        te = timed(eager)
        tg = timed(replay)
        print("round %d  eager %.4f ms per step   graph replay %.4f ms per step   (%+.2f %%)" % (r + 1, te, tg, 100.0 * (tg - te) / te))
    ctx.sync()
    got = [s.to_host() for s in scores]
    print("scores after replay identical to the eager calls':", all(np.array_equal(a, b) for a, b in zip(got, want)))


if __name__ == "__main__":
    main()
