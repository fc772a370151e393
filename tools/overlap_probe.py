"""Can the count kernel of one batch run in the shadow of the proposal / decision kernels of another?
Two contexts (two streams) on one GPU: A scores precomputed counts, B counts a packed batch.
Prints wall times alone and together.  Run on the GPU box:  PHK_CX_CFG=24 python tools/overlap_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from phamers_amd import _lib, device

n, L, k, D = int(os.environ.get("N", 1000000)), 5000, 4, 256
T = n * L
A, B = _lib.Context(0), _lib.Context(0)
pos, neg, cpos, cneg, _ = bench.load_model_inputs(D)
model = _lib.Model(A, pos, neg, cpos, cneg, k_neighbors=3)
bufs = {}
for name, c in (("A", A), ("B", B)):
    packed = device.DeviceArray(c, device.packed_words(T), np.int32)
    offsets = device.DeviceArray(c, n + 1, np.int64)
    counts = device.DeviceArray(c, (n, D), np.int32)
    device.synth_packed(c, 0, 0, n, L, packed, offsets)
    device.count(c, packed, None, T, offsets, n, k, counts)
    bufs[name] = (packed, offsets, counts)
scores = device.DeviceArray(A, n, np.float64)
status = device.DeviceArray.from_host(A, np.zeros(1, np.int32))

def score():
    device.score_counts(A, model, bufs["A"][2], n, "combo", scores, status)
def count():
    p, o, c = bufs["B"]
    device.count(B, p, None, T, o, n, k, c)
def sync():
    A.sync(); B.sync()
def timeit(f, reps=5):
    f(); sync()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    sync()
    return (time.perf_counter() - t0) / reps * 1e3
ts = timeit(score); tc = timeit(count)
tb = timeit(lambda: (score(), count()))
print("cfg=%s  score alone %.3f ms   count alone %.3f ms   both streams %.3f ms   (sum %.3f)" %
      (os.environ.get("PHK_CX_CFG", "default"), ts, tc, tb, ts + tc))
