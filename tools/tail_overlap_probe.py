"""Would scoring the last partial round of proposal workgroups on a second stream pay?  Scores 1,000,000 count rows
(a) in one call, (b) as 917,504 rows (7 full rounds of 256 workgroups x 512 queries) on one stream and the remaining
82,496 on another, concurrently.  Run on the GPU box: python tools/tail_overlap_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from phamers_amd import _lib, device

n, L, k, D = 1000000, 5000, 4, 256
T = n * L
A, B = _lib.Context(0), _lib.Context(0)
pos, neg, cpos, cneg, _ = bench.load_model_inputs(D)
model = _lib.Model(A, pos, neg, cpos, cneg, k_neighbors=3)
packed = device.DeviceArray(A, device.packed_words(T), np.int32)
offsets = device.DeviceArray(A, n + 1, np.int64)
counts = device.DeviceArray(A, (n, D), np.int32)
device.synth_packed(A, 0, 0, n, L, packed, offsets)
device.count(A, packed, None, T, offsets, n, k, counts)
A.sync()
sa = device.DeviceArray(A, n, np.float64)
sb = device.DeviceArray(B, n, np.float64)
st = device.DeviceArray.from_host(A, np.zeros(1, np.int32))
n1 = 7 * 256 * 512

def whole():
    device.score_counts(A, model, counts, n, "combo", sa, st)
def split():
    device.score_counts(A, model, counts.ptr, n1, "combo", sa.ptr, st)
    device.score_counts(B, model, counts.ptr + n1 * D * 4, n - n1, "combo", sb.ptr, st)
def sync():
    A.sync(); B.sync()
def timeit(f, reps=6):
    f(); sync()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    sync()
    return (time.perf_counter() - t0) / reps * 1e3
tw = timeit(whole); ts = timeit(split); tw2 = timeit(whole)
print("one call %.3f ms   split over two streams %.3f ms   one call again %.3f ms" % (tw, ts, tw2))
ref = sa.to_host().copy()
split(); sync()
got = np.concatenate([sa.to_host()[:n1], sb.to_host()[:n - n1]])
print("scores identical:", bool(np.array_equal(got, ref)))
