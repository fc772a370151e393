import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from phamers_amd import _lib, device, workloads
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = _lib.Context(0, stream.cuda_stream)
cfg = dict(workloads.CONFIGS[1]); k = cfg["k"]; L = cfg["length"]; n = cfg["contigs"]; T = n * L; D = 4 ** k
pos, neg, cpos, cneg, _ = workloads.reference_for(ctx, cfg, None)
model = _lib.Model(ctx, pos, neg, cpos, cneg, k_neighbors=3)
packed = torch.empty(device.packed_words(T), dtype=torch.int32, device=dev)
offsets = torch.empty(n + 1, dtype=torch.int64, device=dev)
counts = torch.empty((n, D), dtype=torch.int32, device=dev)
scores = torch.empty(n, dtype=torch.float64, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
device.synth_packed(ctx, 0, 0, n, L, packed.data_ptr(), offsets.data_ptr())
METHOD = sys.argv[1] if len(sys.argv) > 1 else "combo"
def step():
    device.count_score(ctx, model, packed.data_ptr(), None, T, offsets.data_ptr(), n, k, METHOD, counts.data_ptr(), scores.data_ptr(), status.data_ptr())
for _ in range(3): step()
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libphamers_hip.so"))
buf = (ctypes.c_ulonglong * 16)()
lib.phk_dbg_read(buf)
step(); torch.cuda.synchronize()
lib.phk_dbg_read(buf)
w = buf[6]
names = ["phaseA", "phaseB", "B.row_wait+sum", "B.centroid", "B.lo", "phaseC", "waves", "B.centre", "B.issue"]
for i, nm in enumerate(names): print(f"{nm:18s} {buf[i]/w:10.0f} cycles/wave")
