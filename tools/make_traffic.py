#!/usr/bin/env python3
"""profiles/<round>/traffic.json from a pmc_summary.json of tools/profile_gpu.sh: HBM bytes per contig per kernel.
FETCH_SIZE is in KB and, on gfx950, reports half of the bytes of wide coalesced streaming reads
(MI355X_MICROARCH.md, HBM section): it is doubled; WRITE_SIZE (KB) is taken as it is."""
import json
import sys

src, dst, contigs = sys.argv[1], sys.argv[2], float(sys.argv[3])
pmc = json.load(open(src))
out = {"contigs": int(contigs), "note": "FETCH_SIZE doubled (gfx950), WRITE_SIZE as reported; summed over the dispatches of one bench step, "
                                          "separate rocprofv3 --pmc passes", "kernels": {}}
for k, v in pmc.items():
    name = k.split("<")[0].replace("void ", "").strip()
    f, w = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
    if f is None or w is None:
        continue
    e = out["kernels"].setdefault(name, {"fetch_KB_raw": 0.0, "write_KB": 0.0})
    e["fetch_KB_raw"] += f
    e["write_KB"] += w
for e in out["kernels"].values():
    e["hbm_bytes_per_contig"] = round((2.0 * e["fetch_KB_raw"] + e["write_KB"]) * 1024.0 / contigs, 1)
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out["kernels"], indent=1, sort_keys=True))
