#!/usr/bin/env python3
"""Round-3 analysis behind DESIGN.md section 7: how wide would the certification window be if the k = 4 proposal ran on
int8 MFMA (reference columns in 8-bit, or 16-bit, fixed point per column) instead of the fp16 high parts?  For the
benchmark's queries against the real matrix: |r' - quantised| per column, the Cauchy-Schwarz window 2 |q'| |delta_j|
around the 3rd-best value, and the number of columns inside it.  CPU only; not part of the product."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phamers_amd import workloads, synth
from oracle import oracle

pos, neg, cpos, cneg = workloads.phamers_reference()
R = np.vstack((pos, neg)); mu = R.mean(0); Rc = R - mu
seqs = synth.synth_contigs(0, 2048, 5000)
Q = oracle.normalize_counts(oracle.count(seqs, 4)); Qc = Q - mu
V = Qc @ Rc.T - 0.5 * (Rc ** 2).sum(1)[None, :]
Vs = -np.sort(-V, axis=1)
rn = np.sqrt((Rc ** 2).sum(1)); qn = np.sqrt((Qc ** 2).sum(1))


def residual_norm(kind):
    if kind == "fp16 high part":
        hi = (Rc * 4096.0).astype(np.float16).astype(np.float64) / 4096.0
        return np.sqrt(((Rc - hi) ** 2).sum(1))
    bits = 8 if kind.startswith("int8 x 1") else 16
    scale = np.abs(Rc).max(1, keepdims=True) / (2 ** (bits - 1) - 1)
    qz = np.round(Rc / scale) * scale
    return np.sqrt(((Rc - qz) ** 2).sum(1))


for kind in ("fp16 high part", "int8 x 1 (8-bit fixed point per column)", "int8 x 2 (16-bit fixed point per column)"):
    dl = residual_norm(kind)
    lam = dl.max()     # the bound takes the largest residual among the columns within reach; the global one is an upper estimate
    lam_typ = np.median(dl)
    members = []
    for lamv in (lam_typ, lam):
        e = qn * lamv
        members.append(((V >= (Vs[:, 2] - 2 * e)[:, None]).sum(1)))
    print("%-42s |delta|/|r'| median %.2e   window members (median residual): mean %.1f, > 6 in %.1f %% of queries, > 8 in %.1f %%   (largest residual: mean %.1f)"
          % (kind, np.median(dl / rn), members[0].mean(), 100 * (members[0] > 6).mean(), 100 * (members[0] > 8).mean(), members[1].mean()))
c = oracle.count(seqs, 4).reshape(len(seqs), -1)
print("count operand: 5 kb contigs max |c - c0| = %d (int8 carries 127); ragged workload:" % np.abs(c - np.round(c.sum(1, keepdims=True) / 256)).max(), end=" ")
lens = synth.ragged_lengths(1000, 400)
big = 0
for i, L in enumerate(lens):
    cc = oracle.count([synth.synth_ragged_contig(0, i, int(min(L, 120000)), 400, 1000)], 4).reshape(-1)
    big += np.abs(cc - round(cc.sum() / 256)).max() > 127
print("%.0f %% of its contigs exceed 127" % (100.0 * big / len(lens)))
