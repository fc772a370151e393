#!/usr/bin/env python3
"""Round-3 analysis behind DESIGN.md section 7 ("measured before choosing"): two-pass pruning by partial distances in
the reference's principal axes -- a cheap pass over all blocks at P of 256 dimensions, then the surviving blocks in
full -- at the granularity an MFMA tile works at (64 / 256 queries x 32 columns), for the homogeneous benchmark batch and
for the heterogeneous ragged one.  CPU only; prints the fraction of blocks alive and the total MFMA work relative to the
full sweep.  Not part of the product."""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from phamers_amd import workloads, synth
from oracle import oracle
pos,neg,cpos,cneg = workloads.phamers_reference()
R = np.vstack((pos,neg)); M,D = R.shape
mu = R.mean(0); Rc = R-mu
U_,S_,Vt = np.linalg.svd(Rc, full_matrices=False)
def queries(kind, n):
    if kind=="uniform": seqs = synth.synth_contigs(0,n,5000)
    else:
        lens = synth.ragged_lengths(1000, n)
        seqs = [synth.synth_ragged_contig(0, c, int(min(lens[c],20000)), 400, 1000) for c in range(n)]
    C = oracle.count(seqs,4).astype(float); C=C[C.sum(1)>0]
    return C/C.sum(1,keepdims=True)
rn = np.sqrt((Rc**2).sum(1))
for kind in ("uniform","ragged"):
    Q = queries(kind, 2048); Q=Q[:(len(Q)//256)*256]; Qc = Q-mu
    V = (Qc@Rc.T) - 0.5*(Rc**2).sum(1)[None,:]       # v values
    for order_name, order in (("|r'|", np.argsort(rn)), ("PC1", np.argsort(Rc@Vt[0]))):
        Ro = Rc[order]; Vo = V[:,order]
        for B0 in (2,4,8):
            n0 = B0*32
            tau = np.sort(Vo[:,:n0],axis=1)[:,-3]      # 3rd best v among pass-0 columns (lower bound on final 3rd best v)
            for P in (16,32):
                Vp = Vt[:P].T
                Rp = Ro@Vp; Qp = Qc@Vp
                # upper bound on v_j from partial: v_j <= (|q'|^2 - pd2_j)/2 where pd2 = |Qp - Rp|^2
                pd2 = (Qp**2).sum(1)[:,None] + (Rp**2).sum(1)[None,:] - 2*Qp@Rp.T
                vub = 0.5*((Qc**2).sum(1)[:,None] - pd2)
                # fp16-hi style error margin on partial products ~ 2^-10 |q_P||r_P|
                marg = 2.0**-10*np.sqrt((Qp**2).sum(1))[:,None]*np.sqrt((Rp**2).sum(1))[None,:]
                alive = (vub + marg) >= tau[:,None]
                nb = (M+31)//32
                a = np.pad(alive,((0,0),(0,nb*32-M)))
                for G,qsort in ((64,False),(256,False),(256,True)):
                    qo = np.argsort(Qp[:,0]) if qsort else np.arange(len(Q))
                    blk = a[qo].reshape(len(Q)//G,G,nb,32).any(axis=(1,3))
                    blk[:,:B0]=True
                    work = (blk[:,B0:].sum(1)*1.0 + (nb-B0)*P/256.0 + B0)/nb
                    print(kind, order_name, "B0",B0,"P",P,"G",G,"qsort",qsort, "alive blocks %.3f  total MFMA work %.3f"%(blk.mean(), work.mean()))
