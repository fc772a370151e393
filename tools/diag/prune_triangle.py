#!/usr/bin/env python3
"""Round-3 analysis behind DESIGN.md section 7 ("measured before choosing"): can 32-column blocks of the real reference
matrix be skipped by the triangle inequality (block centre + radius) for the benchmark's queries?  CPU only (NumPy,
scikit-learn for the clustering); prints the prunable fraction for three block orderings.  Not part of the product."""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from phamers_amd import workloads, synth
from oracle import oracle
pos,neg,cpos,cneg = workloads.phamers_reference()
R = np.vstack((pos,neg))
M,D = R.shape
# queries
n=4096
seqs = synth.synth_contigs(0,n,5000)
Q = oracle.normalize_counts(oracle.count(seqs,4))
d2 = (Q**2).sum(1)[:,None] + (R**2).sum(1)[None,:] - 2*Q@R.T
d = np.sqrt(np.maximum(d2,0))
ds = np.sort(d,axis=1)
print("d3 median", np.median(ds[:,2]), "d1", np.median(ds[:,0]), "d10", np.median(ds[:,9]), "d100", np.median(ds[:,99]), "d1000", np.median(ds[:,999]), "dmax", np.median(ds[:,-1]))
u = np.full(D,1/256)
du = np.sqrt(((R-u)**2).sum(1))
print("dist from uniform: quantiles", np.quantile(du,[0,0.01,0.1,0.5,0.9,1]))
# blocks: simple clustering - sort by projection? try kmeans-like balanced: greedy
from sklearn.cluster import KMeans
def blocks_by_sorting(R, key):
    order = np.argsort(key)
    return [order[i:i+32] for i in range(0,len(order),32)]
def eval_blocks(blocks, name):
    cent = np.stack([R[b].mean(0) for b in blocks])
    rad = np.array([np.sqrt(((R[b]-c)**2).sum(1)).max() for b,c in zip(blocks,cent)])
    dc = np.sqrt(np.maximum((Q**2).sum(1)[:,None] + (cent**2).sum(1)[None,:] - 2*Q@cent.T,0))
    sizes = np.array([len(b) for b in blocks])
    ub = np.where(sizes[None,:]>=3, dc+rad[None,:], np.inf).min(1)   # U(q)
    need = (dc - rad[None,:]) <= ub[:,None]
    # wave-uniform over groups of 64 queries
    g = need.reshape(-1,64,len(blocks)).any(1)
    print(name, "blocks",len(blocks),"rad median %.4f"%np.median(rad), "U median %.4f"%np.median(ub), "per-query need frac %.3f"%need.mean(), "per-wave(64) need frac %.3f"%g.mean())
    # with true d3 as ub
    need2 = (dc - rad[None,:]) <= ds[:,2][:,None]
    print("   with U = true d3: per-query %.3f per-wave %.3f"%(need2.mean(), need2.reshape(-1,64,len(blocks)).any(1).mean()))
eval_blocks(blocks_by_sorting(R, du), "sort by dist from uniform")
# PCA first component sort
Rc = R - R.mean(0)
U_,S_,Vt = np.linalg.svd(Rc, full_matrices=False)
eval_blocks(blocks_by_sorting(R, Rc@Vt[0]), "sort by PC1")
# kmeans then split into 32-chunks sorted by distance-to-centre within cluster
km = KMeans(n_clusters=141, n_init=1, random_state=0).fit(R)
blocks=[]
for c in range(141):
    idx = np.where(km.labels_==c)[0]
    # order within cluster by PC1 to make chunks tight
    idx = idx[np.argsort((Rc[idx]@Vt[0]))]
    for i in range(0,len(idx),32): blocks.append(idx[i:i+32])
eval_blocks(blocks, "kmeans141 + chunks")
