import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
t0=time.perf_counter()
from phamers_amd import learning, _lib, workloads
pos, neg, cp, cn = workloads.phamers_reference()
ctx=_lib.get_context()
print("setup %.3f" % (time.perf_counter()-t0))
for rep in range(3):
    t0=time.perf_counter()
    from sklearn.cluster import kmeans_plusplus
    t1=time.perf_counter()
    X = np.array(pos, dtype=np.float64, order="C"); X -= X.mean(axis=0)
    t2=time.perf_counter()
    init,_ = kmeans_plusplus(X, 86, random_state=np.random.RandomState(10))
    t3=time.perf_counter()
    got = learning.kmeans_reference_on_device(pos, 86)
    t4=time.perf_counter()
    c = learning.get_centroids(pos, got[0])
    t5=time.perf_counter()
    print("rep %d import %.3f centre %.3f seed %.3f whole-route %.3f (so device part ~%.3f) get_centroids %.3f sweeps %d" % (rep, t1-t0, t2-t1, t3-t2, t4-t3, (t4-t3)-(t3-t2), t5-t4, got[1]))
