"""
learning.py -- drop-in for the hot-path helpers of PhaMers' scripts/learning.py.

    knn(queries, ref_data, ref_labels, k=3)       scripts/learning.py:118-128   -> GPU
    distances(vector, data)                       scripts/learning.py:47-56     -> GPU (float64, direct differences)
    closest_to(point, picks)                      scripts/learning.py:59-66     -> GPU distances + first-index argmin
    kmeans(data, k, ...)                          scripts/learning.py:131-146   -> scikit-learn (default) / GPU
    get_centroids(data, assignment)               scripts/learning.py:69-81     -> NumPy (86 means)

k-means uses scikit-learn by default, exactly as in the reference (a per-run fit that does not
depend on the number of query contigs; the golden scores are pinned to
its centroids); the centroids are an explicit input of the GPU scorer.  PHAMERS_KMEANS=gpu selects
kmeans_gpu, a deterministic device Lloyd k-means (phk_kmeans).
"""
import logging

import numpy as np

from . import _lib

kmeans_seed = 10  # scripts/learning.py:21

logging.basicConfig(format='[%(asctime)s][%(levelname)s][%(funcName)s] - %(message)s')
logger = logging.getLogger(__name__)
logger.setLevel(logging.WARNING)


def knn(queries, ref_data, ref_labels, k=3):
    """K-nearest-neighbours vote (scripts/learning.py:118-128): Euclidean, uniform weights,
    labels in {0, 1}; returns 2*(predicted label - 0.5), i.e. -1.0 / +1.0 per query."""
    ref_data = np.asarray(ref_data, dtype=np.float64)
    labels = np.asarray(ref_labels)
    if not np.all((labels == 0) | (labels == 1)):
        raise NotImplementedError("phamers_amd.learning.knn handles the reference's {0,1} labels only")
    queries = np.asarray(queries, dtype=np.float64)
    if np.isnan(queries).any() or np.isnan(ref_data).any():
        raise ValueError("Input contains NaN.")  # what scikit-learn raises for the reference
    ctx = _lib.get_context()
    model = _lib.Model(ctx, ref_data[labels == 1], ref_data[labels == 0], k_neighbors=k)
    try:
        return model.score(queries, "knn")
    finally:
        model.close()


def distances(vector, data):
    """Distances from one point to many (scripts/learning.py:47-56): ``vector`` (D,) or (1, D), ``data`` (M, D) ->
    (M,) float64, sqrt of the direct-difference sums, computed on the device (phk_distances).  Any other ``vector``
    shape fails to broadcast in the reference's ``np.repeat(vector, M, axis=0) - data`` and raises here as well."""
    vector = np.asarray(vector, dtype=np.float64)
    data = np.ascontiguousarray(data, dtype=np.float64)
    if vector.ndim == 1:
        vector = vector[None, :]
    if vector.shape[0] != 1:
        # np.repeat(vector, M, axis=0) - data only broadcasts for one row; anything else raises in the reference too
        raise ValueError("operands could not be broadcast together with shapes %s %s"
                         % ((vector.shape[0] * data.shape[0], vector.shape[1]), data.shape))
    if vector.shape[1] != data.shape[1]:
        raise ValueError("operands could not be broadcast together with shapes %s %s"
                         % ((data.shape[0], vector.shape[1]), data.shape))
    out = np.empty((1, data.shape[0]), dtype=np.float64)
    ctx = _lib.get_context()
    _lib.check(ctx.lib.phk_distances(ctx.handle, _lib.ptr(np.ascontiguousarray(vector)), 1, _lib.ptr(data),
                                     data.shape[0], data.shape[1], _lib.ptr(out)))
    return out[0]


def closest_to(point, picks):
    """The row of ``picks`` closest to ``point`` (scripts/learning.py:59-66): first index wins ties, as np.argmin."""
    picks = np.asarray(picks)
    return picks[np.argmin(distances(point, picks))]


def kmeans_gpu(data, k, seed=kmeans_seed, max_iter=300):
    """Deterministic device k-means (phk_kmeans): (labels, centroids, sweeps).  Version independent and
    bit-reproducible; NOT the scikit-learn result the reference's scores are pinned to."""
    import ctypes
    X = np.ascontiguousarray(data, dtype=np.float64)
    n, D = X.shape
    centroids = np.empty((k, D), dtype=np.float64)
    labels = np.empty(n, dtype=np.uint32)
    n_iter = ctypes.c_int()
    ctx = _lib.get_context()
    _lib.check(ctx.lib.phk_kmeans(ctx.handle, _lib.ptr(X), n, D, int(k), int(seed), int(max_iter), _lib.ptr(centroids),
                                  _lib.ptr(labels), ctypes.byref(n_iter)))
    return labels.astype(np.int64), centroids, n_iter.value


def kmeans(data, k, verbose=False, sort_by_size=False):
    """K-means labels (scripts/learning.py:131-146): scikit-learn with the reference's seed by default;
    PHAMERS_KMEANS=gpu selects the deterministic device implementation."""
    import os
    if os.environ.get("PHAMERS_KMEANS", "sklearn") == "gpu":
        return kmeans_gpu(data, k)[0]
    from sklearn.cluster import KMeans
    assignment = KMeans(n_clusters=k, random_state=kmeans_seed).fit(data).labels_
    if type(assignment) != np.ndarray:
        assignment = np.array(assignment)
    if sort_by_size:
        raise NotImplementedError("sort_by_size is outside the accelerated path")
    return assignment


def get_centroids(data, assignment):
    """Mean of the member rows per sorted label, -1 excluded (scripts/learning.py:69-81)."""
    data = np.asarray(data)
    labels = sorted(set(assignment) - set([-1]))
    if len(labels) == 0:
        logger.warning("No clusters assigned to data.")
    return np.array([np.mean(data[assignment == c], axis=0) for c in labels])
