"""
learning.py -- drop-in for the hot-path helpers of PhaMers' scripts/learning.py.

    knn(queries, ref_data, ref_labels, k=3)       scripts/learning.py:118-128   -> GPU
    distances(vector, data)                       scripts/learning.py:47-56     -> GPU (float64, direct differences)
    closest_to(point, picks)                      scripts/learning.py:59-66     -> GPU distances + first-index argmin
    kmeans(data, k, ...)                          scripts/learning.py:131-146   -> scikit-learn's seeding on the host + its Lloyd sweeps on the GPU
    get_centroids(data, assignment)               scripts/learning.py:69-81     -> NumPy (86 means)

k-means reproduces the reference's scikit-learn fit (a per-run fit that does not depend on the number of
query contigs; the golden scores are pinned to its centroids): the k-means++
seeding is scikit-learn's own, on the host, the Lloyd iteration runs on the device (phk_kmeans_lloyd) and
gives the same labels; PHAMERS_KMEANS=sklearn keeps the whole fit on the host, PHAMERS_KMEANS=gpu selects
kmeans_gpu, a deterministic, version-independent device k-means (phk_kmeans).
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


def kmeans_plusplus_seeds(X, n_clusters, random_state):
    """scikit-learn's k-means++ seeding (sklearn/cluster/_kmeans.py, ``_kmeans_plusplus`` with its default
    ``n_local_trials = 2 + int(log k)`` and unit sample weights -- what ``KMeans.fit``, and so scripts/learning.py:138, runs
    on the mean-centred rows before its first sweep), restated with NumPy so that the product path does not import
    scikit-learn (0.5 s): the same draws from the same ``RandomState`` (one ``choice``, then ``uniform(size=trials)`` per
    centre), the same expressions in the same order (``-2 X Y^T + |x|^2 + |y|^2`` clipped at 0, ``cumsum`` +
    ``searchsorted``, the greedy choice among the trials).  Pinned to ``sklearn.cluster.kmeans_plusplus`` -- same indices
    -- on the reference matrices and on random ones by tests/test_host_rules.py."""
    X = np.asarray(X, dtype=np.float64)
    n_samples, n_features = X.shape
    x_sq = np.einsum("ij,ij->i", X, X)                       # row_norms(X, squared=True)
    weight = np.ones(n_samples, dtype=np.float64)            # _check_sample_weight(None, X)
    centers = np.empty((n_clusters, n_features), dtype=X.dtype)
    indices = np.full(n_clusters, -1, dtype=int)
    n_local_trials = 2 + int(np.log(n_clusters))

    def sq_dists(A):                                         # _euclidean_distances(A, X, Y_norm_squared=x_sq, squared=True)
        d = -2 * (A @ X.T)
        d += np.einsum("ij,ij->i", A, A)[:, None]
        d += x_sq.reshape(1, -1)
        np.maximum(d, 0, out=d)
        return d

    center_id = random_state.choice(n_samples, p=weight / weight.sum())
    centers[0] = X[center_id]
    indices[0] = center_id
    closest = sq_dists(centers[0, np.newaxis])
    pot = closest @ weight
    for c in range(1, n_clusters):
        rand_vals = random_state.uniform(size=n_local_trials) * pot
        cand = np.searchsorted(np.cumsum(weight * closest, dtype=np.float64), rand_vals)
        np.clip(cand, None, closest.size - 1, out=cand)
        d = sq_dists(X[cand])
        np.minimum(closest, d, out=d)
        cand_pot = d @ weight.reshape(-1, 1)
        best = np.argmin(cand_pot)
        pot = cand_pot[best]
        closest = d[best]
        centers[c] = X[cand[best]]
        indices[c] = cand[best]
    return centers, indices


def _one_blas_thread():
    """The seeding's matrix products are (a few trials) x D by D x n: one thread does them in microseconds, while a BLAS
    pool sized for the whole machine spins on cores the FASTA ingest and the upload are using beside this thread."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(1, user_api="blas")
    except ImportError:      # (threadpoolctl comes with scikit-learn; without it the products just run on BLAS's pool)
        import contextlib
        return contextlib.nullcontext()


KMEANS_MIN_GAP = 1e-9   # below this relative distance gap between a point's two nearest centres the host fit decides


def kmeans_reference_on_device(data, k, seed=kmeans_seed, max_iter=300, tol=1e-4, ctx=None):
    """The labels of ``KMeans(n_clusters=k, random_state=seed).fit(data)`` (scripts/learning.py:138) with only the seeding
    on the host: scikit-learn's k-means++ (kmeans_plusplus_seeds, on the mean-centred rows with a fresh
    ``RandomState(seed)`` -- what ``KMeans.fit`` does before its first sweep, n_init = 1) and its Lloyd iteration,
    stopping rule included, on the device (phk_kmeans_lloyd).  Returns (labels, sweeps), or None when a cluster ran
    empty (scikit-learn relocates it) or when some point came within KMEANS_MIN_GAP (relative) of a tie between its two
    nearest centres in some sweep: the caller then takes the host fit.

    Parity: the device forms distances by float64 direct differences, scikit-learn by chunked matrix products; the labels
    are EQUAL to scikit-learn 1.7.2's on the reference's matrices (pinned: tests/golden centroids) and on every fold
    subset / random matrix the tests try, and can differ in principle only for points nearer to a tie than the two
    formulations' rounding (~1e-13 relative), which the gap guard hands to the host fit.  Outside those pins: parity
    unpinned (scikit-learn is unpinned by the reference itself, requirements.txt:4)."""
    import ctypes
    X = np.array(data, dtype=np.float64, order="C")          # (a copy: centred in place, as KMeans.fit does)
    n, D = X.shape
    X -= X.mean(axis=0)
    with _one_blas_thread():
        init, _ = kmeans_plusplus_seeds(X, int(k), np.random.RandomState(seed))
    init = np.ascontiguousarray(init, dtype=np.float64)
    tol_abs = float(np.mean(np.var(X, axis=0)) * tol)
    labels = np.empty(n, dtype=np.uint32)
    n_iter, n_empty, min_gap = ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
    ctx = ctx or _lib.get_context()
    _lib.check(ctx.lib.phk_kmeans_lloyd(ctx.handle, _lib.ptr(X), n, D, int(k), _lib.ptr(init), tol_abs, int(max_iter), None,
                                        _lib.ptr(labels), ctypes.byref(n_iter), ctypes.byref(n_empty), ctypes.byref(min_gap)))
    if n_empty.value:
        return None
    # a point nearly equidistant from two centres: the device's direct differences and scikit-learn's chunked matrix
    # products (-2 x.c + |c|^2) may order them differently -- such a fit is left to the host
    if not (min_gap.value >= KMEANS_MIN_GAP):
        return None
    return labels.astype(np.int32), n_iter.value


def kmeans(data, k, verbose=False, sort_by_size=False, _ctx=None):
    """K-means labels (scripts/learning.py:131-146), equal to the reference's ``KMeans(n_clusters=k,
    random_state=10).fit(data).labels_``: scikit-learn's seeding on the host, its Lloyd iteration on the device
    (kmeans_reference_on_device).  PHAMERS_KMEANS=sklearn keeps the whole fit on the host, PHAMERS_KMEANS=gpu selects
    the version-independent device k-means (kmeans_gpu: other seeds, other centroids)."""
    import os
    mode = os.environ.get("PHAMERS_KMEANS", "device")
    if sort_by_size:
        raise NotImplementedError("sort_by_size is outside the accelerated path")
    if mode == "gpu":
        return kmeans_gpu(data, k)[0]
    if mode != "sklearn":
        got = kmeans_reference_on_device(data, k, ctx=_ctx)   # (_ctx: a context of the caller's own -- a helper thread's)
        if got is not None:
            return got[0]
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
