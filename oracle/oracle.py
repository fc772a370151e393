"""
oracle.py -- CPU restatement of the PhaMers k-mer count + phage-score hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and there only as the checker / the timed CPU baseline.  Nothing under
``phamers_amd/`` imports this module; the product path is the HIP library and
fails loudly when that library is missing.

Every function cites the reference file:line (relative to the PhaMers tree) it
restates.  Parity pinning: ``tools/gen_golden.py`` executes the reference's own
function bodies (AST-extracted from the reference text at generation time, in
the build container only) on seeded inputs and stores the results under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
against those vectors, bit-exact for counts / normalised rows and to 1e-12 for
the float scores.  Third-party arithmetic the reference delegates to
scikit-learn (unpinned in requirements.txt:4):
  * KNeighborsClassifier(n_neighbors=3) -- restated here as brute-force
    Euclidean k-NN with a uniform majority vote (the published algorithm); the
    golden vectors for it were produced by scikit-learn 1.7.2 through the
    reference's own call site (scripts/learning.py:127-128).
  * KMeans(n_clusters=86, random_state=10) -- version dependent; centroids are
    an explicit *input* here and the golden vectors carry the centroids that
    scikit-learn 1.7.2 produced through scripts/learning.py:138.
"""
import math

import numpy as np

DNA = 'ATGC'  # scripts/kmer.py:28  (A=0, T=1, G=2, C=3 -- NOT ACGT)


# --------------------------------------------------------------------------
# k-mer counting  (scripts/kmer.py)
# --------------------------------------------------------------------------
def sequence_to_codes(sequence, symbols=DNA):
    """scripts/kmer.py:183-196 restated on integers: every character that is
    not (case-sensitively) one of ``symbols`` becomes invalid (-1, the
    reference's '-'), symbol i becomes code i."""
    lut = np.full(256, -1, dtype=np.int8)
    for i, s in enumerate(symbols):
        lut[ord(s)] = i
    if isinstance(sequence, str):
        raw = np.frombuffer(sequence.encode('latin-1', 'replace'), dtype=np.uint8)
    else:
        raw = np.frombuffer(bytes(sequence), dtype=np.uint8)
    return lut[raw]


def sequence_to_integers(sequence, symbols=DNA):
    """scripts/kmer.py:183-196, string form ('0123' digits and '-')."""
    codes = sequence_to_codes(sequence, symbols)
    chars = np.array(list('0123456789'[:len(symbols)]) + ['-'])
    return ''.join(chars[codes])


def count_string_literal(sequence, kmer_length, symbols=DNA, normalize=False):
    """scripts/kmer.py:32-50,77-79 -- the literal sliding-window loop (one
    Python iteration per window), used as the 'reference-equivalent, 1 core'
    CPU baseline.  First base is the most significant base-4 digit."""
    assert len(symbols) < 10, "only the integer-replacement branch is restated"
    s = sequence_to_integers(sequence, symbols)
    n_sym = len(symbols)
    out = np.zeros(pow(n_sym, kmer_length), dtype=(int, float)[normalize])
    for i in range(len(s) - kmer_length + 1):
        w = s[i:i + kmer_length]
        if '-' not in w:
            out[int(w, n_sym)] += 1
    if normalize and np.sum(out) > 0:
        out = normalize_counts(out)
    return out


def count_string(sequence, kmer_length, symbols=DNA, normalize=False):
    """scripts/kmer.py:32-50,77-79 vectorised: rolling base-|symbols| index +
    bincount.  Bit-identical to :func:`count_string_literal`."""
    n_sym = len(symbols)
    k = int(kmer_length)
    D = pow(n_sym, k)
    codes = sequence_to_codes(sequence, symbols).astype(np.int64)
    L = codes.shape[0]
    dtype = (int, float)[normalize]
    if L - k + 1 <= 0:
        return np.zeros(D, dtype=dtype)
    nwin = L - k + 1
    idx = np.zeros(nwin, dtype=np.int64)
    bad = np.zeros(nwin, dtype=bool)
    for j in range(k):
        c = codes[j:j + nwin]
        bad |= c < 0
        idx = idx * n_sym + np.where(c < 0, 0, c)
    out = np.bincount(idx[~bad], minlength=D).astype(dtype)
    if normalize and np.sum(out) > 0:
        out = normalize_counts(out)
    return out


def count(data, kmer_length, symbols=DNA, normalize=False):
    """scripts/kmer.py:82-111 -- dispatcher: str -> 1-D; list of one -> 1-D;
    list of n -> (n, D); anything else -> None."""
    if isinstance(data, list):
        if len(data) == 1:
            return count(data[0], kmer_length, symbols=symbols, normalize=normalize)
        out = np.zeros((len(data), pow(len(symbols), kmer_length)), dtype=(int, float)[normalize])
        for i, seq in enumerate(data):
            out[i, :] = count_string(seq, kmer_length, symbols=symbols, normalize=normalize)
        return out
    elif isinstance(data, str):
        return count_string(data, kmer_length, symbols=symbols, normalize=normalize)
    return None


def normalize_counts(counts):
    """scripts/kmer.py:209-221 -- float64 copy, each row divided by its sum;
    a zero row gives NaN (no guard in the 2-D branch)."""
    counts = np.asarray(counts).astype(float)
    with np.errstate(invalid='ignore', divide='ignore'):
        if counts.ndim == 1:
            counts = counts / np.sum(counts)
        else:
            for i in range(counts.shape[0]):
                counts[i, :] /= np.sum(counts[i, :])
    return counts


def kmers(k, symbols=DNA):
    """scripts/kmer.py:224-251 -- k-mer labels in bin order (first base most
    significant)."""
    mers = ['']
    for _ in range(k):
        mers = [m + s for m in mers for s in symbols]
    return mers


# --------------------------------------------------------------------------
# distances / centroids  (scripts/learning.py)
# --------------------------------------------------------------------------
def distances(vector, data):
    """scripts/learning.py:47-56 -- direct-difference Euclidean distances."""
    vector = np.asarray(vector, dtype=float)
    if vector.ndim == 1:
        vector = vector[None, :]
    return np.linalg.norm(np.repeat(vector, data.shape[0], axis=0) - data, axis=1)


def closest_to(point, picks):
    """scripts/learning.py:59-66 -- argmin (first index wins ties)."""
    return picks[np.argmin(distances(point, picks))]


def get_centroids(data, assignment):
    """scripts/learning.py:69-81 -- mean of member rows per sorted label."""
    labels = sorted(set(assignment) - set([-1]))
    return np.array([np.mean(data[assignment == c], axis=0) for c in labels])


def knn(queries, ref_data, ref_labels, k=3, return_neighbors=False, chunk=512, budget_bytes=1 << 30):
    """scripts/learning.py:118-128 -- brute-force Euclidean k-NN, uniform
    majority vote over labels {0,1}, returned as 2*(pred-0.5) in {-1,+1}.
    Distances are direct differences in float64 (squared); ties go to the
    lower reference index.  Queries and reference rows are walked in blocks
    sized to ``budget_bytes`` of temporaries (each (query, row) distance is
    one sum over the dimensions whatever the blocking)."""
    Q = np.asarray(queries, dtype=float)
    R = np.asarray(ref_data, dtype=float)
    lab = np.asarray(ref_labels, dtype=float)
    N, D = Q.shape
    M = R.shape[0]
    rblock = max(1, min(M, budget_bytes // (8 * D)))
    chunk = max(1, min(chunk, budget_bytes // (8 * D * rblock)))
    nbr = np.zeros((N, k), dtype=np.int64)
    nd = np.zeros((N, k), dtype=float)
    for s in range(0, N, chunk):
        q = Q[s:s + chunk]
        d2 = np.empty((q.shape[0], M))
        for r0 in range(0, M, rblock):
            d2[:, r0:r0 + rblock] = ((q[:, None, :] - R[None, r0:r0 + rblock, :]) ** 2).sum(axis=2)
        order = np.argsort(d2, axis=1, kind='stable')[:, :k]
        nbr[s:s + chunk] = order
        nd[s:s + chunk] = np.take_along_axis(d2, order, axis=1)
    votes = lab[nbr].sum(axis=1)
    pred = (votes * 2 > k).astype(float)       # majority of k labels in {0,1}
    scores = 2 * (pred - 0.5)
    if return_neighbors:
        return scores, nbr, np.sqrt(nd)
    return scores


# --------------------------------------------------------------------------
# scoring  (scripts/phamer.py)
# --------------------------------------------------------------------------
def equalize_reference_data(positive, negative):
    """scripts/phamer.py:159-175 -- truncate both to the first min(n+, n-) rows."""
    n = min(positive.shape[0], negative.shape[0])
    return positive[:n], negative[:n]


def proximity_metric(point, nearest_positive, nearest_negative):
    """scripts/phamer.py:198-210."""
    e_neg = np.linalg.norm(point - nearest_negative)
    e_pos = np.linalg.norm(point - nearest_positive)
    with np.errstate(invalid='ignore', divide='ignore'):
        return np.tanh((e_neg - e_pos) / (e_pos + e_neg))


def centroid_score_points(points, positive_centroids, negative_centroids):
    """scripts/phamer.py:250-256 -- the per-point loop of kmeans_score_points,
    with the centroids (scripts/phamer.py:245-248) taken as inputs."""
    points = np.asarray(points, dtype=float)
    scores = np.zeros(points.shape[0])
    for i in range(points.shape[0]):
        p = points[i]
        scores[i] = proximity_metric(p, closest_to(p, positive_centroids),
                                     closest_to(p, negative_centroids))
    return scores


def centroid_score_points_fast(points, positive_centroids, negative_centroids, chunk=2048):
    """Vectorised form of :func:`centroid_score_points` (same arithmetic per
    element up to summation order inside numpy's norm)."""
    P = np.asarray(points, dtype=float)
    out = np.zeros(P.shape[0])
    for s in range(0, P.shape[0], chunk):
        p = P[s:s + chunk]
        dp = np.sqrt(((p[:, None, :] - positive_centroids[None]) ** 2).sum(axis=2)).min(axis=1)
        dn = np.sqrt(((p[:, None, :] - negative_centroids[None]) ** 2).sum(axis=2)).min(axis=1)
        with np.errstate(invalid='ignore', divide='ignore'):
            out[s:s + chunk] = np.tanh((dn - dp) / (dp + dn))
    return out


def knn_score_points(points, positive, negative, k_neighbors=3):
    """scripts/phamer.py:186-187,268-273 -- train = vstack(pos, neg),
    labels = ones(n+) ++ zeros(n-)."""
    train = np.vstack((positive, negative))
    labels = np.append(np.ones(positive.shape[0]), np.zeros(negative.shape[0]))
    return knn(points, train, labels, k=k_neighbors)


def score_points(points, positive, negative, method='combo', k_neighbors=3,
                 positive_centroids=None, negative_centroids=None):
    """scripts/phamer.py:177-195,303-313,451-468 for the in-scope methods.
    'kmeans' / 'combo' need the centroids the caller's k-means produced."""
    if method == 'knn':
        return knn_score_points(points, positive, negative, k_neighbors)
    if method == 'kmeans':
        return centroid_score_points(points, positive_centroids, negative_centroids)
    if method == 'combo':
        return (knn_score_points(points, positive, negative, k_neighbors)
                + centroid_score_points(points, positive_centroids, negative_centroids))
    raise ValueError("method %r is outside the restated path" % (method,))


# --------------------------------------------------------------------------
# deterministic Lloyd k-means (restates phamers_amd/csrc/kmeans.hip; NOT a reference function --
# the reference uses scikit-learn, scripts/learning.py:138)
# --------------------------------------------------------------------------
# ---- FASTA header -> id (host strings either side of the path) -------------------------------------
def _accession_like(word):
    """scripts/id_parser.py:80-86: not a number and a '.' second to last (IndexError below two characters)."""
    try:
        float(word)
        return False
    except ValueError:
        return word[-2] == '.'


def get_id(header):
    """id_parser.get_id restated (scripts/id_parser.py:89-100; contig rule :18-30, phage rule :71-77, bacteria
    rule :57-68).  Raises IndexError / returns None exactly where the reference does."""
    if '_ID_' in header:
        fields = header.strip().replace('>', '').split('_')
        return fields[fields.index('ID') + 1].replace('-circular', '')
    if header.count('|') == 4:
        return header.split('|')[3].replace('>', '')
    first = header.split(' ')[0]
    if _accession_like(first):
        return first
    second = header.split('\t')[1].replace('>', '')
    return second if _accession_like(second) else None


# ---- count-vector transforms (scripts/transform_kmers.py:21-88) -------------------------------------
def reference_transform_indices(k, reverse, complement, num_symbols=4):
    """The index table the reference's get_transformed_indicies really builds (scripts/transform_kmers.py:21-47,
    Python 2 integer division).  NOT a permutation: decompose() lists k + 1 digits most significant first -- a
    leading zero and then the k digits -- of which positions 0 .. k-1 are used (so the LAST digit is dropped), and
    the multipliers are k**i ascending rather than num_symbols**i descending."""
    b = num_symbols
    comp = {0: 1, 1: 0, 2: 3, 3: 2}
    out = np.zeros(b ** k, dtype=np.int64)
    for j in range(b ** k):
        digits = [(j % b ** (i + 1) - j % b ** i) // b ** i for i in range(k, -1, -1)]   # k + 1 digits
        picked = [digits[p] for p in (range(k - 1, -1, -1) if reverse else range(k))]
        if complement:
            picked = [comp[d] for d in picked]
        out[j] = sum(d * k ** i for i, d in enumerate(picked))
    return out


def transform_kmers(counts, reverse=True, complement=False):
    """scripts/transform_kmers.py:68-88: gather of count columns by the reference's table (IndexError when the
    table reaches past the last column, which it does for k >= 5)."""
    counts = np.asarray(counts)
    if not reverse and not complement:
        return counts
    k = int(round(math.log(counts.shape[1], 4)))
    return counts.T[reference_transform_indices(k, reverse, complement)].T


def _splitmix64(x):
    m = (1 << 64) - 1
    z = (x + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def kmeans_pp_seeds(X, k, seed=10):
    """The k initial centres of kmeans_lloyd: k-means++ seeding by inverse CDF in index order,
    u_j = splitmix64(seed + j) / 2^64 on 53 bits."""
    X = np.asarray(X, dtype=float)
    n, D = X.shape
    u = lambda j: float(_splitmix64(seed + j) >> 11) / 9007199254740992.0
    centres = np.zeros((k, D))
    c0 = min(int(u(0) * n), n - 1)
    centres[0] = X[c0]
    mind2 = ((X - centres[0]) ** 2).sum(axis=1)
    for j in range(1, k):
        # slice sums in index order (256 contiguous slices), then the inverse-CDF walk
        per = (n + 255) // 256
        parts = [sum(mind2[per * t: min(per * t + per, n)].tolist(), 0.0) for t in range(256)]
        total = 0.0
        for p in parts:
            total += p
        target = u(j) * total
        run, pick, done = 0.0, n - 1, False
        for t in range(256):
            if done:
                break
            if run + parts[t] > target:
                lo, hi = per * t, min(per * t + per, n)
                for i in range(lo, hi):
                    run += mind2[i]
                    if run > target:
                        pick, done = i, True
                        break
                if not done:
                    pick, done = (hi - 1 if hi else 0), True
            else:
                run += parts[t]
        centres[j] = X[pick]
        mind2 = np.minimum(mind2, ((X - centres[j]) ** 2).sum(axis=1))
    return centres


def kmeans_lloyd(X, k, seed=10, max_iter=300):
    """k-means++ seeding (kmeans_pp_seeds) + Lloyd sweeps with ties to the lower index, index-ordered means,
    farthest-point re-seeding of empty clusters.  Returns (labels, centroids, sweeps)."""
    X = np.asarray(X, dtype=float)
    n, D = X.shape
    centres = kmeans_pp_seeds(X, k, seed)
    labels = np.full(n, -1, dtype=np.int64)
    sweeps = 0
    for it in range(max_iter):
        d2 = np.stack([((X - centres[c]) ** 2).sum(axis=1) for c in range(k)], axis=1)
        new = np.argmin(d2, axis=1)                      # first minimum = lower centre index
        changed = int((new != labels).sum())
        labels = new
        own = d2[np.arange(n), labels]
        sizes = np.bincount(labels, minlength=k)
        for c in range(k):
            if sizes[c]:
                centres[c] = X[labels == c].sum(axis=0) / sizes[c]
        for c in range(k):
            if sizes[c] == 0:
                ok = sizes[labels] > 1
                if not ok.any():
                    continue
                cand = np.where(ok, own, -1.0)
                far = int(np.argmax(cand))
                sizes[labels[far]] -= 1
                labels[far] = c
                sizes[c] = 1
                own[far] = 0.0
                centres[c] = X[far]
                changed += 1
        sweeps = it + 1
        if changed == 0:
            break
    return labels, centres, sweeps


def kmeans_lloyd_seeded(X, init, tol_abs, max_iter=300):
    """scikit-learn's Lloyd iteration (sklearn/cluster/_kmeans.py, _kmeans_single_lloyd -- what KMeans.fit, and so
    scripts/learning.py:138, runs after its seeding) restated with float64 direct differences: E-step against the current
    centres (ties to the lower centre), M-step (member means), stop when no label changed or when the summed squared
    centre shift is <= tol_abs; in the second case one more E-step.  Returns (labels, sweeps, n_empty): the statement
    phk_kmeans_lloyd is tested against (an empty cluster keeps its centre and is counted; scikit-learn relocates it)."""
    X = np.asarray(X, dtype=float)
    centres = np.array(init, dtype=float)
    k = centres.shape[0]
    labels_old = np.full(X.shape[0], -1, dtype=np.int64)
    strict, sweeps, empties = False, 0, 0
    for it in range(max_iter):
        d2 = np.stack([((X - centres[c]) ** 2).sum(axis=1) for c in range(k)], axis=1)
        labels = np.argmin(d2, axis=1)
        new = centres.copy()
        for c in range(k):
            members = labels == c
            if members.any():
                new[c] = X[members].sum(axis=0) / members.sum()
            else:
                empties += 1
        shift = float(((new - centres) ** 2).sum())
        centres = new
        sweeps = it + 1
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if shift <= tol_abs:
            break
        labels_old = labels
    if not strict:
        d2 = np.stack([((X - centres[c]) ** 2).sum(axis=1) for c in range(k)], axis=1)
        labels = np.argmin(d2, axis=1)
    return labels, sweeps, empties
