"""The CPU oracle (oracle/oracle.py + oracle/kmer_oracle.c) against the golden
vectors produced by the reference's own function bodies (tools/gen_golden.py).
CPU-only."""
import numpy as np
import pytest

from oracle import oracle, c_oracle
from tests import helpers


@pytest.fixture(scope="module")
def counts_golden():
    return helpers.load_npz("counts.npz")


def _case_items(counts_golden):
    seqs, doc = helpers.count_cases()
    for key, want in counts_golden.items():
        name, _, ktag = key.rpartition("__k")
        if name in seqs:
            yield key, seqs[name], int(ktag), want


def test_count_string_vectorised_matches_reference(counts_golden):
    n = 0
    for key, seq, k, want in _case_items(counts_golden):
        got = oracle.count_string(seq, k)
        assert got.dtype == np.int64 and np.array_equal(got, want), key
        n += 1
    assert n > 100


def test_count_string_literal_matches_reference(counts_golden):
    for key, seq, k, want in _case_items(counts_golden):
        if len(seq) <= 5000:
            assert np.array_equal(oracle.count_string_literal(seq, k), want), key


def test_c_oracle_count_matches_reference(counts_golden):
    for key, seq, k, want in _case_items(counts_golden):
        assert np.array_equal(c_oracle.count(seq, k)[0], want), key


def test_known_answer_and_invariants(counts_golden):
    # scripts/kmer.py:89-91: 'AAAT' is at index 1 for DNA
    assert counts_golden["kat_AAAT__k4"].nonzero()[0].tolist() == [1]
    # mixed_invalid: 7 counted windows (N and lower case windows dropped)
    assert counts_golden["mixed_invalid__k4"].sum() == 7
    # len < k and empty -> zeros
    assert counts_golden["len3__k4"].sum() == 0 and counts_golden["empty__k4"].sum() == 0


def test_count_list_shapes(counts_golden):
    seqs, doc = helpers.count_cases()
    lst = [seqs[n] for n in doc["list5"]]
    got = oracle.count(lst, 4)
    assert got.shape == (5, 256) and np.array_equal(got, counts_golden["list5__k4"])
    one = oracle.count([lst[0]], 4)
    assert one.shape == (256,) and np.array_equal(one, counts_golden["list1__k4"])
    assert oracle.count(12345, 4) is None          # scripts/kmer.py:108-110
    assert np.array_equal(c_oracle.count(lst, 4), counts_golden["list5__k4"])


def test_count_string_normalize_guard(counts_golden):
    a = oracle.count_string("ATGCATGCNATGCatgcATGC", 4, normalize=True)
    assert np.array_equal(a, counts_golden["norm_mixed_invalid__k4"])
    z = oracle.count_string("N" * 40, 4, normalize=True)      # guard: stays zeros, no NaN
    assert np.array_equal(z, counts_golden["norm_all_N__k4"]) and not np.isnan(z).any()


def test_labels():
    doc = helpers.load_json("labels.json")
    for k, want in doc["kmers"].items():
        assert oracle.kmers(int(k)) == want
    assert oracle.kmers(4)[:8] == doc["kmers4_first8"] and oracle.kmers(4)[-4:] == doc["kmers4_last4"]
    for s, want in doc["sequence_to_integers"].items():
        assert oracle.sequence_to_integers(s) == want
    mers = oracle.kmers(4)
    for s, want in doc["get_kmer_index"].items():
        assert mers.index(s) == want


def test_normalize_bit_exact():
    g = helpers.load_npz("normalize.npz")
    for fn in (oracle.normalize_counts, c_oracle.normalize):
        got = fn(g["in2d"])
        assert got.dtype == np.float64
        assert np.array_equal(got.view(np.uint64), g["out2d"].view(np.uint64)) or \
            np.array_equal(np.isnan(got), np.isnan(g["out2d"])) and \
            np.array_equal(got[~np.isnan(got)], g["out2d"][~np.isnan(g["out2d"])])
        assert np.isnan(got[3]).all()                 # zero row -> NaN
    got1 = oracle.normalize_counts(g["in1d"])
    assert np.array_equal(got1, g["out1d"])


@pytest.mark.parametrize("tag", ["eq", "full"])
def test_scoring_k4_matches_reference(tag):
    g = helpers.load_npz("scoring_k4.npz")
    ref = helpers.load_npz("ref_features.npz")
    pos = oracle.normalize_counts(ref["pos_counts"].astype(np.int64))
    neg = oracle.normalize_counts(ref["neg_counts"].astype(np.int64))
    if tag == "eq":
        pos, neg = oracle.equalize_reference_data(pos, neg)
        assert [pos.shape[0], neg.shape[0]] == g["n_equalized"].tolist()
    q = oracle.normalize_counts(g["q_counts"])
    assert np.array_equal(q, g["q"])
    knn = oracle.knn_score_points(q, pos, neg, 3)
    assert np.array_equal(knn, g["knn_" + tag])
    cen = oracle.centroid_score_points(q, g["cpos_" + tag], g["cneg_" + tag])
    assert helpers.rel_err(cen, g["kmeans_" + tag]) < 1e-10
    combo = oracle.score_points(q, pos, neg, "combo", 3, g["cpos_" + tag], g["cneg_" + tag])
    assert helpers.rel_err(combo, g["combo_" + tag]) < 1e-10
    assert helpers.rel_err(oracle.centroid_score_points_fast(q, g["cpos_" + tag], g["cneg_" + tag]),
                           g["kmeans_" + tag]) < 1e-10
    # C restatement
    train = np.vstack((pos, neg))
    labels = np.append(np.ones(pos.shape[0]), np.zeros(neg.shape[0]))
    assert np.array_equal(c_oracle.knn_score(q, train, labels, 3), g["knn_" + tag])
    assert helpers.rel_err(c_oracle.centroid_score(q, g["cpos_" + tag], g["cneg_" + tag]),
                           g["kmeans_" + tag]) < 1e-10
    # neighbour indices agree with scikit-learn's (near-ties visible in nbr_dist)
    _, nbr, dist = oracle.knn(q, train, labels, 3, return_neighbors=True)
    assert np.array_equal(nbr, g["nbr_idx_" + tag][:, :3])
    assert helpers.rel_err(dist, g["nbr_dist_" + tag][:, :3]) < 1e-9


def test_scoring_other_neighbour_counts_and_adversarial():
    g = helpers.load_npz("scoring_k4.npz")
    ref = helpers.load_npz("ref_features.npz")
    pos = oracle.normalize_counts(ref["pos_counts"].astype(np.int64))
    neg = oracle.normalize_counts(ref["neg_counts"].astype(np.int64))
    for kn in (1, 5, 7):
        assert np.array_equal(oracle.knn_score_points(g["q"], pos, neg, kn), g["knn_full_kn%d" % kn])
    adv = g["adv_q"]
    assert np.array_equal(oracle.knn_score_points(adv, pos, neg, 3), g["adv_knn_full"])
    cen = oracle.centroid_score_points(adv, g["cpos_full"], g["cneg_full"])
    assert helpers.rel_err(cen, g["adv_kmeans_full"]) < 1e-10


def test_scoring_highdim_matches_reference():
    g = helpers.load_npz("scoring_highdim.npz")
    for tag in ("k5", "k6"):
        q, p, n = g["q_" + tag], g["pos_" + tag], g["neg_" + tag]
        assert np.array_equal(oracle.knn_score_points(q, p, n, 3), g["knn_" + tag])
        cen = oracle.centroid_score_points(q, g["cpos_" + tag], g["cneg_" + tag])
        assert helpers.rel_err(cen, g["kmeans_" + tag]) < 1e-10


def test_reference_matrix_marginal_invariant():
    """in the shipped 4-mer rows the 3-mer prefix marginal and 3-mer suffix marginal
    differ by a small even L1 amount -- confirms the first-base-most-significant layout."""
    ref = helpers.load_npz("ref_features.npz")
    t = ref["pos_counts"].astype(np.int64).reshape(-1, 4, 4, 4, 4)
    l1 = np.abs(t.sum(axis=4) - t.sum(axis=1)).sum(axis=(1, 2, 3))
    assert (l1 % 2 == 0).all() and np.median(l1) == 2
    assert ref["pos_counts"].shape == (2255, 256) and ref["neg_counts"].shape == (2418, 256)


def test_distances_and_closest_to_golden():
    """oracle.distances / closest_to against the reference's learning.distances / closest_to (scripts/learning.py:47-66)
    executed on rows of the real matrix (tests/golden/distances.npz): bit for bit -- same NumPy expression."""
    from oracle import oracle
    g = helpers.load_npz("distances.npz")
    ref = helpers.load_npz("ref_features.npz")
    pos = oracle.normalize_counts(ref["pos_counts"].astype(np.int64))[:300]
    for i, v in enumerate(g["queries"]):
        assert np.array_equal(oracle.distances(v, pos), g["dist_pos"][i])
        assert np.array_equal(oracle.closest_to(v, g["picks"]), g["closest_picks"][i])
    assert np.array_equal(oracle.distances(g["queries"][3:4], pos), g["dist_row_2d"])
    assert g["dist_pos"][10, 0] == 0.0 and g["closest_idx"][14] == 43   # a reference row itself; the first of two equal picks


def test_oracle_lloyd_kmeans_agrees_with_scikit_learn_from_the_same_seeds():
    """oracle.kmeans_lloyd (the statement the device k-means phk_kmeans is tested against) is not a reference function:
    it is checked here against an independent implementation -- scikit-learn's Lloyd iteration started from the same
    initial centres (oracle.kmeans_pp_seeds), run to a fixed point (tol = 0): same labels, same centroids."""
    from sklearn.cluster import KMeans
    from oracle import oracle
    ref = helpers.load_npz("ref_features.npz")
    pos = oracle.normalize_counts(ref["pos_counts"].astype(np.int64))
    for rows, k, seed in ((700, 12, 10), (2255, 86, 10), (400, 7, 3)):
        X = pos[:rows]
        seeds = oracle.kmeans_pp_seeds(X, k, seed)
        labels, cents, sweeps = oracle.kmeans_lloyd(X, k, seed)
        sk = KMeans(n_clusters=k, init=seeds, n_init=1, algorithm="lloyd", tol=0.0, max_iter=300).fit(X)
        assert len(np.unique(labels)) == k                     # no empty cluster on this data: the plain Lloyd iteration
        assert np.array_equal(sk.labels_, labels), (rows, k)
        assert np.allclose(sk.cluster_centers_, cents, rtol=1e-10, atol=1e-14)
        assert 1 < sweeps < 300


def test_seeded_lloyd_restatement_reproduces_the_reference_kmeans_fit():
    """The route the product takes for learning.kmeans: scikit-learn's own k-means++ seeding (kmeans_plusplus on the
    mean-centred rows, RandomState(10) -- what KMeans.fit does first) followed by oracle.kmeans_lloyd_seeded, the statement
    of scikit-learn's Lloyd iteration and stopping rule that phk_kmeans_lloyd is tested against.  On the reference
    matrices it must give the labels of KMeans(n_clusters=86, random_state=10).fit(X) (scripts/learning.py:138) and so,
    through get_centroids (scripts/learning.py:69-81), the centroids the golden scores were generated with."""
    from sklearn.cluster import KMeans, kmeans_plusplus
    from oracle import oracle
    ref = helpers.load_npz("ref_features.npz")
    g = helpers.load_npz("scoring_k4.npz")
    pos = oracle.normalize_counts(ref["pos_counts"].astype(np.int64))
    neg = oracle.normalize_counts(ref["neg_counts"].astype(np.int64))
    n = int(np.asarray(g["n_equalized"]).ravel()[0])
    for X, want in ((pos[:n], g["cpos_eq"]), (neg[:n], g["cneg_eq"]), (neg, g["cneg_full"])):
        Xc = X - X.mean(axis=0)
        init, _ = kmeans_plusplus(Xc, 86, random_state=np.random.RandomState(10))
        labels, sweeps, empties = oracle.kmeans_lloyd_seeded(Xc, init, float(np.mean(np.var(Xc, axis=0)) * 1e-4))
        sk = KMeans(n_clusters=86, random_state=10).fit(X)
        assert empties == 0 and sweeps == sk.n_iter_
        assert np.array_equal(labels, sk.labels_)
        cents = np.array([np.mean(X[labels == c], axis=0) for c in sorted(set(labels))])
        assert np.allclose(cents, want, rtol=0, atol=1e-10)
