"""
phamer.py -- drop-in for the scoring entry points of PhaMers' scripts/phamer.py.

    phamer_scorer (attributes + score_points / knn_ / kmeans_ / combo_score_points,
                   equalize_reference_data)                     scripts/phamer.py:42-313
    score_points(scoring_data, positive_training_data,
                 negative_training_data, method=None)           scripts/phamer.py:451-468

The distance / vote / proximity arithmetic runs on the GPU (libphamers_hip.so); the k-means
fit that yields the centroids is scikit-learn's, as in the reference.  Methods outside
{knn, kmeans, combo} raise NotImplementedError (dbscan / svm / density / silhouette are
out of the accelerated path).
"""
import logging

import numpy as np

from . import _lib
from . import kmer
from . import learning

logging.basicConfig(format='[%(asctime)s][%(levelname)s][%(funcName)s] - %(message)s')
logger = logging.getLogger(__name__)
logger.setLevel(logging.WARNING)


class phamer_scorer(object):

    def __init__(self):
        # the attributes of the reference object that the hot path reads (scripts/phamer.py:59-79)
        self.data_ids = None
        self.data_points = None
        self.positive_ids = None
        self.positive_data = None
        self.negative_ids = None
        self.negative_data = None

        self.length_requirement = 5000

        self.scoring_method = 'combo'
        self.all_scoring_methods = ['dbscan', 'kmeans', 'knn', 'svm', 'density', 'silhouette', 'combo']
        self.method_function_map = {
            'kmeans': self.kmeans_score_points, 'knn': self.knn_score_points,
            'combo': self.combo_score_points,
            'dbscan': self._outside_path, 'svm': self._outside_path,
            'density': self._outside_path, 'silhouette': self._outside_path,
        }

        self.kmer_length = 4
        self.k_clusters = 86
        self.k_neighbors = 3

        # centroids of the last kmeans / combo call (captured for inspection and tests)
        self.positive_centroids = None
        self.negative_centroids = None

    def _outside_path(self):
        raise NotImplementedError("scoring method %r is outside the accelerated path; "
                                  "knn / kmeans / combo are available" % (self.scoring_method,))

    def equalize_reference_data(self):
        """Same number of positive and negative rows: the FIRST min(n+, n-) rows of each
        (scripts/phamer.py:159-175)."""
        num_positive = self.positive_data.shape[0]
        num_negative = self.negative_data.shape[0]
        if num_negative == num_positive:
            return
        num_ref = min(num_positive, num_negative)
        logger.debug("Equalizing reference data to: %d data points" % num_ref)
        self.positive_data = self.positive_data[:num_ref]
        self.negative_data = self.negative_data[:num_ref]
        if self.positive_ids is not None:
            self.positive_ids = self.positive_ids[:num_ref]
        if self.negative_ids is not None:
            self.negative_ids = self.negative_ids[:num_ref]
        self.num_positive = num_ref
        self.num_negative = num_ref

    def score_points(self):
        """Scores ``data_points`` against ``positive_data`` / ``negative_data`` with
        ``scoring_method`` (scripts/phamer.py:177-195)."""
        self.num_points = self.data_points.shape[0]
        self.num_positive = self.positive_data.shape[0]
        self.num_negative = self.negative_data.shape[0]
        scoring_function = self.method_function_map[self.scoring_method]
        logger.debug("Scoring %d points. Method: %s..." % (self.data_points.shape[0], self.scoring_method))
        self.scores = np.array(scoring_function())
        return self.scores

    def _fit_centroids(self):
        """scripts/phamer.py:245-248: k-means with k_clusters on each class, then centroids."""
        pa = learning.kmeans(self.positive_data, self.k_clusters)
        na = learning.kmeans(self.negative_data, self.k_clusters)
        self.positive_centroids = learning.get_centroids(self.positive_data, pa)
        self.negative_centroids = learning.get_centroids(self.negative_data, na)

    def _gpu_score(self, method):
        q = np.asarray(self.data_points, dtype=np.float64)
        if np.isnan(q).any():
            # the reference reaches scikit-learn, which raises on NaN input (zero-count contig)
            raise ValueError("Input contains NaN.")
        centroids = (self.positive_centroids, self.negative_centroids) if method != 'knn' else (None, None)
        model = _lib.Model(_lib.get_context(), self.positive_data, self.negative_data,
                           centroids[0], centroids[1], k_neighbors=self.k_neighbors)
        try:
            return model.score(q, method)
        finally:
            model.close()

    def kmeans_score_points(self):
        """scripts/phamer.py:240-256: tanh proximity metric to the nearest centroid of each class."""
        self._fit_centroids()
        return self._gpu_score('kmeans')

    def knn_score_points(self):
        """scripts/phamer.py:268-273."""
        return self._gpu_score('knn')

    def combo_score_points(self):
        """scripts/phamer.py:303-313: knn score + kmeans score (one fused GPU pass)."""
        self._fit_centroids()
        return self._gpu_score('combo')


def score_points(scoring_data, positive_training_data, negative_training_data, method=None):
    """Functional form of phamer_scorer.score_points (scripts/phamer.py:451-468), used by the
    reference's cross-validation (scripts/cross_validate.py:95)."""
    scorer = phamer_scorer()
    if method is not None:
        scorer.scoring_method = method
    scorer.data_points = scoring_data
    scorer.positive_data = positive_training_data
    scorer.negative_data = negative_training_data
    return scorer.score_points()


def score_contigs(sequences, positive_training_data, negative_training_data, kmer_length=4, method='combo'):
    """Convenience: count -> normalise -> score a list of contig strings (what phamer.py's
    load_data + score_points do for a FASTA input, scripts/phamer.py:131,139,579)."""
    counts = kmer.count(list(sequences), kmer_length)
    counts = counts.reshape(-1, 4 ** kmer_length)
    return score_points(kmer.normalize_counts(counts), positive_training_data, negative_training_data, method)
