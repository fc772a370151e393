"""
cross_validate.py -- N-fold cross validation of the reference matrices (the other caller of the scoring path,
PhaMers' scripts/cross_validate.py:38-101) as a batched service on ONE resident GPU model.

The reference rebuilds everything per fold: it slices the train rows out of the matrices, refits k-means, fits a
k-NN classifier and scores the held-out rows -- N times, although the N train sets share (N-1)/N of their rows.
Here the full reference matrix is uploaded and prepared ONCE (`_lib.Model` over every positive and negative row); a
fold is then
    * a column mask over that model's train rows (the held-out rows are excluded from the k-NN search),
    * the fold's centroids, written into the model's centroid segments (k-means on the fold's train rows: scikit-learn
      by default, as the reference; ``kmeans='gpu'`` selects the deterministic device k-means per fold),
    * one scoring call for the held-out rows.
Scores are those of a model built from the fold's train rows alone (the k-NN search is translation invariant: only
the error bounds depend on the centring vector, and they are evaluated for the one in use);
tests/golden/cross_validation.npz holds what the reference's own cross_validate produced.

    validator = cross_validator(); validator.positive_data = ...; validator.negative_data = ...
    positive_scores, negative_scores = validator.cross_validate()

The reference shuffles the fold assignment with the unseeded global NumPy generator (scripts/cross_validate.py:71-78);
``seed`` makes a run reproducible with the same draws (np.random.seed(seed) right before its two shuffles).  A custom
``scoring_function`` (anything but phamer.score_points) is honoured with the reference's per-fold calls.
"""
import logging

import numpy as np

from . import _lib
from . import learning
from . import phamer

logger = logging.getLogger(__name__)
logger.setLevel(logging.WARNING)


class FoldPlan(object):
    """Which fold every positive / negative row is held out in."""

    def __init__(self, n_positive, n_negative, folds, seed=None):
        self.folds = int(folds)
        self.positive = np.arange(n_positive) % self.folds
        self.negative = np.arange(n_negative) % self.folds
        rng = np.random if seed is None else np.random.RandomState(seed)   # RandomState(seed) == np.random.seed(seed)
        rng.shuffle(self.positive)
        rng.shuffle(self.negative)

    def held_out(self, fold):
        return self.positive == fold, self.negative == fold


class cross_validator(object):

    def __init__(self):
        self.positive_data = self.negative_data = None
        self.positive_ids = self.negative_ids = None
        self.positive_scores = self.negative_scores = None
        self.equalize_reference = False
        self.N = 20                                  # scripts/cross_validate.py:48
        self.method = 'combo'
        self.scoring_function = phamer.score_points  # scripts/cross_validate.py:275
        self.seed = None
        self.kmeans = 'sklearn'                      # or 'gpu': deterministic device k-means per fold
        self.k_clusters = 86                         # scripts/phamer.py:78
        self.k_neighbors = 3                         # scripts/phamer.py:79

    # ---- the reference's entry point ----------------------------------------------------------------------
    def cross_validate(self):
        """Every reference row scored by a model that has not seen its fold (scripts/cross_validate.py:57-101).
        Returns (positive_scores, negative_scores)."""
        self._equalize()
        self.num_positive, self.num_negative = self.positive_data.shape[0], self.negative_data.shape[0]
        plan = FoldPlan(self.num_positive, self.num_negative, self.N, self.seed)
        self.positive_assignment, self.negative_assignment = plan.positive, plan.negative
        resident = self.scoring_function is phamer.score_points and (self.method or 'combo') in ('knn', 'kmeans', 'combo')
        runner = self._folds_on_resident_model if resident else self._folds_through_scoring_function
        self.positive_scores, self.negative_scores = runner(plan)
        logger.info("%d-fold cross validation complete." % self.N)
        return self.positive_scores, self.negative_scores

    def _equalize(self):
        """First min(n+, n-) rows of each class when asked to (scripts/cross_validate.py:63-69)."""
        if not self.equalize_reference:
            return
        m = min(self.positive_data.shape[0], self.negative_data.shape[0])
        self.positive_data, self.negative_data = self.positive_data[:m], self.negative_data[:m]
        self.positive_ids = None if self.positive_ids is None else self.positive_ids[:m]
        self.negative_ids = None if self.negative_ids is None else self.negative_ids[:m]

    # ---- batched: one model, a fold = mask + centroids ---------------------------------------------------------
    def _fold_centroids(self, train_rows):
        if self.kmeans == 'gpu':
            return learning.kmeans_gpu(train_rows, self.k_clusters)[1]
        return learning.get_centroids(train_rows, learning.kmeans(train_rows, self.k_clusters))

    def _folds_on_resident_model(self, plan):
        method = self.method or 'combo'
        P = np.ascontiguousarray(self.positive_data, dtype=np.float64)
        Nm = np.ascontiguousarray(self.negative_data, dtype=np.float64)
        if np.isnan(P).any() or np.isnan(Nm).any():
            raise ValueError("Input contains NaN.")
        pos_scores, neg_scores = np.zeros(len(P)), np.zeros(len(Nm))
        with_centroids = method != 'knn'
        model = None
        ctx = _lib.get_context()
        self.model_uploads = 0
        try:
            for fold in range(plan.folds):
                logger.info('Iteration %d/%d' % (1 + fold, plan.folds))
                out_p, out_n = plan.held_out(fold)
                cents = None
                if with_centroids:
                    cents = self._fold_centroids(P[~out_p]), self._fold_centroids(Nm[~out_n])
                if model is not None and with_centroids and (
                        len(cents[0]), len(cents[1])) != (self._model_centroids[0], self._model_centroids[1]):
                    model.close()      # a fit that came back with another number of clusters: rebuild (rare)
                    model = None
                if model is None:
                    model = _lib.Model(ctx, P, Nm, cents[0] if cents else None, cents[1] if cents else None,
                                       k_neighbors=self.k_neighbors)
                    self._model_centroids = (len(cents[0]), len(cents[1])) if cents else (0, 0)
                    self.model_uploads += 1
                elif with_centroids:
                    model.set_centroids(*cents)
                model.set_column_mask(np.concatenate((out_p, out_n)))
                scores = model.score(np.vstack((P[out_p], Nm[out_n])), method)
                n_p = int(out_p.sum())
                pos_scores[out_p], neg_scores[out_n] = scores[:n_p], scores[n_p:]
        finally:
            if model is not None:
                model.close()
        return pos_scores, neg_scores

    # ---- generic: any scoring function, the reference's per-fold calls ---------------------------------------
    def _folds_through_scoring_function(self, plan):
        pos_scores, neg_scores = np.zeros(self.num_positive), np.zeros(self.num_negative)
        for fold in range(plan.folds):
            out_p, out_n = plan.held_out(fold)
            scores = self.scoring_function(np.vstack((self.positive_data[out_p], self.negative_data[out_n])),
                                           self.positive_data[~out_p], self.negative_data[~out_n], method=self.method)
            n_p = int(out_p.sum())
            pos_scores[out_p], neg_scores[out_n] = scores[:n_p], scores[n_p:]
        return pos_scores, neg_scores
