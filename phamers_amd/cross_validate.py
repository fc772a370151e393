"""
cross_validate.py -- the other caller of the scoring path: N-fold cross validation of the
reference matrices themselves (PhaMers' scripts/cross_validate.py:38-101), every fold scored on the
GPU through phamer.score_points.

    validator = cross_validator(); validator.positive_data = ...; validator.negative_data = ...
    pos_scores, neg_scores = validator.cross_validate()

Fold assignment is ``arange(n) % N`` shuffled (scripts/cross_validate.py:74-78).  The reference
shuffles with the unseeded global NumPy RNG; pass ``seed`` for a reproducible run.
"""
import logging

import numpy as np

from . import phamer

logger = logging.getLogger(__name__)
logger.setLevel(logging.WARNING)


class cross_validator(object):

    def __init__(self):
        self.positive_data = None
        self.negative_data = None
        self.positive_ids = None
        self.negative_ids = None
        self.N = 20                       # scripts/cross_validate.py:48
        self.method = 'combo'
        self.equalize_reference = False
        self.scoring_function = phamer.score_points   # scripts/cross_validate.py:275
        self.seed = None

    def fold_assignment(self, n):
        asmt = np.arange(n) % self.N
        (np.random if self.seed is None else self._rng).shuffle(asmt)
        return asmt

    def cross_validate(self):
        """Scores every reference row with a model trained on the other N-1 folds
        (scripts/cross_validate.py:57-101).  Returns (positive_scores, negative_scores)."""
        self.num_positive = self.positive_data.shape[0]
        self.num_negative = self.negative_data.shape[0]
        if self.equalize_reference and self.num_positive != self.num_negative:
            num_ref = min(self.num_positive, self.num_negative)
            self.positive_data = self.positive_data[:num_ref]
            self.negative_data = self.negative_data[:num_ref]
            if self.positive_ids is not None:
                self.positive_ids = self.positive_ids[:num_ref]
            if self.negative_ids is not None:
                self.negative_ids = self.negative_ids[:num_ref]
            self.num_positive = self.num_negative = num_ref
        self._rng = np.random.RandomState(self.seed) if self.seed is not None else None
        positive_asmt = self.fold_assignment(self.num_positive)
        negative_asmt = self.fold_assignment(self.num_negative)
        self.positive_assignment, self.negative_assignment = positive_asmt, negative_asmt
        self.positive_scores = np.zeros(self.num_positive)
        self.negative_scores = np.zeros(self.num_negative)
        for n in range(self.N):
            logger.info('Iteration %d/%d' % (1 + n, self.N))
            where_positive = (positive_asmt == n)
            where_negative = (negative_asmt == n)
            n_pos = int(np.sum(where_positive))
            scoring_data = np.vstack((self.positive_data[where_positive], self.negative_data[where_negative]))
            scores = self.scoring_function(scoring_data, self.positive_data[~where_positive],
                                           self.negative_data[~where_negative], method=self.method)
            self.positive_scores[where_positive] = scores[:n_pos]
            self.negative_scores[where_negative] = scores[n_pos:]
        return self.positive_scores, self.negative_scores
