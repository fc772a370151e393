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
import argparse
import logging
import os

import numpy as np

from . import _lib
from . import fileIO
from . import kmer
from . import learning

logging.basicConfig(format='[%(asctime)s][%(levelname)s][%(funcName)s] - %(message)s')
logger = logging.getLogger(__name__)
logger.setLevel(logging.WARNING)


class phamer_scorer(object):

    def __init__(self):
        # file locations (scripts/phamer.py:46-57)
        self.input_directory = None
        self.features_file = None
        self.fasta_file = None
        self.data_directory = None
        self.positive_features_file = None
        self.negative_features_file = None
        self.output_directory = None

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

    # ---- files either side of the path (scripts/phamer.py:103-157, 316-323, 406-440) ------------
    def find_data_files(self):
        """Reference feature files at their default place under the data directory
        (scripts/phamer.py:406-415)."""
        self.positive_features_file = os.path.join(self.data_directory, "reference_features", "positive_features.csv")
        self.negative_features_file = os.path.join(self.data_directory, "reference_features", "negative_features.csv")

    def find_input_files(self):
        """The lone *.fasta|*.fa (not '*genes') and the lone *.csv of the input directory
        (scripts/phamer.py:417-436)."""
        if self.input_directory and os.path.isdir(self.input_directory):
            fasta_files = [f for f in os.listdir(self.input_directory) if f.endswith('.fasta') or f.endswith('.fa')]
            if len(fasta_files) == 1:
                self.fasta_file = os.path.join(self.input_directory, fasta_files[0])
            elif fasta_files:
                for candidate in fasta_files:
                    if not os.path.splitext(candidate)[0].endswith("genes"):
                        self.fasta_file = os.path.join(self.input_directory, candidate)
                        break
                if self.fasta_file is None:
                    self.fasta_file = os.path.join(self.input_directory, fasta_files[0])
            features_files = [f for f in os.listdir(self.input_directory) if f.endswith('.csv')]
            if len(features_files) == 1:
                self.features_file = os.path.join(self.input_directory, features_files[0])

    def load_data(self, length_requirement=True):
        """Reference matrices (normalised) + query features: a cached features CSV if the input
        directory has one, else count the FASTA on the GPU and write the cache next to it
        (scripts/phamer.py:103-142).  ``length_requirement`` truthy applies the length screen, which --
        as in the reference -- always uses self.length_requirement (5000), not the CLI value."""
        self.positive_ids, self.positive_data = fileIO.read_feature_file(self.positive_features_file, normalize=True)
        self.negative_ids, self.negative_data = fileIO.read_feature_file(self.negative_features_file, normalize=True)
        self.find_input_files()
        if self.features_file is not None and os.path.exists(self.features_file):
            logger.info("Reading features from: %s..." % os.path.basename(self.features_file))
            self.data_ids, self.data_points = fileIO.read_feature_file(self.features_file)
        elif self.fasta_file is not None and os.path.exists(self.fasta_file):
            logger.info("Calculating features of: %s" % os.path.basename(self.fasta_file))
            self.data_ids, self.data_points = kmer.count_file(self.fasta_file, self.kmer_length, normalize=False)
            self.features_file = "{base}_features.csv".format(base=os.path.splitext(self.fasta_file)[0])
            fileIO.save_counts(self.data_points, self.data_ids, self.features_file)
        else:
            raise SystemExit("No input fasta file or features file. Exiting...")
        self.data_points = kmer.normalize_counts(self.data_points)
        if length_requirement:
            self.screen_by_length()

    def screen_by_length(self, length_requirement=None):
        """Keep contigs with at least ``length_requirement`` bases (scripts/phamer.py:144-157)."""
        if length_requirement:
            self.length_requirement = length_requirement
        if self.fasta_file is None or not os.path.exists(self.fasta_file):
            return
        unknown_ids, lengths = kmer.fasta_lengths(self.fasta_file)
        long_ids = [unknown_ids[i] for i in range(len(unknown_ids)) if lengths[i] >= self.length_requirement]
        self.data_points = self.data_points[np.isin(self.data_ids, long_ids)]
        self.data_ids = np.array(long_ids)

    def get_phamer_output_filename(self):
        return os.path.join(self.output_directory, "phamer_scores.csv")

    def make_summary_file(self, args=None):
        """Write phamer_scores.csv (scripts/phamer.py:316-323)."""
        self.phamer_output_filename = self.get_phamer_output_filename()
        fileIO.save_phamer_scores(self.data_ids, self.scores, self.phamer_output_filename, args=args)

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


def main(argv=None):
    """Command-line driver with the reference's flags for this path (scripts/phamer.py:512-598):
        python -m phamers_amd.phamer -in <input_dir> -data <data_dir> [--equalize_reference]
    Counting, normalising and scoring run on the GPU; the scores go to
    <input_dir>/phamer_output/phamer_scores.csv (or -out).  As in the reference the scoring method
    is always 'combo' (--method is parsed but never applied there)."""
    parser = argparse.ArgumentParser(description='This script scores contigs based on feature similarity',
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument('-in', '--input_directory', help='Directory containing input files')
    parser.add_argument('-fasta', '--fasta_file', help='Fasta compilation file of unknown sequences')
    parser.add_argument('-features', '--features_file', help='Input feature file')
    parser.add_argument('-data', '--data_directory', help='Directory containing reference_features/')
    parser.add_argument('-pf', '--positive_features', help='Positive reference features CSV')
    parser.add_argument('-nf', '--negative_features', help='Negative reference features CSV')
    parser.add_argument('-out', '--output_directory', help='Output directory path')
    parser.add_argument('-k', '--kmer_length', type=int, default=4, help='k-mer length')
    parser.add_argument('-l', '--length_requirement', type=int, default=5000, help='Input sequence length requirement')
    parser.add_argument('-e', '--equalize_reference', action='store_true', help='Same number of reference points')
    parser.add_argument('-v', '--verbose', action='store_true')
    parser.add_argument('--debug', action='store_true')
    args = parser.parse_args(argv)
    logger.setLevel(logging.DEBUG if args.debug else logging.INFO if args.verbose else logging.WARNING)

    scorer = phamer_scorer()
    scorer.kmer_length = args.kmer_length
    scorer.input_directory = args.input_directory
    scorer.fasta_file = args.fasta_file
    scorer.features_file = args.features_file
    if args.data_directory:
        scorer.data_directory = args.data_directory
        scorer.find_data_files()
    if args.positive_features:
        scorer.positive_features_file = args.positive_features
    if args.negative_features:
        scorer.negative_features_file = args.negative_features
    if not (scorer.positive_features_file and scorer.negative_features_file):
        parser.error("give -data <dir with reference_features/> or -pf and -nf")
    if args.output_directory:
        scorer.output_directory = args.output_directory
    else:
        base = args.input_directory or os.path.dirname(args.fasta_file or args.features_file or '.')
        scorer.output_directory = os.path.join(base, "phamer_output")
    scorer.load_data(length_requirement=args.length_requirement)
    if args.equalize_reference:
        scorer.equalize_reference_data()
    if not os.path.isdir(scorer.output_directory):
        os.makedirs(scorer.output_directory)
    scorer.score_points()
    scorer.make_summary_file(args=args)
    return scorer


if __name__ == '__main__':
    main()
