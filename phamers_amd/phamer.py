"""
phamer.py -- drop-in for the scoring entry points of PhaMers' scripts/phamer.py, device resident.

    phamer_scorer            attribute surface + load_data / screen_by_length / equalize_reference_data /
                             score_points / knn_ / kmeans_ / combo_score_points      scripts/phamer.py:42-313
    score_points(scoring_data, positive_training_data, negative_training_data, method=None)
                                                                                    scripts/phamer.py:451-468
    main()                   `python -m phamers_amd.phamer -in <dir> -data <dir> [-e]`  scripts/phamer.py:512-598

Data path.  A FASTA input is parsed once by the native reader, its bases are uploaded once, and from there the
contigs live on the GPU as a `_lib.Batch` (uint32 counts + row sums in HBM): the length screen is a device row
gather, scoring runs on the resident counts, and the host sees the scores -- plus the count matrix once, for the
features cache the reference writes next to the FASTA (scripts/phamer.py:132-134).  ``data_points`` keeps its
reference meaning (the normalised float64 matrix) but is materialised from the device only if somebody reads it;
assigning it (as phamer.score_points and the cross-validation do) switches the object to host rows, which are
scored through the float64-row entry point.

The k-means fit that yields the centroids is scikit-learn's, as in the reference (learning.kmeans).  Methods
outside {knn, kmeans, combo} raise NotImplementedError: dbscan / svm / density / silhouette are outside the
accelerated path.
"""
import argparse
import logging
import os

import numpy as np

from . import _lib
from . import fileIO
from . import learning

logging.basicConfig(format='[%(asctime)s][%(levelname)s][%(funcName)s] - %(message)s')
logger = logging.getLogger(__name__)
logger.setLevel(logging.WARNING)

_GPU_METHODS = ('knn', 'kmeans', 'combo')
_OTHER_METHODS = ('dbscan', 'svm', 'density', 'silhouette')


def _lone(directory, suffixes, avoid_stem_suffix=None):
    """The file of `directory` the reference would pick (scripts/phamer.py:417-436): the only one with one of
    `suffixes`; among several, the first whose stem does not end in `avoid_stem_suffix`, else the first."""
    if not directory or not os.path.isdir(directory):
        return None
    names = [f for f in os.listdir(directory) if f.endswith(suffixes)]
    if not names:
        return None
    if len(names) > 1 and avoid_stem_suffix is None:
        return None
    for name in names:
        if len(names) == 1 or not os.path.splitext(name)[0].endswith(avoid_stem_suffix):
            return os.path.join(directory, name)
    return os.path.join(directory, names[0])


class phamer_scorer(object):

    def __init__(self):
        # where things are (scripts/phamer.py:46-57)
        self.input_directory = self.fasta_file = self.features_file = None
        self.data_directory = self.positive_features_file = self.negative_features_file = None
        self.output_directory = None
        # what the hot path reads (scripts/phamer.py:59-79)
        self.data_ids = None
        self.positive_ids = self.positive_data = None
        self.negative_ids = self.negative_data = None
        self.length_requirement = 5000
        self.scoring_method = 'combo'
        self.all_scoring_methods = ['dbscan', 'kmeans', 'knn', 'svm', 'density', 'silhouette', 'combo']
        self.method_function_map = dict(
            [(m, getattr(self, m + '_score_points')) for m in _GPU_METHODS] + [(m, self._outside_path) for m in _OTHER_METHODS])
        self.kmer_length = 4
        self.k_clusters = 86
        self.k_neighbors = 3
        # centroids of the last kmeans / combo call (inspection, tests)
        self.positive_centroids = self.negative_centroids = None
        self.scores = None
        # query side: device-resident batch and / or host rows
        self._batch = None
        self._rows = None
        # host work that runs beside the path (main() only; the methods below are synchronous for API callers):
        # the features-cache write and the per-run k-means fit of the reference matrices
        self._defer_io = False
        self._pending_io = []
        self._centroid_future = None
        self._centroid_start = None

    # ---- data_points: the reference's attribute, lazily backed by the device batch --------------------
    @property
    def data_points(self):
        if self._rows is None and self._batch is not None:
            self._rows = self._batch.normalized()
        return self._rows

    @data_points.setter
    def data_points(self, rows):
        self._rows = rows
        self._drop_batch()

    def _drop_batch(self):
        if self._batch is not None:
            self._batch.close()
            self._batch = None

    # ---- files either side of the path ------------------------------------------------------------------
    def find_data_files(self):
        """Default reference files under the data directory (scripts/phamer.py:406-415)."""
        ref = os.path.join(self.data_directory, "reference_features")
        self.positive_features_file = os.path.join(ref, "positive_features.csv")
        self.negative_features_file = os.path.join(ref, "negative_features.csv")

    def find_input_files(self):
        """scripts/phamer.py:417-436: the FASTA (not the '*genes' one) and the lone features CSV of the input directory."""
        self.fasta_file = _lone(self.input_directory, ('.fasta', '.fa'), avoid_stem_suffix="genes") or self.fasta_file
        self.features_file = _lone(self.input_directory, ('.csv',)) or self.features_file

    def load_data(self, length_requirement=True):
        """Reference matrices (normalised) and the query contigs (scripts/phamer.py:103-142).  Queries come from the
        cached features CSV when the input directory has one; otherwise the FASTA is counted on the GPU, the counts are
        written through to ``<fasta>_features.csv`` and stay on the device for scoring.  A truthy
        ``length_requirement`` applies the length screen -- always with self.length_requirement (5000), as in the
        reference, whose CLI value is never forwarded."""
        self._load_reference()
        self._load_queries(length_requirement)

    def _load_reference(self):
        self.positive_ids, self.positive_data = fileIO.read_feature_file(self.positive_features_file, normalize=True)
        self.negative_ids, self.negative_data = fileIO.read_feature_file(self.negative_features_file, normalize=True)

    def _load_queries(self, length_requirement=True):
        self.find_input_files()
        lengths = None
        if self.features_file and os.path.exists(self.features_file):
            logger.info("Reading features from: %s..." % os.path.basename(self.features_file))
            from . import kmer
            self.data_ids, counts = fileIO.read_feature_file(self.features_file)
            _lap("features file read")
            # the cached counts go up once and the run scores from the same resident integers as a run that counted the
            # FASTA file (data_points, the reference's normalised float rows, is formed on demand); a file that is not an
            # integer k-mer count matrix keeps the reference's float rows
            self._drop_batch()
            self._rows = None
            batch = None
            if np.issubdtype(np.asarray(counts).dtype, np.integer) and np.asarray(counts).ndim == 2 and len(counts):
                batch = _lib.Batch.from_counts(_lib.get_context(), counts)
            if batch is not None:
                self._batch = batch
                self.kmer_length = int(round(np.log(batch.D) / np.log(4)))
            else:
                self.data_points = kmer.normalize_counts(counts)
            del counts
            _lap("features on the device")
            self._start_pending_centroids()      # (beside the scan of the FASTA file for the length screen)
        elif self.fasta_file and os.path.exists(self.fasta_file):
            logger.info("Calculating features of: %s" % os.path.basename(self.fasta_file))
            lengths = self._count_fasta_on_device()
        else:
            raise SystemExit("No input fasta file or features file. Exiting...")
        if length_requirement:
            self.screen_by_length(_lengths=lengths)

    def _count_fasta_on_device(self):
        ctx = _lib.get_context()
        self._drop_batch()
        self._rows = None
        # the file is parsed straight into the upload's staging buffers (phk_batch_from_fasta_file): no host copy of the
        # sequences, nothing to free afterwards; the k-means thread starts beside it
        self._start_pending_centroids()
        fasta, self._batch = _lib.Fasta.count_file(ctx, self.fasta_file, self.kmer_length)
        _lap("FASTA parsed, uploaded, counted")
        try:
            self.data_ids = fasta.phamers_ids()
            lengths = fasta.lengths()
            counts = self._batch.counts_u32()
            _lap("counts on the host")
        finally:
            fasta.close()
        self.features_file = "{base}_features.csv".format(base=os.path.splitext(self.fasta_file)[0])
        ids, path = self.data_ids, self.features_file
        if self._defer_io:
            self._write_cache_async(counts, ids, path)
        else:
            fileIO.save_counts(counts, ids, path)
        return lengths

    def _write_cache_async(self, counts, ids, path):
        """main(): the 0.6 s (1M contigs) of formatting and writing the features cache run beside the k-means fit and
        the scoring; the file appears under its name only when complete, and a failed write is raised by finish_io()."""
        import threading

        def write():
            tmp = path + ".part"
            try:
                _lap("features cache: write begins")
                fileIO.save_counts(counts, ids, tmp)
                _lap("features cache: written")
                os.replace(tmp, path)
            except BaseException as e:   # noqa: BLE001 -- handed to finish_io(), which raises it where the
                th.error = e             # reference's sequential save_counts would have
                try:
                    os.unlink(tmp)
                except OSError:
                    pass
        th = threading.Thread(target=write, name="phamers-features-cache")
        th.error = None
        th.start()
        self._pending_io.append(th)

    def finish_io(self):
        """Waits for file writes main() left running beside the scoring."""
        pending, self._pending_io = self._pending_io, []
        first = None
        for th in pending:
            th.join()
            first = first or getattr(th, "error", None)
        if first is not None:   # a failed write (disk full, read-only input directory) is an error of the run
            raise first

    def screen_by_length(self, length_requirement=None, _lengths=None):
        """Keep contigs of at least ``length_requirement`` bases (scripts/phamer.py:144-157): ids filtered on the host,
        rows by a device gather when the contigs are resident."""
        if length_requirement:
            self.length_requirement = length_requirement
        if _lengths is None:
            if not self.fasta_file or not os.path.exists(self.fasta_file):
                return
            from . import kmer
            fasta_ids, _lengths = kmer.fasta_lengths(self.fasta_file)
            _lap("FASTA indexed (ids, lengths)")
        else:
            fasta_ids = self.data_ids
        keep, long_ids = _length_screen(self.data_ids, fasta_ids, _lengths, self.length_requirement)
        _lap("length screen: rows matched")
        if keep is None:
            self.data_ids = np.asarray(self.data_ids)
            return      # every row is a long contig's own: nothing to drop (and 10^6 ids not sorted and copied, 0.4 s)
        if self._batch is not None:
            old, self._batch = self._batch, self._batch.select(np.flatnonzero(keep))
            old.close()
            self._rows = None if self._rows is None else self._rows[keep]
        elif self._rows is not None:
            self._rows = self._rows[keep]
        self.data_ids = long_ids      # (already a copy: boolean indexing)

    def get_phamer_output_filename(self):
        return os.path.join(self.output_directory, "phamer_scores.csv")

    def make_summary_file(self, args=None):
        """phamer_scores.csv (scripts/phamer.py:316-323)."""
        self.phamer_output_filename = self.get_phamer_output_filename()
        fileIO.save_phamer_scores(self.data_ids, self.scores, self.phamer_output_filename, args=args)

    # ---- scoring ------------------------------------------------------------------------------------------
    def equalize_reference_data(self):
        """The FIRST min(n+, n-) rows of each class (scripts/phamer.py:159-175)."""
        m = min(self.positive_data.shape[0], self.negative_data.shape[0])
        if self.positive_data.shape[0] != self.negative_data.shape[0]:
            logger.debug("Equalizing reference data to: %d data points" % m)
            self.positive_data, self.negative_data = self.positive_data[:m], self.negative_data[:m]
            self.positive_ids = None if self.positive_ids is None else self.positive_ids[:m]
            self.negative_ids = None if self.negative_ids is None else self.negative_ids[:m]
            self.num_positive = self.num_negative = m

    def score_points(self):
        """Scores the query contigs against ``positive_data`` / ``negative_data`` with ``scoring_method``
        (scripts/phamer.py:177-195)."""
        self.num_points = self._batch.n if self._batch is not None else self.data_points.shape[0]
        self.num_positive, self.num_negative = self.positive_data.shape[0], self.negative_data.shape[0]
        logger.debug("Scoring %d points. Method: %s..." % (self.num_points, self.scoring_method))
        self.scores = np.array(self.method_function_map[self.scoring_method]())
        return self.scores

    def _outside_path(self):
        raise NotImplementedError("scoring method %r is outside the accelerated path; knn / kmeans / combo are "
                                  "available" % (self.scoring_method,))

    def _centroids_of(self, pos, neg, k_clusters, deliver=None):
        """k-means labels -> the reference's per-label means, for both classes.  deliver: called from a helper thread,
        which gets a device context of its own (a context is not shared between threads) and hands the result (or the
        exception) to this Future BEFORE it closes that context -- freeing the context's buffers waits for the device,
        0.2 s beside the main thread's upload."""
        _lap("k-means: begin")
        ctx = None
        try:
            try:
                # (inside the guarded block: a context that cannot be created -- no memory for its streams, a HIP error --
                # must reach the Future as an exception, or the main thread waits for this result for ever)
                ctx = _lib.Context(_lib.default_device()) if deliver is not None else None
                out = []
                for d in (pos, neg):
                    out.append(learning.get_centroids(d, learning.kmeans(d, k_clusters, _ctx=ctx)))
                    _lap("k-means: one class done")
                out = tuple(out)
            except BaseException as e:   # noqa: BLE001 -- raised again by Future.result() in the main thread
                if deliver is None:
                    raise
                deliver.set_exception(e)
                return None
            if deliver is not None:
                deliver.set_result(out)
            return out
        finally:
            if ctx is not None:
                ctx.close()

    def prefetch_centroids(self, after_fasta_read=False):
        """Starts the k-means fit of the reference matrices as they are NOW on a helper thread, so that it runs beside the
        upload, the counting and the cache write (the seeding's matrix products release the GIL; the Lloyd sweeps run on
        the device through a context of the thread's own); _fit_centroids takes the result if the matrices have not been
        replaced since, and fits afresh otherwise.  after_fasta_read: the thread starts when the FASTA file has been
        parsed -- beside the native reader's threads it took the parse from 0.32 to 0.48 s (1M contigs) and gained
        nothing, since the fit (0.11 s) fits beside what follows."""
        if os.environ.get("PHAMERS_KMEANS", "device") == "gpu":
            return
        pos, neg, k = self.positive_data, self.negative_data, self.k_clusters

        def start():
            import threading
            from concurrent.futures import Future
            fut = Future()
            self._centroid_future = (pos, neg, k, fut)
            threading.Thread(target=self._centroids_of, args=(pos, neg, k, fut), name="phamers-kmeans").start()
        if after_fasta_read and os.environ.get("PHAMERS_KMEANS_START", "after_read") != "early":
            self._centroid_start = start      # _count_fasta_on_device() calls it once the file is parsed
        else:
            start()

    def _start_pending_centroids(self):
        start, self._centroid_start = self._centroid_start, None
        if start is not None:
            start()

    def _fit_centroids(self):
        """k-means with k_clusters on each class, then the cluster means (scripts/phamer.py:245-248)."""
        self._centroid_start = None       # (never started: no FASTA was read -- fit here, in the main thread's context)
        fut, self._centroid_future = self._centroid_future, None
        if fut is not None and fut[0] is self.positive_data and fut[1] is self.negative_data and fut[2] == self.k_clusters:
            self.positive_centroids, self.negative_centroids = fut[3].result()
            return
        self.positive_centroids, self.negative_centroids = self._centroids_of(self.positive_data, self.negative_data,
                                                                              self.k_clusters)

    def _gpu_score(self, method):
        with_centroids = method != 'knn'
        if with_centroids:
            self._fit_centroids()
        model = _lib.Model(_lib.get_context(), self.positive_data, self.negative_data,
                           self.positive_centroids if with_centroids else None,
                           self.negative_centroids if with_centroids else None, k_neighbors=self.k_neighbors)
        try:
            if self._batch is not None:
                return self._batch.score(model, method)          # resident counts; NaN rows raise ValueError
            q = np.asarray(self._rows, dtype=np.float64)
            if np.isnan(q).any():
                raise ValueError("Input contains NaN.")           # what scikit-learn raises for the reference
            return model.score(q, method)
        finally:
            model.close()

    def knn_score_points(self):
        """scripts/phamer.py:268-273."""
        return self._gpu_score('knn')

    def kmeans_score_points(self):
        """scripts/phamer.py:240-256: tanh proximity metric to the nearest centroid of each class."""
        return self._gpu_score('kmeans')

    def combo_score_points(self):
        """scripts/phamer.py:303-313: knn score + kmeans score (one fused GPU pass)."""
        return self._gpu_score('combo')


def _length_screen(data_ids, fasta_ids, lengths, length_requirement):
    """The rows screen_by_length keeps (scripts/phamer.py:144-157): (keep mask over data_ids, ids of the long contigs), or
    (None, None) when every row is a long contig's own and nothing is dropped."""
    is_long = np.asarray(lengths) >= length_requirement
    # the rows are the file's records in the file's order: always so when they were just counted from it, and the
    # usual case when they came from the features cache written beside it (one vectorised comparison to know)
    own_rows = fasta_ids is data_ids or (
        len(fasta_ids) == len(data_ids) and bool(np.array_equal(np.asarray(fasta_ids), np.asarray(data_ids))))
    if own_rows and is_long.all():
        return None, None
    long_ids = np.asarray(fasta_ids)[is_long]
    if own_rows:
        # the reference keeps a row when its id is that of SOME contig long enough (np.in1d): true for every long contig's
        # own row, so only the short contigs' ids have to be looked up (none on a batch of long contigs; a sort of 10^6
        # strings otherwise, 0.2 s)
        keep = is_long.copy()
        short = ~is_long
        if short.any() and is_long.any():
            keep[short] = np.isin(np.asarray(data_ids)[short], long_ids)
    else:
        keep = np.isin(data_ids, long_ids)
    return keep, long_ids


def score_points(scoring_data, positive_training_data, negative_training_data, method=None):
    """Functional form of phamer_scorer.score_points (scripts/phamer.py:451-468), the scoring function of the
    reference's cross-validation (scripts/cross_validate.py:95)."""
    scorer = phamer_scorer()
    scorer.scoring_method = method or scorer.scoring_method
    scorer.data_points = scoring_data
    scorer.positive_data, scorer.negative_data = positive_training_data, negative_training_data
    return scorer.score_points()


def score_contigs(sequences, positive_training_data, negative_training_data, kmer_length=4, method='combo'):
    """Count -> normalise -> score a list of contig strings, device resident (what load_data + score_points do for a
    FASTA input, scripts/phamer.py:131,139,579)."""
    scorer = phamer_scorer()
    scorer.scoring_method = method
    scorer.kmer_length = kmer_length
    scorer._batch = _lib.Batch.from_sequences(_lib.get_context(), list(sequences), kmer_length)
    scorer.positive_data, scorer.negative_data = positive_training_data, negative_training_data
    try:
        return scorer.score_points()
    finally:
        scorer._drop_batch()


_clock0 = [None]


def _lap(what):
    """--debug: seconds since main() began, from whichever thread."""
    import time
    if _clock0[0] is not None:
        logger.debug("%-34s %.3f s" % (what, time.perf_counter() - _clock0[0]))


def main(argv=None):
    """The reference's command line for this path (scripts/phamer.py:512-598), cut to what the path uses:
        python -m phamers_amd.phamer -in <input_dir> -data <data_dir> [--equalize_reference]
    Scores go to <input_dir>/phamer_output/phamer_scores.csv (or -out).  As in the reference the method is always
    'combo' (its --method is parsed and never applied)."""
    ap = _parser()
    args = ap.parse_args(argv)
    if args.do_tsne or args.plot_tsne:
        raise NotImplementedError("t-SNE / plots (-do_tsne, -plot) are outside the accelerated path (SURVEY.md section 8)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or os.environ.get("PHAMERS_FORCE_RANK_PATH") == "1":   # one rank of a sharded run (started below, or by
                                                                        # torch.distributed.run; forced: a 1-rank group, tests)
        return _run_rank(ap, args)
    if args.gpus and args.gpus > 1:    # the launcher: it never touches the GPU itself -- the ranks are its children
        import sys
        from . import dist as pdist
        cmd = [sys.executable, "-m", "phamers_amd.phamer"] + list(sys.argv[1:] if argv is None else argv)
        rc = pdist.launch_ranks(args.gpus, cmd, require_gpus=os.environ.get("PHAMERS_DIST_BACKEND", "nccl") == "nccl")
        if rc:
            raise SystemExit(rc)
        return None
    return _run(ap, args)


def _parser():
    ap = argparse.ArgumentParser(description='This script scores contigs based on feature similarity',
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for flags, kw in (
            (('-in', '--input_directory'), dict(help='Directory containing input files')),
            (('-fasta', '--fasta_file'), dict(help='Fasta compilation file of unknown sequences')),
            (('-features', '--features_file'), dict(help='Input feature file')),
            (('-data', '--data_directory'), dict(help='Directory containing reference_features/')),
            (('-pf', '--positive_features'), dict(help='Positive reference features CSV')),
            (('-nf', '--negative_features'), dict(help='Negative reference features CSV')),
            (('-out', '--output_directory'), dict(help='Output directory path')),
            (('-k', '--kmer_length'), dict(type=int, default=4, help='k-mer length')),
            (('-l', '--length_requirement'), dict(type=int, default=5000, help='Input sequence length requirement')),
            (('-equal', '-e', '--equalize_reference'), dict(action='store_true', help='Same number of reference points')),
            (('--gpus',), dict(type=int, default=1, help='GPUs of this node to shard the contigs over (one process per GPU)')),
            (('-v', '--verbose'), dict(action='store_true')),
            (('--debug',), dict(action='store_true')),
            # the rest of the reference's command line (scripts/phamer.py:515-553), so that its launch scripts
            # (run.sh, submit/*.sh) parse unchanged.  The reference parses --method / --eps / --minPts and never applies
            # them (the method is always 'combo'); t-SNE, plots and counting the reference sets from FASTA are outside
            # the accelerated path and are refused when asked for.
            (('-tsne', '--tsne_file'), dict(help='(outside the accelerated path)')),
            (('-p', '--positive_fasta'), dict(help='(outside the accelerated path: give -pf)')),
            (('-n', '--negative_fasta'), dict(help='(outside the accelerated path: give -nf)')),
            (('-id', '--file_identifier'), dict(default='.fna', help='(used with -n only)')),
            (('-do_tsne', '--do_tsne'), dict(action='store_true', help='(outside the accelerated path)')),
            (('-pxty', '--perplexity'), dict(type=float, default=30, help='(t-SNE only)')),
            (('-plot', '--plot_tsne'), dict(action='store_true', help='(outside the accelerated path)')),
            (('-m', '--method'), dict(default='combo', help='parsed and, as in the reference, not applied')),
            (('-eps', '--eps'), dict(type=float, default=2.1, help='parsed and, as in the reference, not applied')),
            (('-mp', '--minPts'), dict(type=int, default=2, help='parsed and, as in the reference, not applied'))):
        ap.add_argument(*flags, **kw)
    return ap


def _run(ap, args):
    logger.setLevel(logging.DEBUG if args.debug else logging.INFO if args.verbose else logging.WARNING)

    scorer = phamer_scorer()
    scorer.kmer_length = args.kmer_length
    scorer.input_directory, scorer.fasta_file, scorer.features_file = args.input_directory, args.fasta_file, args.features_file
    if args.data_directory:
        scorer.data_directory = args.data_directory
        scorer.find_data_files()
    scorer.positive_features_file = args.positive_features or scorer.positive_features_file
    scorer.negative_features_file = args.negative_features or scorer.negative_features_file
    if not (scorer.positive_features_file and scorer.negative_features_file):
        ap.error("give -data <dir with reference_features/> or -pf and -nf")
    scorer.output_directory = args.output_directory or os.path.join(
        args.input_directory or os.path.dirname(args.fasta_file or args.features_file or '.'), "phamer_output")
    # load_data(), with the host work that does not depend on the contigs started early: the reference matrices are
    # read (and equalised) first, their k-means fit runs beside the FASTA ingest and the counting, and the features
    # cache is written beside the scoring.  Same files, same numbers as the sequential order of the reference.
    import time
    _clock0[0] = time.perf_counter()
    lap = _lap
    scorer._defer_io = True
    try:
        scorer._load_reference()
        if args.equalize_reference:
            scorer.equalize_reference_data()
        lap("reference matrices")
        scorer.prefetch_centroids(after_fasta_read=True)
        scorer._load_queries(args.length_requirement)
        lap("contigs counted")
        os.makedirs(scorer.output_directory, exist_ok=True)
        scorer.score_points()
        lap("scored")
        scorer.make_summary_file(args=args)
        lap("scores written")
    finally:
        scorer.finish_io()
        lap("features cache complete")
    return scorer


def _rank_count_and_score(fasta_file, part, kmer_length, method, positive, negative, cpos, cneg, k_neighbors, keep_of, gpu):
    """One rank's share of the FASTA file on its GPU: the records that begin in byte range ``part`` = (rank, world) are parsed
    straight into the upload's staging buffers (phk_batch_from_fasta_part), counted, screened (``keep_of(ids, lengths)`` ->
    boolean mask over this rank's records, decided with every rank's ids) and scored.  Returns (ids, counts uint32 of every
    record, keep mask, scores of the kept).  (A module-level function so that the CPU tests can put the oracle in its
    place: the product has no CPU path.)"""
    ctx = _lib.get_context(gpu)
    threads = max(1, (os.cpu_count() or 1) // max(1, part[1]))
    fasta, batch = _lib.Fasta.count_file(ctx, fasta_file, kmer_length, threads=threads, part=part)
    try:
        ids, lengths = fasta.phamers_ids(), fasta.lengths()
    finally:
        fasta.close()
    try:
        counts = batch.counts_u32() if batch.n else np.zeros((0, 4 ** int(kmer_length)), dtype=np.uint32)
        keep = keep_of(ids, lengths)
        scores = np.zeros(0)
        if keep.any():
            sub = batch if keep.all() else batch.select(np.flatnonzero(keep))
            model = _lib.Model(ctx, positive, negative, cpos if method != 'knn' else None, cneg if method != 'knn' else None,
                               k_neighbors=k_neighbors)
            try:
                scores = sub.score(model, method)
            finally:
                model.close()
                if sub is not batch:
                    sub.close()
        return ids, counts, keep, scores
    finally:
        batch.close()


def _run_rank(ap, args):
    """One rank of `python -m phamers_amd.phamer ... --gpus N` (scripts/phamer.py:512-598 on a shard): the
    reference matrices and centroids are replicated, every rank loads, counts and scores the records of ITS byte range of the
    FASTA file, and the ranks exchange ids, lengths (for the length screen, which looks ids up across the whole file), scores
    and -- for the features cache rank 0 writes -- counts.  Rank 0 writes the same phamer_scores.csv and <fasta>_features.csv
    as the one-GPU run, byte for byte."""
    import torch
    import torch.distributed as tdist
    from . import dist as pdist
    logger.setLevel(logging.DEBUG if args.debug else logging.INFO if args.verbose else logging.WARNING)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("PHAMERS_DIST_BACKEND", "nccl")
    gpu = None
    if backend == "nccl":
        gpu = pdist.rank_device()
        tdist.init_process_group("nccl", device_id=torch.device("cuda", gpu))
    else:
        tdist.init_process_group(backend)
    dev = torch.device("cuda", gpu) if backend == "nccl" else torch.device("cpu")
    try:
        scorer = phamer_scorer()
        scorer.kmer_length = args.kmer_length
        scorer.input_directory, scorer.fasta_file, scorer.features_file = args.input_directory, args.fasta_file, args.features_file
        if args.data_directory:
            scorer.data_directory = args.data_directory
            scorer.find_data_files()
        scorer.positive_features_file = args.positive_features or scorer.positive_features_file
        scorer.negative_features_file = args.negative_features or scorer.negative_features_file
        if not (scorer.positive_features_file and scorer.negative_features_file):
            ap.error("give -data <dir with reference_features/> or -pf and -nf")
        scorer.output_directory = args.output_directory or os.path.join(
            args.input_directory or os.path.dirname(args.fasta_file or args.features_file or '.'), "phamer_output")
        scorer.find_input_files()
        if not (scorer.fasta_file and os.path.exists(scorer.fasta_file)):
            raise SystemExit("--gpus needs the input FASTA file (a features cache alone is scored by one GPU). Exiting...")
        scorer._load_reference()
        if args.equalize_reference:
            scorer.equalize_reference_data()
        # centroids: fitted once, by rank 0, and handed to every rank (the fit is deterministic; this makes "replicated" literal)
        shape = (scorer.k_clusters, scorer.positive_data.shape[1])
        cen = torch.zeros((2,) + shape, dtype=torch.float64, device=dev)
        if rank == 0:
            scorer._fit_centroids()
            cen[0] = torch.from_numpy(np.ascontiguousarray(scorer.positive_centroids)).to(dev)
            cen[1] = torch.from_numpy(np.ascontiguousarray(scorer.negative_centroids)).to(dev)
        tdist.broadcast(cen, src=0)
        cpos, cneg = cen[0].cpu().numpy(), cen[1].cpu().numpy()

        screen = int(scorer.length_requirement) if args.length_requirement else 0   # (the CLI value is a switch, as in the reference)

        def keep_of(ids, lengths):
            # the screen keeps a row whose id is that of SOME contig long enough, anywhere in the file: decided on every
            # rank from all ranks' ids and lengths, exactly as the one-GPU run decides it
            everything = [None] * world
            tdist.all_gather_object(everything, (list(ids), np.asarray(lengths, dtype=np.int64)))
            keep_of.all_ids = np.array([x for part in everything for x in part[0]], dtype=object if any(
                x is None for part in everything for x in part[0]) else None)
            if keep_of.all_ids.dtype != object:
                keep_of.all_ids = keep_of.all_ids.astype(str)
            all_len = np.concatenate([part[1] for part in everything]) if everything else np.zeros(0, dtype=np.int64)
            first = sum(len(part[0]) for part in everything[:rank])
            keep_of.kept_ids = keep_of.all_ids
            keep_of.all_keep = np.ones(len(all_len), dtype=bool)
            if screen:
                k_all, long_ids = _length_screen(keep_of.all_ids, keep_of.all_ids, all_len, screen)
                if k_all is not None:    # (rows and ids exactly as screen_by_length leaves them in the one-GPU run)
                    keep_of.all_keep, keep_of.kept_ids = k_all, long_ids
            return keep_of.all_keep[first:first + len(ids)]

        ids, counts, keep, scores = _rank_count_and_score(scorer.fasta_file, (rank, world), scorer.kmer_length, scorer.scoring_method,
                                                          scorer.positive_data, scorer.negative_data, cpos, cneg, scorer.k_neighbors,
                                                          keep_of, gpu)
        all_scores = pdist.gather_variable(torch.from_numpy(np.ascontiguousarray(scores, dtype=np.float64)).to(dev)).cpu().numpy()
        all_counts = pdist.gather_rows_to_root(torch.from_numpy(np.ascontiguousarray(counts).view(np.int32)).to(dev))
        if rank == 0:
            os.makedirs(scorer.output_directory, exist_ok=True)
            scorer.data_ids = keep_of.kept_ids
            scorer.scores = all_scores
            scorer.make_summary_file(args=args)
            cache = "{base}_features.csv".format(base=os.path.splitext(scorer.fasta_file)[0])
            fileIO.save_counts(all_counts.cpu().numpy().view(np.uint32), keep_of.all_ids, cache)
        tdist.barrier()
        return scorer if rank == 0 else None
    finally:
        tdist.destroy_process_group()


if __name__ == '__main__':
    main()
