"""
kmer.py -- drop-in for the hot-path functions of PhaMers' scripts/kmer.py, executed
by hand-written HIP kernels on an MI355X through libphamers_hip.so.

Same names, argument meaning, defaults and return shapes as the reference:

    count_string(sequence, kmer_length, symbols=DNA, normalize=False)   scripts/kmer.py:32
    count(data, kmer_length, symbols=DNA, normalize=False)              scripts/kmer.py:82
    count_file(input_file, kmer_length, symbols=DNA, normalize=False)   scripts/kmer.py:114
    count_directory(directory, kmer_length, identifier='fna', ...)       scripts/kmer.py:143
    normalize_counts(counts)                                            scripts/kmer.py:209
    kmers(k, symbols=DNA), sequence_to_integers, get_kmer_index         scripts/kmer.py:183-251
    main()   `python -m phamers_amd.kmer <fasta | dir> <out.csv> -k K`     scripts/kmer.py:283-334

Scope notes: only 4-symbol alphabets run on the GPU (the reference's
integer-replacement branch with DNA/RNA); other alphabets raise NotImplementedError --
there is no CPU fallback in this package.
"""
import logging
import os
import random

import numpy as np

from . import _lib

logging.basicConfig(format='[%(asctime)s][%(levelname)s][%(funcName)s] - %(message)s')
logger = logging.getLogger(__name__)
logger.setLevel(logging.WARNING)

DNA = 'ATGC'
RNA = 'AUGC'
protein = 'RHKDESTNQCUGPAVILMFYW'


def _check_symbols(symbols):
    if len(symbols) != 4 or len(set(symbols)) != 4:
        raise NotImplementedError(
            "phamers_amd counts k-mers over 4-symbol alphabets (DNA/RNA) on the GPU; symbols=%r is "
            "outside the accelerated path (use PhaMers' own kmer.py for it)" % (symbols,))
    try:
        return symbols.encode('latin-1')
    except UnicodeEncodeError:
        raise NotImplementedError("symbols must be single-byte characters")


def _count_batch(sequences, kmer_length, symbols):
    """n sequences -> (n, 4^k) int64 via phk_count_ascii (one upload, one launch chain)."""
    k = int(kmer_length)
    sym = _check_symbols(symbols)
    n = len(sequences)
    D = 4 ** k
    raw = [s.encode('latin-1', 'replace') for s in sequences]
    offsets = np.zeros(n + 1, dtype=np.uint64)
    if n:
        offsets[1:] = np.cumsum([len(r) for r in raw], dtype=np.uint64)
    bases = np.frombuffer(b''.join(raw) or b'\0', dtype=np.uint8)
    out = np.zeros((n, D), dtype=np.int64)
    ctx = _lib.get_context()
    _lib.check(ctx.lib.phk_count_ascii(ctx.handle, _lib.ptr(np.ascontiguousarray(bases)), _lib.ptr(offsets),
                                       n, k, sym, _lib.ptr(out)))
    return out


def count_string(sequence, kmer_length, symbols=DNA, normalize=False):
    """k-mer counting function (scripts/kmer.py:32-79): forward-strand, stride-1 windows;
    windows touching a character outside ``symbols`` (case-sensitive) are skipped; bin index
    has the first base as the most significant base-4 digit ('AAAT' -> 1).  Returns a 1-D
    int64 array of length 4^k, or float64 frequencies when ``normalize`` (all zeros stay
    zeros)."""
    counts = _count_batch([sequence], kmer_length, symbols)[0]
    if normalize:
        counts = counts.astype(float)
        if np.sum(counts) > 0:
            counts = normalize_counts(counts)
    return counts


def count(data, kmer_length, symbols=DNA, normalize=False):
    """K-mer counting dispatcher (scripts/kmer.py:82-111): str -> 1-D; list of one string ->
    1-D; list of n strings -> (n, 4^k); anything else -> None (logged)."""
    if isinstance(data, list):
        if len(data) == 1:
            return count(data[0], kmer_length, symbols=symbols, normalize=normalize)
        logger.info("Counting %d-mers in %d sequences..." % (kmer_length, len(data)))
        kmer_count = _count_batch(data, kmer_length, symbols)
        if normalize:
            # per-row count_string(normalize=True): zero rows stay zero (scripts/kmer.py:77)
            sums = kmer_count.sum(axis=1)
            kmer_count = normalize_counts(kmer_count) if len(data) else kmer_count.astype(float)
            kmer_count[sums == 0] = 0.0
    elif isinstance(data, str):
        kmer_count = count_string(data, kmer_length, symbols=symbols, normalize=normalize)
    else:
        logger.info("Data was not str or list: %s\n%s ..." % (type(data), data.__str__()[:25]))
        kmer_count = None
    return kmer_count


def count_file(input_file, kmer_length, symbols=DNA, normalize=False):
    """Counts k-mers of every record of a FASTA file (scripts/kmer.py:114-140).  Returns
    (ids, counts): ids parsed by the reference's header rules (scripts/id_parser.py:89-100),
    counts (n, 4^k).  An unreadable file gives (None, None).  The file is parsed once by the
    native multi-threaded reader (phk_fasta_read; plain or .gz) and counted on the GPU."""
    sym = _check_symbols(symbols)
    try:
        fasta = _lib.Fasta(input_file)
    except IOError:
        logger.warning("Could not read file: %s" % os.path.basename(input_file))
        return None, None
    try:
        ids = fasta.phamers_ids()
        counts = np.zeros((len(ids), pow(len(symbols), kmer_length)), dtype=(int, float)[normalize])
        if fasta.n_records:
            # bases up once, counts down once (device-resident batch)
            batch = _lib.Batch.from_fasta(_lib.get_context(), fasta, kmer_length, sym)
            try:
                got = batch.counts()
                if normalize:
                    sums = got.sum(axis=1)
                    got = batch.normalized()
                    got[sums == 0] = 0.0
            finally:
                batch.close()
            counts[:, :] = got
    finally:
        fasta.close()
    return ids, counts


def count_directory(directory, kmer_length, identifier='fna', symbols=DNA, sum_file=True, sample=0):
    """Counts k-mers of all FASTA files of a directory whose base name contains `identifier`
    (scripts/kmer.py:143-181): one row per file -- with sum_file the column sums over the file's records,
    labelled with the id of its first record (how the reference matrix is regenerated from genome files) --
    as a float array like the reference's.  Unreadable / empty / all-zero files are skipped with a warning;
    `sample` > 0 shuffles the files and stops after that many rows."""
    selected_files = [os.path.join(directory, f) for f in os.listdir(directory) if identifier in os.path.basename(f)]
    if sample:
        random.shuffle(selected_files)
    ids, rows = [], []
    sym = _check_symbols(symbols)
    for path in selected_files:
        # one file = one device-resident batch; with sum_file its column sums are reduced on the device
        # (phk_batch_column_sums) and only the 4^k sums come back -- the per-record count matrix never does
        try:
            fasta = _lib.Fasta(path)
        except IOError:
            logger.warning("Could not read file: %s" % os.path.basename(path))
            continue
        try:
            file_ids = fasta.phamers_ids()
            if fasta.n_records == 0:
                logger.warning("Could not read file: %s" % os.path.basename(path))
                continue
            batch = _lib.Batch.from_fasta(_lib.get_context(), fasta, kmer_length, sym)
        finally:
            fasta.close()
        try:
            if sum_file:
                file_counts = batch.column_sums()
            elif batch.n == 1:
                file_counts = batch.counts()[0]
            else:
                raise ValueError("count_directory(sum_file=False) needs single-record files (%s has %d records)"
                                 % (os.path.basename(path), batch.n))
        finally:
            batch.close()
        if np.sum(file_counts) == 0:
            logger.warning("Could not read file: %s" % os.path.basename(path))
            continue
        ids.append(file_ids[0])
        rows.append(file_counts)
        if sample and len(ids) == sample:
            break
    counts = np.zeros((len(rows), pow(len(symbols), kmer_length)))
    for i, r in enumerate(rows):
        counts[i] = r
    return ids, counts


def read_fasta(fasta_file):
    """(ids, sequences) of a FASTA file (scripts/fileIO.py:28-42), through the native reader."""
    fasta = _lib.Fasta(fasta_file)
    try:
        return fasta.phamers_ids(), fasta.sequences()
    finally:
        fasta.close()


def fasta_lengths(fasta_file):
    """(ids, sequence lengths): what phamer_scorer.screen_by_length needs (scripts/phamer.py:144-157)
    without materialising the sequences as Python strings."""
    fasta = _lib.Fasta(fasta_file, index_only=True)
    try:
        return fasta.phamers_ids(), fasta.lengths()
    finally:
        fasta.close()


def normalize_counts(counts):
    """Row-normalise a count array (scripts/kmer.py:209-221): float64 copy, each row divided
    by its sum; a zero row becomes NaN, exactly as in the reference."""
    counts = np.asarray(counts)
    shape = counts.shape
    if counts.ndim not in (1, 2):
        raise ValueError("normalize_counts expects a 1-D or 2-D array")
    rows = counts.reshape(1, -1) if counts.ndim == 1 else counts
    n, D = rows.shape
    out = np.empty((n, D), dtype=np.float64)
    if n == 0 or D == 0:
        return out.reshape(shape)
    ctx = _lib.get_context()
    if np.issubdtype(rows.dtype, np.integer) or rows.dtype == np.bool_:
        src = np.ascontiguousarray(rows, dtype=np.int64)
        _lib.check(ctx.lib.phk_normalize_i64(ctx.handle, _lib.ptr(src), n, D, _lib.ptr(out)))
    else:
        src = np.ascontiguousarray(rows, dtype=np.float64)
        _lib.check(ctx.lib.phk_normalize_f64(ctx.handle, _lib.ptr(src), n, D, _lib.ptr(out)))
    return out.reshape(shape)


# ---- label utilities (host-side strings; not arithmetic) ---------------------------------
def sequence_to_integers(sequence, symbols):
    """scripts/kmer.py:183-196: non-symbol characters -> '-', symbol i -> str(i)."""
    table = {s: str(i) for i, s in enumerate(symbols)}
    return ''.join(table.get(ch, '-') for ch in sequence)


def get_kmer_index(kmer, symbols):
    """scripts/kmer.py:199-206 (NB the reference parses in base len(kmer), right only when
    k == len(symbols); reproduced as is)."""
    return int(sequence_to_integers(kmer, symbols), len(kmer))


def kmers(k, symbols=DNA):
    """All k-mers in bin order (scripts/kmer.py:224-251)."""
    mers = ['']
    for _ in range(k):
        mers = [m + s for m in mers for s in symbols]
    return mers


def _parser():
    """The counting command line of scripts/kmer.py:283-303 (how the reference's data/reference_features CSVs were made:
    their '#' headers echo this Namespace): positional input (a FASTA file, or a directory of genome files) and output
    CSV; -k, -s / --sample, -sym / --symbols, -id / --file_identifier, -v, --debug."""
    import argparse
    ap = argparse.ArgumentParser(description="This script counts k-mers in sequence data",
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument('input_file', type=str, help='FASTA file, or a directory of FASTA files (one output row per file)')
    ap.add_argument('-id', '--file_identifier', type=str, default='.fna', help='File identifier if directory')
    ap.add_argument('output_file', type=str, help='Output CSV file of k-mer counts')
    ap.add_argument('-k', '--kmer_length', type=int, default=4, help='Length of k-mer to count')
    ap.add_argument('-s', '--sample', type=int, help='Number of sequences to sample and count')
    ap.add_argument('-sym', '--symbols', type=str, default=DNA, help='Symbols to use in k-mer counting')
    ap.add_argument('-v', '--verbose', action='store_true', help='verbose output')
    ap.add_argument('--debug', action='store_true', help='Debug console')
    return ap


def main(argv=None):
    """`python -m phamers_amd.kmer <input> <output.csv> [-k K]` (scripts/kmer.py:283-334): a file goes through
    count_file, a directory through count_directory (column sums per file, reduced on the device); the result is
    written by fileIO.save_counts with the Namespace stamped into the '#' header, as the reference does."""
    from . import fileIO
    args = _parser().parse_args(argv)
    logger.setLevel(logging.DEBUG if args.debug else logging.INFO if args.verbose else logging.WARNING)
    logger.info("Counting k-mers...")
    if args.input_file and os.path.isfile(args.input_file):
        ids, counts = count_file(args.input_file, args.kmer_length, symbols=args.symbols)
        if ids is None:
            raise SystemExit(1)
    elif args.input_file and os.path.isdir(args.input_file):
        ids, counts = count_directory(args.input_file, args.kmer_length, symbols=args.symbols,
                                      identifier=args.file_identifier, sample=args.sample or 0)
    else:
        logger.error("%s was not an acceptable file or directory" % args.input_file)
        raise SystemExit(1)
    fileIO.save_counts(counts, ids, args.output_file, args=args)
    logger.info("K-mer counting complete.")
    return ids, counts


if __name__ == '__main__':
    main()
