"""
fileIO.py -- the on-disk formats either side of the hot path, byte-compatible with PhaMers'
scripts/fileIO.py (host I/O):

    read_feature_file(feature_file, normalize=False, id=None)       scripts/fileIO.py:134-166
    save_counts(counts, ids, file_name, args=None, header=...)      scripts/fileIO.py:169-181
    save_phamer_scores(ids, scores, file_name, args=None)           scripts/fileIO.py:241-253
    read_phamer_output(filename)                                    scripts/fileIO.py:256-272
    generate_summary(args, line_start='', header='')                scripts/basic.py:22-37

tests/test_file_formats.py checks the written bytes against files produced by the reference's own
writer functions (tests/golden/files.json).
"""
import numpy as np

from . import kmer


def generate_summary(args, line_start='', header=''):
    """The '#'-header text PhaMers stamps into its output files: the argparse Namespace's repr,
    one 'name:<TAB>value' per line under a title line (scripts/basic.py:22-37)."""
    if args is None:
        return ""
    text = str(args).replace('Namespace(', line_start).replace(')', '')
    text = text.replace(', ', '\n' + line_start).replace('=', ':\t') + '\n'
    return line_start + header + '\n' + text


def read_feature_file(feature_file, normalize=False, id=None):
    """'id,c0,c1,...' rows ('#' comment lines skipped) -> (ids, int features), optionally row
    normalised on the GPU (kmer.normalize_counts); ``id`` selects one row's features."""
    data = np.loadtxt(feature_file, delimiter=',', dtype=str)
    if data.ndim == 1:
        data = np.array([data])
    ids = np.array(list(data[:, 0].transpose()))
    features = data[:, 1:].astype(int)
    if normalize:
        features = kmer.normalize_counts(features)
    if id:
        return features[ids == id]
    return ids, features


def save_counts(counts, ids, file_name, args=None, header='K-mer count file'):
    """One 'id,count,count,...' line per sequence under a '# ' header."""
    if args is not None:
        header = generate_summary(args, header=header)
    table = np.hstack((np.array([ids]).transpose(), np.asarray(counts).astype(int).astype(str)))
    np.savetxt(file_name, table, fmt='%s', delimiter=',', header=header)


def save_phamer_scores(ids, scores, file_name, args=None):
    """'id, score' lines under a '# ' header (phamer_output/phamer_scores.csv)."""
    header = "PhaMers score file"
    if args is not None:
        header = generate_summary(args, header=header)
    table = np.vstack((np.asarray(ids).astype(str), np.asarray(scores).astype(str))).transpose()
    np.savetxt(file_name, table, delimiter=', ', header=header, comments="# ", fmt="%s")


def read_phamer_output(filename):
    """{contig id: score} of a PhaMers score file."""
    out = {}
    for line in open(filename, 'r').readlines():
        if '#' not in line:
            out[line.split(',')[0]] = float(line.split()[1])
    return out
