"""
fileIO.py -- the on-disk formats either side of the hot path, byte-compatible with PhaMers'
scripts/fileIO.py (host I/O):

    read_feature_file(feature_file, normalize=False, id=None)       scripts/fileIO.py:134-166
    save_counts(counts, ids, file_name, args=None, header=...)      scripts/fileIO.py:169-181
    save_phamer_scores(ids, scores, file_name, args=None)           scripts/fileIO.py:241-253
    read_phamer_output(filename)                                    scripts/fileIO.py:256-272
    generate_summary(args, line_start='', header='')                scripts/basic.py:22-37

The writers are native (csrc/csvio.cpp: rows formatted on all cores -- the features cache of a 1 M-contig FASTA is
~0.7 GB of text); what they must reproduce is the output of the reference's np.savetxt calls, and
tests/test_file_formats.py compares the bytes with files written by the reference's own functions
(tests/golden/files.json).

Which runtime that pins: the fixtures were produced by the reference's functions under Python 3 with NumPy >= 1.14,
where `scores.astype(str)` / `str(numpy.float64)` give the shortest round-trip digits -- the notation the native writer
reproduces (a 300 000-value fuzz agrees).  The reference itself is a Python 2.7 script; under its era's NumPy the same
calls print 12 significant digits.  No file the reference ships covers this, so byte compatibility with the ORIGINAL
runtime's score text is unpinned; `read_phamer_output` parses either form, and the values agree to those 12 digits.
"""
import numpy as np

from . import _lib


def generate_summary(args, line_start='', header=''):
    """The text PhaMers stamps into its output files (scripts/basic.py:22-37): `header` on the first line, then one
    'name:<TAB>value' line per field of the argparse Namespace, derived from its repr."""
    if args is None:
        return ""
    fields = str(args)
    for old, new in (('Namespace(', line_start), (')', ''), (', ', '\n' + line_start), ('=', ':\t')):
        fields = fields.replace(old, new)
    return line_start + header + '\n' + fields + '\n'


def _comment_block(header, comments='# '):
    """What np.savetxt writes for header=...: every line of it behind `comments`, then a newline."""
    return (comments + header.replace('\n', '\n' + comments) + '\n').encode('latin-1')


def _id_table(ids):
    """ids (any sequence) -> (concatenated bytes, offsets[n+1]) for the native writers."""
    raw = [str(i).encode('latin-1', 'replace') for i in np.asarray(ids).tolist()]
    offsets = np.zeros(len(raw) + 1, dtype=np.uint64)
    if raw:
        offsets[1:] = np.cumsum([len(r) for r in raw], dtype=np.uint64)
    return np.frombuffer(b''.join(raw) or b'\0', dtype=np.uint8), offsets


def _ucs4_ids(ids, n_rows):
    """A NumPy 'U' id array (what count_file / read_feature_file / the FASTA reader return) as the native writers take
    it -- (code point matrix, width in characters) -- or None for any other kind of id sequence."""
    a = ids if isinstance(ids, np.ndarray) else None
    if a is None or a.dtype.kind != 'U' or a.ndim != 1 or a.dtype.itemsize == 0:
        return None
    if a.shape[0] != n_rows:
        raise ValueError("%d ids for %d rows" % (a.shape[0], n_rows))
    a = np.ascontiguousarray(a)
    if not a.dtype.isnative:
        a = a.astype(a.dtype.newbyteorder('='))
    return a.view(np.uint32), a.dtype.itemsize // 4


def _read_feature_file_native(feature_file):
    """(ids, int64 features) of a file of the shape save_counts writes, parsed on all cores by the native reader
    (np.loadtxt needs minutes for the features cache of 10^6 contigs); None for any other shape of file, which
    np.loadtxt then reads as the reference does."""
    import ctypes
    lib = _lib.load()
    handle = ctypes.c_void_p()
    n, D, w = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    rc = lib.phk_features_open(str(feature_file).encode(), ctypes.byref(handle), ctypes.byref(n), ctypes.byref(D),
                               ctypes.byref(w))
    if rc == _lib.PHK_ERR_UNSUPPORTED:
        return None
    if rc == _lib.PHK_ERR_IO:
        np.loadtxt(feature_file, delimiter=',', dtype=str)   # raises what the reference raises for this path
        return None
    _lib.check(rc)
    try:
        if n.value == 0:
            return None          # np.loadtxt's empty-file behaviour (a warning and an empty array) stays np.loadtxt's
        features = np.empty((n.value, D.value), dtype=np.int64)
        idb = np.zeros((n.value, w.value), dtype=np.uint8)
        rc = lib.phk_features_read(handle, _lib.ptr(features), _lib.ptr(idb), w.value)
        if rc == _lib.PHK_ERR_UNSUPPORTED:
            return None
        _lib.check(rc)
    finally:
        lib.phk_features_close(handle)
    # (ASCII bytes -> code points -> a 'U' view: no per-id Python work)
    ids = np.ascontiguousarray(idb.astype(np.uint32)).view("U%d" % w.value)[:, 0]
    return ids, features


def read_feature_file(feature_file, normalize=False, id=None):
    """'id,c0,c1,...' rows ('#' comment lines skipped) -> (ids, int features), optionally row
    normalised on the GPU (kmer.normalize_counts); ``id`` selects one row's features."""
    native = _read_feature_file_native(feature_file)
    if native is not None:
        ids, features = native
    else:   # any other shape of file: the reference's own parse
        data = np.atleast_2d(np.loadtxt(feature_file, delimiter=',', dtype=str))
        ids = np.array(list(data[:, 0]))
        features = data[:, 1:].astype(int)
    if normalize:
        from . import kmer
        features = kmer.normalize_counts(features)
    return features[ids == id] if id else (ids, features)


def save_counts(counts, ids, file_name, args=None, header='K-mer count file'):
    """One 'id,count,count,...' line per sequence under a '# ' header block.  ``counts``: any integer (or
    integer-valued) matrix; uint32 / int64 go to the writer as they are."""
    if args is not None:
        header = generate_summary(args, header=header)
    counts = np.asarray(counts)
    if counts.ndim == 1:
        counts = counts[None, :]
    if counts.dtype not in (np.dtype(np.uint32), np.dtype(np.int64)):
        counts = counts.astype(np.int64)
    counts = np.ascontiguousarray(counts)
    lib = _lib.load()
    u = _ucs4_ids(ids, counts.shape[0])
    if u is not None:
        _lib.check(lib.phk_write_counts_csv_ucs4(str(file_name).encode(), _comment_block(header), _lib.ptr(u[0]), u[1],
                                                 _lib.ptr(counts), counts.dtype.itemsize, counts.shape[0], counts.shape[1]))
        return
    idb, ido = _id_table(ids)
    if len(ido) - 1 != counts.shape[0]:
        raise ValueError("%d ids for %d rows" % (len(ido) - 1, counts.shape[0]))
    _lib.check(lib.phk_write_counts_csv(str(file_name).encode(), _comment_block(header), _lib.ptr(idb), _lib.ptr(ido),
                                        _lib.ptr(counts), counts.dtype.itemsize, counts.shape[0], counts.shape[1]))


def save_phamer_scores(ids, scores, file_name, args=None):
    """'id, score' lines under a '# ' header block (phamer_output/phamer_scores.csv); scores in str(float64) notation."""
    header = "PhaMers score file"
    if args is not None:
        header = generate_summary(args, header=header)
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    lib = _lib.load()
    u = _ucs4_ids(ids, scores.shape[0])
    if u is not None:
        _lib.check(lib.phk_write_scores_csv_ucs4(str(file_name).encode(), _comment_block(header), _lib.ptr(u[0]), u[1],
                                                 _lib.ptr(scores), scores.shape[0]))
        return
    idb, ido = _id_table(ids)
    if len(ido) - 1 != scores.shape[0]:
        raise ValueError("%d ids for %d scores" % (len(ido) - 1, scores.shape[0]))
    _lib.check(lib.phk_write_scores_csv(str(file_name).encode(), _comment_block(header), _lib.ptr(idb), _lib.ptr(ido),
                                        _lib.ptr(scores), scores.shape[0]))


def read_phamer_output(filename):
    """{contig id: score} of a PhaMers score file."""
    with open(filename, 'r') as f:
        pairs = (line.split(',', 1) for line in f if '#' not in line and ',' in line)
        return {cid: float(val.split()[0]) for cid, val in pairs}
