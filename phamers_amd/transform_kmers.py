"""
transform_kmers.py -- count-vector transforms of PhaMers' scripts/transform_kmers.py:68-88 as a device column
gather (host arrays: phk_permute_columns_i64; device-resident batches: phk_batch_gather_columns; SURVEY.md section
8(f)-4).

Two families of index tables:

* ``reference_indices`` -- the table the reference really builds (scripts/transform_kmers.py:21-47 under Python 2
  integer division).  It is NOT a permutation of the 4^k columns: the reference lists k + 1 base-4 digits of a
  column index (a leading zero, then the k digits), uses positions 0 .. k-1 of that list -- so the last digit is
  dropped -- and weights them by k**i ascending.  At k = 4 only 64 distinct source columns occur; at k >= 5 the table
  points past the last column and the reference's gather raises IndexError.  ``transform_kmers`` uses this table by
  default, because this module is a drop-in: same inputs, same outputs (tests/golden/transform.npz holds the
  reference's tables and outputs).
* ``exact_indices`` -- what the docstring of the reference describes: the column of the reversed / complemented /
  reverse-complemented k-mer, a true permutation for every k (``transform_kmers(..., exact=True)``).  For all-valid
  sequences counts(transform(sequence)) == transform_kmers(counts(sequence), exact=True).
"""
import numpy as np

from . import _lib

_COMPLEMENT = np.array([1, 0, 3, 2])   # 'ATGC': A<->T, G<->C


def _digits(k):
    """(4^k, k) base-4 digits of every column index, first base (most significant) first."""
    j = np.arange(4 ** k)
    return np.stack([(j >> (2 * (k - 1 - i))) & 3 for i in range(k)], axis=1)


def exact_indices(k, reverse=True, complement=False):
    """perm with out[:, j] = counts[:, perm[j]]: column of the reversed / complemented k-mer."""
    d = _digits(k)
    if complement:
        d = _COMPLEMENT[d]
    if reverse:
        d = d[:, ::-1]
    return (d << (2 * np.arange(k - 1, -1, -1))).sum(axis=1).astype(np.int64)


def reference_indices(k, reverse=True, complement=False):
    """The reference's table (see the module docstring): digits (0, d_0, .., d_{k-2}) -- reversed when asked,
    complemented when asked -- weighted by k**0, k**1, .., k**(k-1)."""
    d = np.concatenate([np.zeros((4 ** k, 1), dtype=np.int64), _digits(k)[:, : k - 1]], axis=1)
    if reverse:
        d = d[:, ::-1]
    if complement:
        d = _COMPLEMENT[d]
    return (d * (k ** np.arange(k))).sum(axis=1).astype(np.int64)


def _table(D, reverse, complement, exact):
    k = int(round(np.log(D) / np.log(4)))
    table = (exact_indices if exact else reference_indices)(k, reverse, complement)
    if table.max() >= D:
        raise IndexError("index %d is out of bounds for axis 0 with size %d" % (int(table.max()), D))
    return np.ascontiguousarray(table, dtype=np.uint32)


def transform_batch(batch, reverse=True, complement=False, exact=False):
    """transform_kmers on a device-resident batch (``_lib.Batch``): a new resident batch whose counts are the
    transformed ones -- a device-to-device column gather (phk_batch_gather_columns), nothing crosses the bus but the
    4^k-entry table.  Same tables, same IndexError as ``transform_kmers``."""
    if not reverse and not complement:
        return batch
    return batch.gather_columns(_table(batch.D, reverse, complement, exact))


def transform_kmers(counts, reverse=True, complement=False, symbols='ATGC', exact=False):
    """Counts as though the reverse / complement / reverse-complement k-mers had been counted
    (scripts/transform_kmers.py:68-88).  ``counts``: (n, 4^k) array.  Default: the reference's own table, bug for
    bug (IndexError at k >= 5 as there); ``exact=True``: the true permutation."""
    if symbols != 'ATGC':
        raise NotImplementedError("transform_kmers handles the DNA alphabet 'ATGC'")
    counts = np.asarray(counts)
    if not reverse and not complement:
        return counts
    n, D = counts.shape
    perm = _table(D, reverse, complement, exact)
    src = np.ascontiguousarray(counts, dtype=np.int64)
    out = np.empty_like(src)
    ctx = _lib.get_context()
    _lib.check(ctx.lib.phk_permute_columns_i64(ctx.handle, _lib.ptr(src), n, D, _lib.ptr(perm), _lib.ptr(out)))
    return out.astype(counts.dtype, copy=False)
