"""
transform_kmers.py -- count-vector transforms of PhaMers' scripts/transform_kmers.py:68-88 as a
device column gather: the counts a sequence's reverse, complement or
reverse complement would have produced, obtained by permuting the 4^k columns.

For a k-mer w with digits d_0..d_{k-1} (first base most significant, symbols 'ATGC' = 0..3):
    reverse            : column of d_{k-1}..d_0
    complement         : column of c(d_0)..c(d_{k-1}),  c = A<->T, G<->C  = {0:1, 1:0, 2:3, 3:2}
    reverse complement : both
(the reference builds these index tables with base-k arithmetic that is only right for k = 4,
scripts/transform_kmers.py:43; the tables here are right for every k and equal to it at k = 4).
Identity used by the tests: counts(transform(sequence)) == transform_kmers(counts(sequence)) for
all-valid sequences.
"""
import ctypes

import numpy as np

from . import _lib

_COMPLEMENT = np.array([1, 0, 3, 2])


def transformed_indices(k, reverse=True, complement=False):
    """perm with out[:, j] = counts[:, perm[j]]."""
    j = np.arange(4 ** k)
    digits = np.stack([(j // 4 ** (k - 1 - i)) % 4 for i in range(k)], axis=1)   # d_0 .. d_{k-1}
    if complement:
        digits = _COMPLEMENT[digits]
    if reverse:
        digits = digits[:, ::-1]
    weights = 4 ** np.arange(k - 1, -1, -1)
    return (digits * weights).sum(axis=1).astype(np.uint32)


def transform_kmers(counts, reverse=True, complement=False, symbols='ATGC'):
    """Counts as though the reverse / complement / reverse-complement k-mers had been counted
    (scripts/transform_kmers.py:68-88).  ``counts``: (n, 4^k) array."""
    if symbols != 'ATGC':
        raise NotImplementedError("transform_kmers handles the DNA alphabet 'ATGC'")
    counts = np.asarray(counts)
    if not reverse and not complement:
        return counts
    n, D = counts.shape
    k = int(round(np.log(D) / np.log(4)))
    perm = np.ascontiguousarray(transformed_indices(k, reverse, complement))
    src = np.ascontiguousarray(counts, dtype=np.int64)
    out = np.empty_like(src)
    ctx = _lib.get_context()
    _lib.check(ctx.lib.phk_permute_columns_i64(ctx.handle, _lib.ptr(src), n, D, _lib.ptr(perm), _lib.ptr(out)))
    return out.astype(counts.dtype, copy=False)
