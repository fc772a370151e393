"""
device.py -- thin Python handles over the device-pointer half of the C ABI
(include/phamers_hip.h "device API"): device buffers without torch, and the
device-resident count -> normalise -> score pipeline that bench.py and the parity tests
drive.  Any integer device address (e.g. ``torch.Tensor.data_ptr()``) can be passed wherever
a DeviceArray is accepted.
"""
import ctypes

import numpy as np

from . import _lib


class DeviceArray(object):
    """A hipMalloc'd buffer with a NumPy dtype/shape attached (owned by ``ctx``)."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = ctypes.c_void_p()
        _lib.check(ctx.lib.phk_malloc(ctx.handle, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_host(cls, ctx, array):
        a = np.ascontiguousarray(array)
        d = cls(ctx, a.shape, a.dtype)
        _lib.check(ctx.lib.phk_memcpy_h2d(ctx.handle, ctypes.c_void_p(d.ptr), _lib.ptr(a), d.nbytes))
        return d

    def to_host(self):
        out = np.empty(self.shape, dtype=self.dtype)
        _lib.check(self.ctx.lib.phk_memcpy_d2h(self.ctx.handle, _lib.ptr(out), ctypes.c_void_p(self.ptr),
                                               self.nbytes))
        return out

    def rows_to_host(self, first, count):
        """Rows first .. first + count of the leading axis (a piece of an array too large to bring down whole)."""
        first, count = int(first), int(count)
        if first < 0 or count < 0 or first + count > self.shape[0]:
            raise IndexError("rows %d..%d of %d" % (first, first + count, self.shape[0]))
        out = np.empty((count,) + self.shape[1:], dtype=self.dtype)
        row = self.nbytes // self.shape[0] if self.shape[0] else 0
        if count:
            _lib.check(self.ctx.lib.phk_memcpy_d2h(self.ctx.handle, _lib.ptr(out), ctypes.c_void_p(self.ptr + first * row),
                                                   count * row))
        return out

    def free(self):
        if getattr(self, "ptr", None) and getattr(self.ctx, "handle", None):
            self.ctx.lib.phk_free(self.ctx.handle, ctypes.c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _p(x):
    if x is None:
        return None
    return ctypes.c_void_p(x.ptr if isinstance(x, DeviceArray) else int(x))


def packed_words(total_bases):
    return (int(total_bases) + 15) // 16 + 1


def mask_words(total_bases):
    return (int(total_bases) + 31) // 32 + 1


def synth_packed(ctx, seed, first_contig, n, L, packed, offsets, mask=None, invalid_ppm=0):
    """Fill ``packed`` / ``mask`` / ``offsets`` with the seeded synthetic batch (phamers_amd/synth.py)."""
    _lib.check(ctx.lib.phk_synth_packed_dev(ctx.handle, int(seed), int(first_contig), int(n), int(L),
                                            int(invalid_ppm), _p(packed), _p(mask), _p(offsets)))


def synth_ragged(ctx, seed, first_contig, n, offsets, total_bases, packed, mask=None, gc_spread_permille=400,
                 invalid_ppm=0):
    """Ragged, composition-skewed seeded batch over the caller's contig boundaries (synth.synth_ragged_codes)."""
    _lib.check(ctx.lib.phk_synth_ragged_dev(ctx.handle, int(seed), int(first_contig), int(n), _p(offsets),
                                            int(total_bases), int(gc_spread_permille), int(invalid_ppm), _p(packed),
                                            _p(mask)))


def pack_ascii(ctx, bases, total_bases, packed, mask, any_invalid=None, symbols="ATGC"):
    _lib.check(ctx.lib.phk_pack_ascii_dev(ctx.handle, _p(bases), int(total_bases), symbols.encode("latin-1"),
                                          _p(packed), _p(mask), _p(any_invalid)))


def count(ctx, packed, mask, total_bases, offsets, n, k, counts, nwin=None):
    _lib.check(ctx.lib.phk_count_dev(ctx.handle, _p(packed), _p(mask), int(total_bases), _p(offsets), int(n),
                                     int(k), _p(counts), _p(nwin)))


def check_counts(ctx, counts, other, n, D, expected_rowsum=None):
    """(rows whose sum differs from expected_rowsum, words differing from `other`), evaluated on the device."""
    res = DeviceArray(ctx, 2, np.uint64)
    _lib.check(ctx.lib.phk_check_counts_dev(ctx.handle, _p(counts), _p(other), int(n), int(D),
                                            0xFFFFFFFFFFFFFFFF if expected_rowsum is None else int(expected_rowsum),
                                            _p(res)))
    out = res.to_host()
    res.free()
    return int(out[0]), int(out[1])


def read_rows(ctx, matrix, rows, D, dtype=np.uint32):
    """Rows `rows` of a device-resident [n][D] matrix, fetched one by one (sampling a batch too large to download)."""
    out = np.empty((len(rows), D), dtype=dtype)
    base = matrix.ptr if isinstance(matrix, DeviceArray) else int(matrix)
    item = np.dtype(dtype).itemsize
    for i, r in enumerate(rows):
        _lib.check(ctx.lib.phk_memcpy_d2h(ctx.handle, ctypes.c_void_p(out[i].ctypes.data),
                                          ctypes.c_void_p(base + int(r) * D * item), D * item))
    return out


def normalize(ctx, counts, n, D, out):
    _lib.check(ctx.lib.phk_normalize_dev(ctx.handle, _p(counts), int(n), int(D), _p(out)))


def score(ctx, model, Q, N, method, scores, status=None):
    _lib.check(ctx.lib.phk_score_dev(ctx.handle, model.handle, _p(Q), int(N), _lib.METHODS[method], _p(scores),
                                     _p(status)))


def score_counts(ctx, model, counts, N, method, scores, status=None):
    _lib.check(ctx.lib.phk_score_counts_dev(ctx.handle, model.handle, _p(counts), int(N), _lib.METHODS[method],
                                            _p(scores), _p(status)))


def count_score(ctx, model, packed, mask, total_bases, offsets, n, k, method, counts, scores, status=None):
    """The whole hot path on device-resident input (phk_count_score_dev)."""
    _lib.check(ctx.lib.phk_count_score_dev(ctx.handle, model.handle, _p(packed), _p(mask), int(total_bases),
                                           _p(offsets), int(n), int(k), _lib.METHODS[method], _p(counts),
                                           _p(scores), _p(status)))
