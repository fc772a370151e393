"""
ctypes binding of libphamers_hip.so (include/phamers_hip.h).

The library is the product: there is no CPU fallback.  If the shared object is
missing, or no gfx950 device can be opened, every operation raises.
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libphamers_hip.so")

PHK_OK = 0
PHK_ERR_ARG, PHK_ERR_HIP, PHK_ERR_NOMEM, PHK_ERR_UNSUPPORTED, PHK_ERR_NAN, PHK_ERR_IO = -1, -2, -3, -4, -5, -6
METHOD_KNN, METHOD_KMEANS, METHOD_COMBO = 1, 2, 3
METHODS = {"knn": METHOD_KNN, "kmeans": METHOD_KMEANS, "combo": METHOD_COMBO}
MAX_K = 7
ABI_VERSION = 2

c_void_p, c_int, c_u32, c_u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64
c_char_p, c_double = ctypes.c_char_p, ctypes.c_double
P = ctypes.POINTER

# name -> (restype, argtypes); one entry per function declared in include/phamers_hip.h
SIGNATURES = {
    "phk_abi_version": (c_int, []),
    "phk_last_error": (c_char_p, []),
    "phk_device_count": (c_int, [P(c_int)]),
    "phk_create": (c_int, [c_int, c_void_p, P(c_void_p)]),
    "phk_destroy": (c_int, [c_void_p]),
    "phk_sync": (c_int, [c_void_p]),
    "phk_set_option": (c_int, [c_void_p, c_char_p, c_char_p]),
    "phk_malloc": (c_int, [c_void_p, c_u64, P(c_void_p)]),
    "phk_free": (c_int, [c_void_p, c_void_p]),
    "phk_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_void_p, c_u64]),
    "phk_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_void_p, c_u64]),
    "phk_count_ascii": (c_int, [c_void_p, c_void_p, c_void_p, c_u64, c_int, c_char_p, c_void_p]),
    "phk_normalize_i64": (c_int, [c_void_p, c_void_p, c_u64, c_u64, c_void_p]),
    "phk_normalize_f64": (c_int, [c_void_p, c_void_p, c_u64, c_u64, c_void_p]),
    "phk_permute_columns_i64": (c_int, [c_void_p, c_void_p, c_u64, c_u64, c_void_p, c_void_p]),
    "phk_fasta_read": (c_int, [c_char_p, c_int, P(c_void_p)]),
    "phk_fasta_index": (c_int, [c_char_p, c_int, P(c_void_p)]),
    "phk_fasta_read_range": (c_int, [c_char_p, c_u64, c_u64, c_int, P(c_void_p)]),
    "phk_fasta_read_part": (c_int, [c_char_p, c_u32, c_u32, c_int, P(c_void_p)]),
    "phk_fasta_shape": (c_int, [c_void_p, P(c_u64), P(c_u64), P(c_u64)]),
    "phk_fasta_data": (c_int, [c_void_p, P(c_void_p), P(c_void_p), P(c_void_p), P(c_void_p)]),
    "phk_fasta_free": (c_int, [c_void_p]),
    "phk_parse_id": (c_int, [c_char_p, c_u64, c_char_p, c_u64, P(c_u64), P(c_int)]),
    "phk_fasta_ids": (c_int, [c_void_p, P(c_void_p), P(c_void_p), P(c_void_p)]),
    "phk_fasta_ids_fixed": (c_int, [c_void_p, c_u64, c_void_p]),
    "phk_count_fasta": (c_int, [c_void_p, c_void_p, c_int, c_char_p, c_void_p]),
    "phk_batch_from_ascii": (c_int, [c_void_p, c_void_p, c_void_p, c_u64, c_int, c_char_p, P(c_void_p)]),
    "phk_batch_from_fasta": (c_int, [c_void_p, c_void_p, c_int, c_char_p, P(c_void_p)]),
    "phk_batch_from_counts": (c_int, [c_void_p, c_void_p, c_u64, c_u64, P(c_void_p)]),
    "phk_batch_from_fasta_file": (c_int, [c_void_p, c_char_p, c_int, c_char_p, c_int, P(c_void_p), P(c_void_p)]),
    "phk_batch_from_fasta_part": (c_int, [c_void_p, c_char_p, c_u32, c_u32, c_int, c_char_p, c_int, P(c_void_p), P(c_void_p)]),
    "phk_batch_shape": (c_int, [c_void_p, P(c_u64), P(c_u64), P(c_u64), P(c_int)]),
    "phk_batch_device_ptrs": (c_int, [c_void_p, P(c_void_p), P(c_void_p)]),
    "phk_batch_counts_i64": (c_int, [c_void_p, c_void_p, c_void_p]),
    "phk_batch_counts_u32": (c_int, [c_void_p, c_void_p, c_void_p]),
    "phk_write_counts_csv": (c_int, [c_char_p, c_char_p, c_void_p, c_void_p, c_void_p, c_int, c_u64, c_u64]),
    "phk_write_scores_csv": (c_int, [c_char_p, c_char_p, c_void_p, c_void_p, c_void_p, c_u64]),
    "phk_write_counts_csv_ucs4": (c_int, [c_char_p, c_char_p, c_void_p, c_u64, c_void_p, c_int, c_u64, c_u64]),
    "phk_write_scores_csv_ucs4": (c_int, [c_char_p, c_char_p, c_void_p, c_u64, c_void_p, c_u64]),
    "phk_format_float": (c_int, [c_double, c_char_p, c_int]),
    "phk_features_open": (c_int, [c_char_p, P(c_void_p), P(c_u64), P(c_u64), P(c_u64)]),
    "phk_features_read": (c_int, [c_void_p, c_void_p, c_void_p, c_u64]),
    "phk_features_close": (c_int, [c_void_p]),
    "phk_batch_normalized": (c_int, [c_void_p, c_void_p, c_void_p]),
    "phk_batch_select": (c_int, [c_void_p, c_void_p, c_void_p, c_u64, P(c_void_p)]),
    "phk_batch_column_sums": (c_int, [c_void_p, c_void_p, c_void_p]),
    "phk_batch_gather_columns": (c_int, [c_void_p, c_void_p, c_void_p, P(c_void_p)]),
    "phk_batch_score": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "phk_batch_free": (c_int, [c_void_p, c_void_p]),
    "phk_kmeans": (c_int, [c_void_p, c_void_p, c_u64, c_u64, c_u32, c_u64, c_int, c_void_p, c_void_p, P(c_int)]),
    "phk_kmeans_lloyd": (c_int, [c_void_p, c_void_p, c_u64, c_u64, c_u32, c_void_p, c_double, c_int, c_void_p, c_void_p, P(c_int),
                                 P(c_int), P(c_double)]),
    "phk_model_create": (c_int, [c_void_p, c_void_p, c_u64, c_void_p, c_u64, c_void_p, c_u64,
                                 c_void_p, c_u64, c_u64, c_int, P(c_void_p)]),
    "phk_model_destroy": (c_int, [c_void_p, c_void_p]),
    "phk_model_set_centroids": (c_int, [c_void_p, c_void_p, c_void_p, c_u64, c_void_p, c_u64]),
    "phk_model_set_column_mask": (c_int, [c_void_p, c_void_p, c_void_p]),
    "phk_score": (c_int, [c_void_p, c_void_p, c_void_p, c_u64, c_int, c_void_p]),
    "phk_distances": (c_int, [c_void_p, c_void_p, c_u64, c_void_p, c_u64, c_u64, c_void_p]),
    "phk_pack_ascii_dev": (c_int, [c_void_p, c_void_p, c_u64, c_char_p, c_void_p, c_void_p, c_void_p]),
    "phk_count_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_u64, c_void_p, c_u64, c_int,
                              c_void_p, c_void_p]),
    "phk_normalize_dev": (c_int, [c_void_p, c_void_p, c_u64, c_u64, c_void_p]),
    "phk_score_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_u64, c_int, c_void_p, c_void_p]),
    "phk_score_counts_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_u64, c_int, c_void_p, c_void_p]),
    "phk_count_score_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_u64, c_void_p, c_u64,
                                    c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "phk_check_counts_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_u64, c_u64, c_u64, c_void_p]),
    "phk_score_stats": (c_int, [c_void_p, P(c_u64), P(c_u64)]),
    "phk_score_stats_ex": (c_int, [c_void_p, c_void_p, c_int]),
    "phk_mfma_f16_probe": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_u64, c_u32, c_void_p]),
    "phk_synth_packed_dev": (c_int, [c_void_p, c_u64, c_u64, c_u64, c_u64, c_u32, c_void_p,
                                     c_void_p, c_void_p]),
    "phk_synth_ragged_dev": (c_int, [c_void_p, c_u64, c_u64, c_u64, c_void_p, c_u64, c_u32, c_u32, c_void_p, c_void_p]),
    "phk_profile_enable": (c_int, [c_void_p, c_int]),
    "phk_profile_reset": (c_int, [c_void_p]),
    "phk_profile_count": (c_int, [c_void_p, P(c_int)]),
    "phk_profile_get": (c_int, [c_void_p, c_int, c_char_p, c_int, P(c_double), P(c_u64)]),
}

_lib = None
_lock = threading.Lock()
_contexts = {}


class PhkError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "libphamers_hip: %s (code %d)" % (message, code))
        self.code = code


def load():
    """dlopen the in-tree library and declare every signature; raises if it is absent."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                path = LIB_PATH
                # A/B timing runs on one box (tools/diag): another build of the library -- an earlier round's, a candidate's --
                # in place of the in-tree one.  Only together with PHK_ALLOW_DIAGNOSTIC_BUILD=1, never for results; entry points
                # such a build lacks stay unbound.
                other = os.environ.get("PHAMERS_AB_LIB") if os.environ.get("PHK_ALLOW_DIAGNOSTIC_BUILD") == "1" else None
                if other:
                    path = other
                if not os.path.exists(path):
                    raise ImportError(
                        "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(or `make -C phamers_amd/csrc`). There is no CPU fallback." % path)
                lib = ctypes.CDLL(path)
                for name, (res, args) in SIGNATURES.items():
                    if other and not hasattr(lib, name):
                        continue
                    fn = getattr(lib, name)
                    fn.restype = res
                    fn.argtypes = args
                ver = lib.phk_abi_version()
                if other:
                    ver = ABI_VERSION
                if ver != ABI_VERSION and not (ver == (ABI_VERSION | 0x40000000)
                                               and os.environ.get("PHK_ALLOW_DIAGNOSTIC_BUILD") == "1"):
                    raise ImportError(
                        "%s reports ABI version %#x, this binding is for %d%s" % (
                            LIB_PATH, ver, ABI_VERSION,
                            " (a -DPHK_DIAGNOSTIC_BUILD library: timing only, never for results; tools/diag scripts "
                            "load it with PHK_ALLOW_DIAGNOSTIC_BUILD=1)" if ver & 0x40000000 else ""))
                _lib = lib
    return _lib


def check(rc):
    if rc != PHK_OK:
        raise PhkError(rc, load().phk_last_error().decode("utf-8", "replace"))


def default_device():
    for var in ("PHAMERS_HIP_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(var)
        if v not in (None, ""):
            return int(v)
    return 0


def ptr(a):
    """void* of a C-contiguous NumPy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


class Context(object):
    """One per (process, GPU).  ``stream`` may be a raw hipStream_t integer
    (e.g. ``torch.cuda.current_stream().cuda_stream``)."""

    def __init__(self, device=None, stream=None):
        self.lib = load()
        self.device = default_device() if device is None else int(device)
        h = ctypes.c_void_p()
        check(self.lib.phk_create(self.device, ctypes.c_void_p(stream) if stream else None, ctypes.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.phk_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.lib.phk_sync(self.handle))

    def set_option(self, key, value):
        """Tuning / diagnostic knob (phk_set_option); knobs start from PHK_<KEY> in the environment at creation."""
        check(self.lib.phk_set_option(self.handle, key.encode(), str(value).encode()))

    def options(self, **kw):
        """Context manager: set knobs for a block, restore the given ``(value, restore)`` pairs afterwards:
        ``with ctx.options(force_exact=("1", "0")): ...``"""
        ctx = self

        class _Scope(object):
            def __enter__(self_):
                for k, (v, _) in kw.items():
                    ctx.set_option(k, v)

            def __exit__(self_, *exc):
                for k, (_, r) in kw.items():
                    ctx.set_option(k, r)
        return _Scope()

    def score_stats(self):
        """(queries sent to the float64 fallback, orderings decided by exact candidate distances)
        of the most recent scoring call."""
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        check(self.lib.phk_score_stats(self.handle, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def score_stats_ex(self):
        """Diagnostics of the most recent scoring call (phk_score_stats_ex) as a dict."""
        out = np.zeros(9, dtype=np.uint64)
        check(self.lib.phk_score_stats_ex(self.handle, ptr(out), 9))
        names = ("brute_forced", "exact_distance_decisions", "second_chance", "window_wider_than_refined",
                 "window_past_lists", "refined_values_too_close", "centroid_leader_uncertified",
                 "reswept_three_digits", "swept_f16_beyond_int8")
        return {k: int(v) for k, v in zip(names, out)}

    # ---- timing ----
    def profile_enable(self, on=True):
        check(self.lib.phk_profile_enable(self.handle, 1 if on else 0))

    def profile_reset(self):
        check(self.lib.phk_profile_reset(self.handle))

    def profile(self):
        """{kernel name: (total_ms, launches)} since the last reset."""
        n = ctypes.c_int()
        check(self.lib.phk_profile_count(self.handle, ctypes.byref(n)))
        out = {}
        for i in range(n.value):
            name = ctypes.create_string_buffer(128)
            ms, cnt = ctypes.c_double(), ctypes.c_uint64()
            check(self.lib.phk_profile_get(self.handle, i, name, 128, ctypes.byref(ms), ctypes.byref(cnt)))
            out[name.value.decode()] = (ms.value, cnt.value)
        return out


class Model(object):
    """Device-resident training side of phamer_scorer.score_points."""

    def __init__(self, ctx, positive, negative, positive_centroids=None, negative_centroids=None,
                 k_neighbors=3):
        self.ctx = ctx
        pos = np.ascontiguousarray(positive, dtype=np.float64)
        neg = np.ascontiguousarray(negative, dtype=np.float64)
        if pos.ndim != 2 or neg.ndim != 2 or pos.shape[1] != neg.shape[1]:
            raise ValueError("positive / negative data must be 2-D with the same number of columns")
        cp = cn = None
        if positive_centroids is not None:
            cp = np.ascontiguousarray(positive_centroids, dtype=np.float64)
            cn = np.ascontiguousarray(negative_centroids, dtype=np.float64)
        # a zero-count reference row is NaN after normalisation; the reference's scikit-learn fit raises on it
        # (scripts/learning.py:127, 138), so no model is ever built from such data
        for a in (pos, neg, cp, cn):
            if a is not None and np.isnan(a).any():
                raise ValueError("Input contains NaN.")
        self.D = pos.shape[1]
        h = ctypes.c_void_p()
        check(ctx.lib.phk_model_create(ctx.handle, ptr(pos), pos.shape[0], ptr(neg), neg.shape[0],
                                       ptr(cp), 0 if cp is None else cp.shape[0],
                                       ptr(cn), 0 if cn is None else cn.shape[0],
                                       self.D, int(k_neighbors), ctypes.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx.lib.phk_model_destroy(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_centroids(self, positive_centroids, negative_centroids):
        """Replace the centroid segments (same counts as at creation): a cross-validation fold's k-means result."""
        cp = np.ascontiguousarray(positive_centroids, dtype=np.float64)
        cn = np.ascontiguousarray(negative_centroids, dtype=np.float64)
        if np.isnan(cp).any() or np.isnan(cn).any():
            raise ValueError("Input contains NaN.")
        check(self.ctx.lib.phk_model_set_centroids(self.ctx.handle, self.handle, ptr(cp), cp.shape[0], ptr(cn), cn.shape[0]))

    def set_column_mask(self, mask):
        """Exclude train rows from the k-NN search (``mask``: bool over vstack(positive, negative); None lifts it)."""
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        check(self.ctx.lib.phk_model_set_column_mask(self.ctx.handle, self.handle, ptr(m)))

    def score(self, Q, method="combo"):
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        if Q.ndim != 2 or Q.shape[1] != self.D:
            raise ValueError("query rows must be (N, %d)" % self.D)
        out = np.empty(Q.shape[0], dtype=np.float64)
        check(self.ctx.lib.phk_score(self.ctx.handle, self.handle, ptr(Q), Q.shape[0], METHODS[method], ptr(out)))
        return out


class Batch(object):
    """Device-resident contig batch (phk_batch): counts + row sums in HBM; see include/phamers_hip.h."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.handle = handle
        n, D, T, inv = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_int()
        check(ctx.lib.phk_batch_shape(handle, ctypes.byref(n), ctypes.byref(D), ctypes.byref(T), ctypes.byref(inv)))
        self.n, self.D, self.total_bases, self.any_invalid = n.value, D.value, T.value, bool(inv.value)

    @classmethod
    def from_fasta(cls, ctx, fasta, kmer_length, symbols=b"ATGC"):
        h = ctypes.c_void_p()
        check(ctx.lib.phk_batch_from_fasta(ctx.handle, fasta.handle, int(kmer_length), symbols, ctypes.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_sequences(cls, ctx, sequences, kmer_length, symbols=b"ATGC"):
        raw = [s.encode("latin-1", "replace") for s in sequences]
        offsets = np.zeros(len(raw) + 1, dtype=np.uint64)
        if raw:
            offsets[1:] = np.cumsum([len(r) for r in raw], dtype=np.uint64)
        bases = np.frombuffer(b"".join(raw) or b"\0", dtype=np.uint8)
        h = ctypes.c_void_p()
        check(ctx.lib.phk_batch_from_ascii(ctx.handle, ptr(np.ascontiguousarray(bases)), ptr(offsets), len(raw),
                                           int(kmer_length), symbols, ctypes.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_counts(cls, ctx, counts):
        """A batch from an integer count matrix (n, 4^k) on the host -- the features cache of an earlier run (phk_batch_from_counts).
        None when the matrix is not such a one (another width, negative or huge entries): the caller keeps the float rows."""
        c = np.ascontiguousarray(counts, dtype=np.int64)
        if c.ndim != 2:
            return None
        h = ctypes.c_void_p()
        rc = ctx.lib.phk_batch_from_counts(ctx.handle, ptr(c), c.shape[0], c.shape[1], ctypes.byref(h))
        if rc == PHK_ERR_UNSUPPORTED:
            return None
        check(rc)
        return cls(ctx, h)

    def counts(self):
        """int64 (n, 4^k) on the host: what kmer.count_file returns (one download, widened on the device)."""
        out = np.zeros((self.n, self.D), dtype=np.int64)
        check(self.ctx.lib.phk_batch_counts_i64(self.ctx.handle, self.handle, ptr(out)))
        return out

    def counts_u32(self):
        """uint32 (n, 4^k) on the host, as the device holds them (what the features-cache writer takes)."""
        out = np.zeros((self.n, self.D), dtype=np.uint32)
        check(self.ctx.lib.phk_batch_counts_u32(self.ctx.handle, self.handle, ptr(out)))
        return out

    def normalized(self):
        out = np.empty((self.n, self.D), dtype=np.float64)
        check(self.ctx.lib.phk_batch_normalized(self.ctx.handle, self.handle, ptr(out)))
        return out

    def select(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        h = ctypes.c_void_p()
        check(self.ctx.lib.phk_batch_select(self.ctx.handle, self.handle, ptr(rows), rows.shape[0], ctypes.byref(h)))
        return Batch(self.ctx, h)

    def column_sums(self):
        """int64 (4^k,): the column sums over the batch's rows, reduced on the device (phk_batch_column_sums)."""
        out = np.zeros(self.D, dtype=np.int64)
        check(self.ctx.lib.phk_batch_column_sums(self.ctx.handle, self.handle, ptr(out)))
        return out

    def gather_columns(self, table):
        """A new resident batch whose column j is this batch's column table[j] (phk_batch_gather_columns)."""
        table = np.ascontiguousarray(table, dtype=np.uint32)
        if table.shape != (self.D,):
            raise ValueError("gather_columns: the table must have %d entries" % self.D)
        h = ctypes.c_void_p()
        check(self.ctx.lib.phk_batch_gather_columns(self.ctx.handle, self.handle, ptr(table), ctypes.byref(h)))
        return Batch(self.ctx, h)

    def score(self, model, method="combo"):
        out = np.empty(self.n, dtype=np.float64)
        rc = self.ctx.lib.phk_batch_score(self.ctx.handle, model.handle, self.handle, METHODS[method], ptr(out))
        if rc == PHK_ERR_NAN:
            raise ValueError("Input contains NaN.")   # what scikit-learn raises for the reference
        check(rc)
        return out

    def close(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx.lib.phk_batch_free(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Fasta(object):
    """A FASTA file parsed by the native multi-threaded reader (phk_fasta_read)."""

    def __init__(self, path, threads=0, part=None, byte_range=None, index_only=False):
        """``part`` = (i, n): only the records that begin in the i-th of n equal byte ranges of the file (one rank's
        share, phk_fasta_read_part); ``byte_range`` = (lo, hi): those that begin in [lo, hi) (phk_fasta_read_range);
        ``index_only``: ids, titles and lengths without the sequences (phk_fasta_index)."""
        self.lib = load()
        h = ctypes.c_void_p()
        if index_only:
            rc = self.lib.phk_fasta_index(os.fsencode(path), int(threads), ctypes.byref(h))
        elif part is not None:
            rc = self.lib.phk_fasta_read_part(os.fsencode(path), int(part[0]), int(part[1]), int(threads), ctypes.byref(h))
        elif byte_range is not None:
            hi = 0xFFFFFFFFFFFFFFFF if byte_range[1] is None else int(byte_range[1])
            rc = self.lib.phk_fasta_read_range(os.fsencode(path), int(byte_range[0]), hi, int(threads), ctypes.byref(h))
        else:
            rc = self.lib.phk_fasta_read(os.fsencode(path), int(threads), ctypes.byref(h))
        if rc == PHK_ERR_IO:
            raise IOError(self.lib.phk_last_error().decode("utf-8", "replace"))
        check(rc)
        self._attach(h)

    def _attach(self, h):
        self.handle = h
        n, t, tb = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        check(self.lib.phk_fasta_shape(h, ctypes.byref(n), ctypes.byref(t), ctypes.byref(tb)))
        self.n_records, self.total_bases, self.title_bytes = n.value, t.value, tb.value
        b, o, ti, to = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        check(self.lib.phk_fasta_data(h, ctypes.byref(b), ctypes.byref(o), ctypes.byref(ti), ctypes.byref(to)))
        self._bases, self._offsets, self._titles, self._title_off = b.value, o.value, ti.value, to.value

    @classmethod
    def count_file(cls, ctx, path, kmer_length, symbols=b"ATGC", threads=0, part=None):
        """(index, batch) of a FASTA file in one call (phk_batch_from_fasta_file): the sequences are parsed straight into the
        upload's staging buffers -- no host copy of them exists; ``index`` is a Fasta as ``index_only=True`` gives it (titles,
        ids, lengths).  ``part`` = (i, n): the records that begin in the i-th of n equal byte ranges of the file only (one
        rank's share, phk_batch_from_fasta_part).  IOError when the file cannot be read."""
        self = cls.__new__(cls)
        self.lib = load()
        h, hb = ctypes.c_void_p(), ctypes.c_void_p()
        if part is not None:
            rc = self.lib.phk_batch_from_fasta_part(ctx.handle, os.fsencode(path), int(part[0]), int(part[1]), int(kmer_length),
                                                    symbols, int(threads), ctypes.byref(h), ctypes.byref(hb))
        else:
            rc = self.lib.phk_batch_from_fasta_file(ctx.handle, os.fsencode(path), int(kmer_length), symbols, int(threads),
                                                    ctypes.byref(h), ctypes.byref(hb))
        if rc == PHK_ERR_IO:
            raise IOError(self.lib.phk_last_error().decode("utf-8", "replace"))
        check(rc)
        self._attach(h)
        return self, Batch(ctx, hb)

    def offsets(self):
        return np.ctypeslib.as_array(ctypes.cast(self._offsets, ctypes.POINTER(ctypes.c_uint64)),
                                     shape=(self.n_records + 1,)).copy()

    def lengths(self):
        return np.diff(self.offsets().astype(np.int64))

    def titles(self):
        off = np.ctypeslib.as_array(ctypes.cast(self._title_off, ctypes.POINTER(ctypes.c_uint64)),
                                    shape=(self.n_records + 1,))
        raw = ctypes.string_at(self._titles, self.title_bytes) if self.title_bytes else b""
        return [raw[int(off[i]):int(off[i + 1])].decode("latin-1") for i in range(self.n_records)]

    def ids(self):
        """Bio.SeqIO record.id: the first white-space delimited word of each title."""
        return [(t.split(None, 1) or [""])[0] for t in self.titles()]

    def phamers_ids(self):
        """What fileIO.get_fasta_ids returns (scripts/fileIO.py:62-77): id_parser.get_id of every record.id,
        parsed by the native reader.  Raises IndexError where the reference does (a header that matches none of
        its three shapes); a header for which the reference returns None gives None (object array)."""
        n = self.n_records
        ids, off, st = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        check(self.lib.phk_fasta_ids(self.handle, ctypes.byref(ids), ctypes.byref(off), ctypes.byref(st)))
        if n == 0:
            return np.array([])
        status = np.ctypeslib.as_array(ctypes.cast(st.value, ctypes.POINTER(ctypes.c_uint8)), shape=(n,))
        if (status == 1).any():
            bad = int(np.argmax(status == 1))
            raise IndexError("list index out of range (FASTA header %r has none of the id shapes)" % self.titles_at(bad))
        offs = np.ctypeslib.as_array(ctypes.cast(off.value, ctypes.POINTER(ctypes.c_uint64)), shape=(n + 1,))
        width = max(int(np.diff(offs.astype(np.int64)).max()), 1)
        fixed = np.zeros(n, dtype="S%d" % width)
        check(self.lib.phk_fasta_ids_fixed(self.handle, width, ptr(fixed)))
        # latin-1 decode = code point identity: bytes -> uint32 -> 'U' view, without a Python loop over the ids
        out = np.ascontiguousarray(fixed.view(np.uint8).reshape(n, width).astype(np.uint32)).view("U%d" % width)[:, 0]
        if (status == 2).any():
            out = out.astype(object)
            out[status == 2] = None
        return out

    def titles_at(self, i):
        off = np.ctypeslib.as_array(ctypes.cast(self._title_off, ctypes.POINTER(ctypes.c_uint64)),
                                    shape=(self.n_records + 1,))
        return ctypes.string_at(self._titles + int(off[i]), int(off[i + 1] - off[i])).decode("latin-1")

    def sequences(self):
        off = self.offsets()
        if self.total_bases and not self._bases:
            raise ValueError("this Fasta was opened with index_only=True: it holds no sequences")
        raw = ctypes.string_at(self._bases, self.total_bases) if self.total_bases else b""
        return [raw[int(off[i]):int(off[i + 1])].decode("latin-1") for i in range(self.n_records)]

    def count(self, ctx, kmer_length, symbols=b"ATGC"):
        out = np.zeros((self.n_records, 4 ** int(kmer_length)), dtype=np.int64)
        check(self.lib.phk_count_fasta(ctx.handle, self.handle, int(kmer_length), symbols, ptr(out)))
        return out

    def close(self, wait=True):
        """Frees the parsed file.  ``wait=False``: on a helper thread -- unmapping the sequence buffer of a multi-GB file
        takes tenths of a second (0.23 s for 5 GB) that the caller need not stand in."""
        h, self.handle = getattr(self, "handle", None), None
        if not h:
            return
        if wait:
            self.lib.phk_fasta_free(h)
        else:
            threading.Thread(target=self.lib.phk_fasta_free, args=(h,), name="phamers-fasta-free").start()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def get_context(device=None):
    """Process-wide default context for ``device`` (created on first use)."""
    dev = default_device() if device is None else int(device)
    with _lock:
        ctx = _contexts.get(dev)
    if ctx is None:
        ctx = Context(dev)
        with _lock:
            _contexts.setdefault(dev, ctx)
            ctx = _contexts[dev]
    return ctx
