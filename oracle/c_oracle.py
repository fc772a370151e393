"""ctypes binding of oracle/kmer_oracle.c (TEST INFRASTRUCTURE ONLY -- see the
header of oracle.py for who may import this)."""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libphk_oracle.so')
_lib = None

_u64p = ctypes.POINTER(ctypes.c_uint64)
_i64p = ctypes.POINTER(ctypes.c_int64)
_f64p = ctypes.POINTER(ctypes.c_double)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def count(seqs, k, symbols='ATGC'):
    if isinstance(seqs, str):
        seqs = [seqs]
    raw = ''.join(seqs).encode('latin-1', 'replace')
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    D = len(symbols) ** k
    out = np.zeros((len(seqs), D), dtype=np.int64)
    rc = lib().oracle_count(ctypes.c_char_p(raw), _p(off, _u64p), ctypes.c_uint64(len(seqs)),
                            ctypes.c_int(k), ctypes.c_char_p(symbols.encode()),
                            ctypes.c_int(len(symbols)), _p(out, _i64p))
    assert rc == 0
    return out


def normalize(counts):
    c = np.ascontiguousarray(counts, dtype=np.int64)
    c2 = c.reshape(-1, c.shape[-1])
    out = np.empty(c2.shape, dtype=np.float64)
    lib().oracle_normalize(_p(c2, _i64p), ctypes.c_uint64(c2.shape[0]), ctypes.c_uint64(c2.shape[1]),
                           _p(out, _f64p))
    return out.reshape(c.shape)


def knn_score(Q, R, labels, kn=3):
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    R = np.ascontiguousarray(R, dtype=np.float64)
    lab = np.ascontiguousarray(labels, dtype=np.uint8)
    out = np.empty(Q.shape[0], dtype=np.float64)
    rc = lib().oracle_knn_score(_p(Q, _f64p), ctypes.c_uint64(Q.shape[0]), _p(R, _f64p),
                                ctypes.c_uint64(R.shape[0]), _p(lab, _u8p),
                                ctypes.c_uint64(Q.shape[1]), ctypes.c_int(kn), _p(out, _f64p))
    assert rc == 0
    return out


def centroid_score(Q, Cpos, Cneg):
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    Cp = np.ascontiguousarray(Cpos, dtype=np.float64)
    Cn = np.ascontiguousarray(Cneg, dtype=np.float64)
    out = np.empty(Q.shape[0], dtype=np.float64)
    lib().oracle_centroid_score(_p(Q, _f64p), ctypes.c_uint64(Q.shape[0]), _p(Cp, _f64p),
                                ctypes.c_uint64(Cp.shape[0]), _p(Cn, _f64p),
                                ctypes.c_uint64(Cn.shape[0]), ctypes.c_uint64(Q.shape[1]),
                                _p(out, _f64p))
    return out
