"""
Seeded synthetic contigs: bases are i.i.d. uniform over
'ATGC' from a counter-based hash of (seed, contig, position), reproducible
bit-for-bit on the host (this file, NumPy) and on the device
(``phk_synth_packed_dev`` in csrc/synth.hip), so any contig of a device-resident
benchmark batch can be re-derived on the host for parity checks.

    key(c)   = splitmix64(splitmix64(seed) ^ c)
    word(c,j)= splitmix64(key(c) + j)                 # 32 bases per 64-bit word
    code(c,i)= (word(c, i >> 5) >> (62 - 2*(i & 31))) & 3      -> 'ATGC'[code]
    invalid(c,i) (only when invalid_ppm > 0):
               (splitmix64((key(c) ^ 0xA5A5A5A5A5A5A5A5) + i) >> 32) < floor(invalid_ppm * 2^32 / 1e6)
               -> the base is written as 'N' / its validity bit is cleared.
"""
import numpy as np

_U = np.uint64
_GOLD = _U(0x9E3779B97F4A7C15)
_M1 = _U(0xBF58476D1CE4E5B9)
_M2 = _U(0x94D049BB133111EB)
_INV_SALT = _U(0xA5A5A5A5A5A5A5A5)
_ATGC = np.frombuffer(b'ATGC', dtype=np.uint8)


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 (wrap-around arithmetic)."""
    with np.errstate(over='ignore'):
        z = np.asarray(x, dtype=_U) + _GOLD
        z = (z ^ (z >> _U(30))) * _M1
        z = (z ^ (z >> _U(27))) * _M2
        return z ^ (z >> _U(31))


def contig_key(seed, c):
    return splitmix64(splitmix64(_U(seed)) ^ np.asarray(c, dtype=_U))


def invalid_threshold(invalid_ppm):
    return int(invalid_ppm) * (1 << 32) // 1000000


def synth_codes(seed, c, L, invalid_ppm=0):
    """Codes (0..3 = A,T,G,C) of contig ``c``; -1 where the base is invalid."""
    L = int(L)
    key = contig_key(seed, c)
    i = np.arange(L, dtype=_U)
    with np.errstate(over='ignore'):
        words = splitmix64(key + np.arange((L + 31) // 32, dtype=_U))
    w = words[(i >> _U(5)).astype(np.int64)] if L else words[:0]
    codes = ((w >> (_U(62) - _U(2) * (i & _U(31)))) & _U(3)).astype(np.int8)
    if invalid_ppm:
        with np.errstate(over='ignore'):
            h = splitmix64((key ^ _INV_SALT) + i)
        codes[(h >> _U(32)) < _U(invalid_threshold(invalid_ppm))] = -1
    return codes


_GOLD2 = _U(0xD1B54A32D192ED03)


def synth_ragged_codes(seed, c, L, gc_spread_permille=400, invalid_ppm=0):
    """Codes of contig ``c`` of the ragged, composition-skewed batch: one hash per base, G/C with the contig's own
    probability, the hash's low bit picks within the pair; -1 where the base is invalid."""
    L = int(L)
    key = contig_key(seed, c)
    i = np.arange(L, dtype=_U)
    with np.errstate(over='ignore'):
        h = splitmix64(key + _GOLD2 + i)
    # the device computes the deviation with a signed division that truncates toward zero
    k48 = int(key) >> 48
    num = (k48 - 32768) * int(gc_spread_permille) * 65536
    dev = abs(num) // 1000 * (1 if num >= 0 else -1)
    thr = _U((2147483648 + dev) & 0xFFFFFFFF)
    codes = (np.where((h >> _U(32)) < thr, 2, 0) | (h & _U(1)).astype(np.int64)).astype(np.int8)
    if invalid_ppm:
        with np.errstate(over='ignore'):
            hv = splitmix64((key ^ _INV_SALT) + i)
        codes[(hv >> _U(32)) < _U(invalid_threshold(invalid_ppm))] = -1
    return codes


def synth_ragged_contig(seed, c, L, gc_spread_permille=400, invalid_ppm=0):
    return codes_to_str(synth_ragged_codes(seed, c, L, gc_spread_permille, invalid_ppm))


def ragged_lengths(seed, n, lo=5000, hi=500000, shape=1.1):
    """Heavy-tailed contig lengths in [lo, hi]: lo * Pareto(shape), clipped (NumPy generator seeded with `seed`)."""
    rng = np.random.default_rng(int(seed))
    return np.minimum(lo * (1.0 + rng.pareto(shape, int(n))), hi).astype(np.int64)


def codes_to_str(codes):
    out = _ATGC[np.where(codes < 0, 0, codes)].copy()
    out[codes < 0] = ord('N')
    return out.tobytes().decode('ascii')


def synth_contig(seed, c, L, invalid_ppm=0):
    return codes_to_str(synth_codes(seed, c, L, invalid_ppm))


def synth_contigs(seed, n, L, invalid_ppm=0, start=0):
    """``n`` contigs of length ``L`` (contig ids start..start+n-1)."""
    return [synth_contig(seed, start + c, L, invalid_ppm) for c in range(int(n))]


def contig_header(c, L):
    """FASTA header parsable by the reference's id rules
    (scripts/id_parser.py:18-30,95-96): id = str(c)."""
    return "SuperContig_%d_length_%d_ID_%d" % (c, L, c)
