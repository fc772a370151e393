#!/usr/bin/env python3
"""Characterise the internal arithmetic of v_mfma_f32_32x32x16_f16 through the library's probe entry
(phk_mfma_f16_probe): which low-order bits of a product survive beside a big term (alignment width), whether dropped
bits are truncated or rounded, how the 16 products and C are grouped.  Run on the GPU box; prints a report."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phamers_amd import _lib  # noqa: E402

ctx = _lib.get_context()


def run(A, B, C):
    A = np.ascontiguousarray(A, dtype=np.float16)[None, None]
    B = np.ascontiguousarray(B, dtype=np.float16)[None, None]
    C = np.ascontiguousarray(C, dtype=np.float32)[None]
    D = np.empty((1, 1, 32, 32), dtype=np.float32)
    _lib.check(ctx.lib.phk_mfma_f16_probe(ctx.handle, _lib.ptr(A.view(np.uint16)), _lib.ptr(B.view(np.uint16)), _lib.ptr(C), 1, 1,
                                          _lib.ptr(D)))
    return D[0, 0]


def elem(terms, c=0.0):
    """terms: list of 16 (a, b) for element (0, 0); returns the device result"""
    A = np.zeros((32, 16)); B = np.zeros((16, 32)); C = np.zeros((32, 32))
    for k, (a, b) in enumerate(terms):
        A[0, k] = a; B[k, 0] = b
    C[0, 0] = c
    return float(run(A, B, C)[0, 0])


Z = (0.0, 0.0)
print("== 1. one big product 2^24 at k = kb, one small product 2^-s at k = ks (C = -2^24): which s survive?")
for kb, ks in ((0, 1), (0, 7), (0, 8), (0, 15), (8, 9), (8, 0), (15, 14)):
    surv = []
    for s in range(0, 40):
        t = [Z] * 16
        t[kb] = (4096.0, 4096.0)
        t[ks] = (2.0 ** -(s // 2), 2.0 ** -(s - s // 2))
        got = elem(t, -2.0 ** 24)
        surv.append(got == 2.0 ** -s)
    print("  big k=%2d small k=%2d: 2^-s survives for s <= %d   (pattern %s)" % (kb, ks, max([i for i, v in enumerate(surv) if v] or [-1]),
                                                                              "".join("1" if v else "0" for v in surv)))
print("== 2. big term in C (2^24), product -2^24 at k=0, small product at k = ks")
for ks in (1, 7, 8, 15):
    surv = []
    for s in range(0, 40):
        t = [Z] * 16
        t[0] = (-4096.0, 4096.0)
        t[ks] = (2.0 ** -(s // 2), 2.0 ** -(s - s // 2))
        surv.append(elem(t, 2.0 ** 24) == 2.0 ** -s)
    print("  small k=%2d: survives for s <= %d" % (ks, max([i for i, v in enumerate(surv) if v] or [-1])))
print("== 3. rounding or truncation of what is dropped?  big 2^24 (k=0) + small x at k=1, C = 0: result vs exact")
for x in (0.5, 0.75, 1.0, 1.25, 1.5, 1.75, 2.5, 3.0, -0.5, -0.75, -1.0, -1.5, -1.75, -2.5, -3.0):
    t = [Z] * 16
    t[0] = (4096.0, 4096.0)
    t[1] = (x, 1.0)
    print("  2^24 + %5.2f -> %.1f   (RN-even of exact: %.1f)" % (x, elem(t, 0.0), float(np.float32(2.0 ** 24 + x))))
print("== 4. many small terms beside a big one: 2^24 (k=0) + m x 0.5 at k = 1..m (exact 2^24 + m/2)")
for m in (1, 2, 3, 4, 7, 8, 15):
    t = [Z] * 16
    t[0] = (4096.0, 4096.0)
    for k in range(1, m + 1):
        t[k] = (0.5, 1.0)
    print("  m=%2d -> %.1f  (exact %.1f)" % (m, elem(t, 0.0), 2.0 ** 24 + m / 2))
print("== 4b. same with the small terms in the other half (k = 8..)")
for m in (1, 2, 4, 8):
    t = [Z] * 16
    t[0] = (4096.0, 4096.0)
    for k in range(8, 8 + m):
        t[k] = (0.5, 1.0)
    print("  m=%2d -> %.1f  (exact %.1f)" % (m, elem(t, 0.0), 2.0 ** 24 + m / 2))
print("== 5. intermediate rounding between the halves?  k<8: 2^24 + 1 ; k>=8: -2^24   (exact 1)")
t = [Z] * 16
t[0] = (4096.0, 4096.0); t[1] = (1.0, 1.0); t[8] = (-4096.0, 4096.0)
print("  ->", elem(t, 0.0))
t = [Z] * 16
t[0] = (4096.0, 4096.0); t[1] = (1.0, 1.0); t[2] = (-4096.0, 4096.0)
print("  same inside one half (k=0,1,2) ->", elem(t, 0.0))
t = [Z] * 16
t[0] = (4096.0, 4096.0); t[1] = (2.0 ** -6, 2.0 ** -6); t[8] = (-4096.0, 4096.0)
print("  2^24 + 2^-12 (k<8), -2^24 (k=8) ->", elem(t, 0.0), "(exact", 2.0 ** -12, ")")
t = [Z] * 16
t[8] = (4096.0, 4096.0); t[9] = (2.0 ** -6, 2.0 ** -6); t[0] = (-4096.0, 4096.0)
print("  -2^24 (k=0), 2^24 + 2^-12 (k=8,9) ->", elem(t, 0.0))
print("== 6. worst case search: C = +-X, products random sign with magnitudes spread 2^0 .. 2^-30 below X")
rng = np.random.default_rng(0)
u = 2.0 ** -24
worst = (0, None)
for trial in range(3000):
    e = rng.integers(-26, 1, 16)
    sg = rng.choice([-1.0, 1.0], 16)
    terms = []
    for k in range(16):
        a = float(np.float16(rng.uniform(1.0, 2.0) * 2.0 ** (12 + e[k] // 2))) * sg[k]
        b = float(np.float16(rng.uniform(1.0, 2.0) * 2.0 ** (11 + (e[k] - e[k] // 2))))
        terms.append((a, b))
    c = float(np.float32(rng.choice([-1.0, 1.0]) * rng.uniform(1.0, 2.0) * 2.0 ** rng.integers(0, 25)))
    got = elem(terms, c)
    ex = math.fsum([a * b for a, b in terms] + [c])
    mag = sum(abs(a * b) for a, b in terms) + abs(c)
    r = abs(got - ex) / (u * mag)
    if r > worst[0]:
        worst = (r, (terms, c, got, ex))
print("  worst error / (u (|C| + sum|products|)) over 3000 trials: %.3f" % worst[0])
terms, c, got, ex = worst[1]
print("  C = %r  got %r exact %r" % (c, got, ex))
for k, (a, b) in enumerate(terms):
    print("   k=%2d  %r x %r = %r" % (k, a, b, a * b))
