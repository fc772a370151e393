"""Worker of tests/test_gpu_score.py::test_mfma_f16_rounding_charge_bulk_fuzz, run as a CHILD process
(`python -m tests.mfma_fuzz_worker`): the checker evaluates every result of the instruction in float64 on the device
with torch (test plumbing), and torch's HIP runtime has to be the first one initialised in its process -- the pytest
process has already opened the device through libphamers_hip.so.  Prints one JSON object: worst error / charge per family."""
import json
import os
import sys

import numpy as np
import torch

torch.cuda.init()          # before libphamers_hip.so is loaded (see above)
# The charge's p: the largest product by VALUE (round 3's statement of the charge, refuted by this fuzz for float16
# subnormal operands: argument "value" reproduces that) or the largest NOMINAL product (the charge as of round 4).
NOMINAL = not (len(sys.argv) > 1 and sys.argv[1] == "value")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def _mfma_probe(ctx, A, B, C):
    """A [T][S][32][16], B [T][S][16][32] float16, C [T][32][32] float32 -> D [T][S][32][32] float32 (phk_mfma_f16_probe)."""
    from phamers_amd import _lib
    A = np.ascontiguousarray(A, dtype=np.float16)
    B = np.ascontiguousarray(B, dtype=np.float16)
    C = np.ascontiguousarray(C, dtype=np.float32)
    T, S = A.shape[0], A.shape[1]
    D = np.empty((T, S, 32, 32), dtype=np.float32)
    _lib.check(ctx.lib.phk_mfma_f16_probe(ctx.handle, _lib.ptr(A.view(np.uint16)), _lib.ptr(B.view(np.uint16)), _lib.ptr(C),
                                          T, S, _lib.ptr(D)))
    return D


def _mfma_ratios_bulk(A, B, acc_in, D):
    """Vectorised _mfma_step_error_ratio for many instructions at once, on the device in float64 (test plumbing: torch):
    A [T][32][16], B [T][16][32] float16, acc_in / D [T][32][32] float32 -> the worst ratio per instruction [T].  A
    product of two float16 numbers is exact in float64; the float64 sums of 17 terms err by < 2^-48 of the largest term,
    2^-20 of the smallest possible charge u (11 A + 18 p) -- far inside the margin asserted."""
    import torch
    dev = torch.device("cuda", 0)
    u = 2.0 ** -24
    out = np.empty(A.shape[0])
    step = 8192
    for lo in range(0, A.shape[0], step):
        a = torch.from_numpy(np.ascontiguousarray(A[lo:lo + step]).view(np.int16)).to(dev).view(torch.float16).double()
        b = torch.from_numpy(np.ascontiguousarray(B[lo:lo + step]).view(np.int16)).to(dev).view(torch.float16).double()
        c = torch.from_numpy(np.ascontiguousarray(acc_in[lo:lo + step])).to(dev).double()
        d = torch.from_numpy(np.ascontiguousarray(D[lo:lo + step])).to(dev).double()
        prod = a[:, :, None, :] * b.transpose(1, 2)[:, None, :, :]            # [t][i][j][k] = A[i][k] B[k][j]
        half = c + prod[..., :8].sum(-1)
        exact = half + prod[..., 8:].sum(-1)
        amax = torch.maximum(torch.maximum(c.abs(), half.abs()), exact.abs())
        if NOMINAL:   # p = the largest NOMINAL product: a non-zero float16 subnormal counts as 2^-14 (its exponent field)
            an = torch.where(a != 0, a.abs().clamp_min(2.0 ** -14), a.abs())
            bn = torch.where(b != 0, b.abs().clamp_min(2.0 ** -14), b.abs())
            pmax = (an[:, :, None, :] * bn.transpose(1, 2)[:, None, :, :]).amax(-1)
        else:
            pmax = prod.abs().amax(-1)
        charge = u * (11.0 * amax + 18.0 * pmax)
        err = (d - exact).abs()
        assert bool(((charge > 0) | (err == 0)).all())
        ratio = torch.where(charge > 0, err / charge.clamp_min(1e-300), torch.zeros_like(err))
        out[lo:lo + step] = ratio.amax(dim=(1, 2)).cpu().numpy()
        del prod
    return out


def _rand_f16(rng, shape, emin, emax, exps=None):
    """Random float16 values with exponents drawn uniformly from [emin, emax] (per element, or the given array), a random
    11-bit significand and a random sign; exponents below -14 give float16 subnormals (fewer significant bits)."""
    e = rng.integers(emin, emax + 1, shape) if exps is None else exps
    m = rng.integers(1024, 2048, shape).astype(np.float64) / 1024.0
    v = np.ldexp(m, e) * rng.choice([-1.0, 1.0], shape)
    return v.astype(np.float16)


def main():
    """Bulk random fuzz of the per-instruction charge u (11 A + 18 p) of v_mfma_f32_32x32x16_f16 (the
    structured families of the test above are the author's idea of a worst case -- this one is not): >= 10^5
    instructions per family through phk_mfma_f16_probe, every one of its 1024 results checked against float64.
    Families: (a) exponents of A, B and C drawn independently per element over the float16 / float32 ranges, random
    signs; (b) per-tile exponent windows of width 0..6 with C at the scale of the products -- terms of nearly equal
    magnitude, where the cuts bite most; (c) float16 subnormals mixed into B beside integer A (the low parts of split
    columns beside counts); (d) chains of 16 instructions (k = 4 kernels); (e) chains of 256 (the D = 4096 f16 kernel),
    each instruction checked against the accumulator the device really fed it."""
    import torch
    from phamers_amd import _lib
    ctx = _lib.get_context()
    # (a one-off larger run for the record: PHK_FUZZ_LOG2N=20 PHK_FUZZ_SEED=... python -m tests.mfma_fuzz_worker)
    rng = np.random.default_rng(int(os.environ.get("PHK_FUZZ_SEED", "20261005")))
    n = 1 << int(os.environ.get("PHK_FUZZ_LOG2N", "17"))   # 131 072 instructions per family
    worst = {}

    def run_single(name, gen, chunk=16384):
        w = 0.0
        for lo in range(0, n, chunk):
            A, B, C = gen(chunk)
            D = _mfma_probe(ctx, A[:, None], B[:, None], C)[:, 0]
            assert np.isfinite(D).all()
            r = _mfma_ratios_bulk(A, B, C, D)
            w = max(w, float(r.max()))
        worst[name] = w

    def fam_a(t):
        A = _rand_f16(rng, (t, 32, 16), -24, 15)
        B = _rand_f16(rng, (t, 16, 32), -24, 15)
        C = (np.ldexp(rng.uniform(1.0, 2.0, (t, 32, 32)), rng.integers(-50, 41, (t, 32, 32)))
             * rng.choice([-1.0, 1.0, 0.0], (t, 32, 32), p=[0.45, 0.45, 0.1])).astype(np.float32)
        return A, B, C

    def fam_b(t):
        w = rng.integers(0, 7, (t, 1, 1))
        ea = rng.integers(-12, 10, (t, 1, 1))
        eb = rng.integers(-12, 10, (t, 1, 1))
        A = _rand_f16(rng, (t, 32, 16), 0, 0, exps=ea + rng.integers(0, 7, (t, 32, 16)) % (w + 1))
        B = _rand_f16(rng, (t, 16, 32), 0, 0, exps=eb + rng.integers(0, 7, (t, 16, 32)) % (w + 1))
        ec = ea + eb + rng.integers(-6, 11, (t, 32, 32))
        C = (np.ldexp(rng.uniform(1.0, 2.0, (t, 32, 32)), ec)
             * rng.choice([-1.0, 1.0, 0.0], (t, 32, 32), p=[0.4, 0.4, 0.2])).astype(np.float32)
        return A, B, C

    def fam_c(t):
        A = (rng.integers(0, 2049, (t, 32, 16)) * rng.choice([-1.0, 1.0], (t, 32, 16))).astype(np.float16)
        sub = _rand_f16(rng, (t, 16, 32), -24, -15)
        nor = _rand_f16(rng, (t, 16, 32), -14, 10)
        B = np.where(rng.random((t, 16, 32)) < rng.uniform(0.0, 1.0, (t, 1, 1)), sub, nor)
        C = (rng.standard_normal((t, 32, 32)) * np.ldexp(1.0, rng.integers(-30, 25, (t, 1, 1)))).astype(np.float32)
        return A, B, C

    run_single("independent exponents", fam_a)
    run_single("exponent windows", fam_b)
    run_single("subnormals", fam_c)

    def run_chains(name, S, gen):
        w, wc = 0.0, 0.0
        u = 2.0 ** -24
        chains = n // S
        per = max(1, 4096 // S)
        for lo in range(0, chains, per):
            t = min(per, chains - lo)
            A, B = gen(t, S)                       # [t][S][32][16], [t][S][16][32]
            C = np.zeros((t, 32, 32), dtype=np.float32)
            D = _mfma_probe(ctx, A, B, C)          # [t][S][32][32]
            assert np.isfinite(D).all()
            acc_in = np.concatenate((C[:, None], D[:, :-1]), axis=1)
            r = _mfma_ratios_bulk(A.reshape(t * S, 32, 16), B.reshape(t * S, 16, 32), acc_in.reshape(t * S, 32, 32),
                                  D.reshape(t * S, 32, 32))
            w = max(w, float(r.max()))
            # the whole chain against what the error model sums up: n u (11 |x| |y| + 18 |x|_inf |y|_inf)
            x = torch.from_numpy(A.view(np.int16)).cuda().view(torch.float16).double().permute(0, 2, 1, 3).reshape(t, 32, S * 16)
            y = torch.from_numpy(B.view(np.int16)).cuda().view(torch.float16).double().reshape(t, S * 16, 32)
            exact = x @ y
            bound = S * u * (11.0 * x.norm(dim=2)[:, :, None] * y.norm(dim=1)[:, None, :]
                             + 18.0 * x.abs().amax(2)[:, :, None] * y.abs().amax(1)[:, None, :])
            err = (torch.from_numpy(D[:, -1]).cuda().double() - exact).abs()
            slack = 1e-12 * (x.abs() @ y.abs())    # float64 rounding of the reference product itself
            wc = max(wc, float(((err - slack).clamp_min(0) / bound.clamp_min(1e-300)).max()))
        worst[name] = w
        worst[name + ", whole chain vs the model's sum"] = wc

    def fam_chain(t, S):
        kind = rng.integers(0, 3, (t, 1, 1, 1))
        counts = rng.poisson(rng.uniform(2.0, 40.0, (t, 1, 1, 1)), (t, S, 32, 16)).astype(np.float64)
        counts = counts - np.rint(counts.mean(axis=(1, 3), keepdims=True))               # centred rows
        wide = rng.integers(-2048, 2049, (t, S, 32, 16)).astype(np.float64)
        spike = np.where(rng.random((t, S, 32, 16)) < 0.01, 2048.0, counts)
        A = np.where(kind == 0, counts, np.where(kind == 1, spike, wide)).astype(np.float16)
        B = (rng.standard_normal((t, S, 16, 32)) * rng.uniform(1.0, 700.0, (t, 1, 1, 1))).astype(np.float16)
        pair = rng.random((t, 1, 1, 1)) < 0.3                                             # cancelling column pairs
        Bp = B.copy()
        Bp[:, :, 1::2] = -B[:, :, 0::2] * np.float16(1.0 + 2.0 ** -9)
        Ap = A.copy()
        Ap[..., 1::2] = A[..., 0::2]
        return np.where(pair, Ap, A), np.where(pair, Bp, B)

    run_chains("chains of 16", 16, fam_chain)
    run_chains("chains of 256", 256, fam_chain)
    print(json.dumps({"instructions_per_family": n, "p": "nominal" if NOMINAL else "value", "worst_error_over_charge": worst}))


if __name__ == "__main__":
    main()
