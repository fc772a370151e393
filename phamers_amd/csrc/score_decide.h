// score_decide.h -- what the decision stage's translation units share (round 5: score_mfma.hip, 3200 lines, split by stage):
// the parameter blocks the driver fills (RerankParams, HiParams), the error model (ErrBound, certify_segments), the
// canonical float64 distance forms (exact_d2, exact_d2_g16) with the wave / 16-lane-group reductions under them, the
// brute force's record, and one launcher per kernel family.
//     score_rerank.hip    phk_rerank_kernel (one wave per query), phk_rerank16_kernel (four queries per wave, D = 256),
//                         phk_rerank_h_kernel (high-part lists at general D)
//     score_decide.hip    phk_decide_kernel, phk_decide_gen_kernel, phk_decide_h_kernel (one lane per query)
//     score_fallback.hip  the exact float64 brute force for queued queries + the small queue / gather kernels
//     score_mfma.hip      model build and the driver phk_score_fast: it fills the parameter blocks and calls the launchers
// Device code only (every function here is a template, __forceinline__ or static): include from .hip files.
#pragma once
#include "phk_common.h"
#include "score_model.h"
#include "score_lists.h"

#ifndef PHK_HI_REFINE
#define PHK_HI_REFINE 6  // candidates per query whose low product the high-parts-only decision stage evaluates
#endif
#define FB_CHUNKS 16   // column chunks per queued query in the exact brute-force fallback
#define FB_LDS_MAX (160u * 1024u - 1024u)   // its dynamic LDS: one chunk of distances + the query, float64

// ------------------------------------------------------------------------------------
// 2. certify / exact re-rank: one wavefront per query, lane = 4 dimensions
// ------------------------------------------------------------------------------------
#define PHK_STRIPES 256
#define PHK_SUBPASS_MIN 32 // general D: a hand-over queue shorter than this is brute-forced, not swept (phk_score_fast)
#define PHK_SUB_LISTS 16   // pairs of hand-over lists between phk_decide_h_kernel and phk_rerank16_kernel (<= PHK_STRIPES)
struct RerankParams {
    uint64_t N, M, n_cpos, n_cneg, D;
    int kn, method;
    double rmax, mu_norm;
    double vscale;          // computed values are in units of 1/vscale (split-f16 path: S^2)
    double eb_cA, eb_cP, eb_cR, eb_abs;  // error model of the proposal pass (see ErrBound)
    double eb_cQ;           // coefficient of Q (count-exact proposal: |c - c0| / T, the centred count operand)
    double eb_cI = 0.0;     // coefficient of I, the maximum norm of the query operand (see ErrBound)
    double eb_cIf = 0.0;    // the same without rho_inf: the floor term of the nominal products (float16 subnormals, see ErrBound)
    double eb_hsum = 0.0;   // max_j |sum_i r~'_ji| (count-exact proposals: the residue of centring the counts)
    int per_row_scale;      // count-exact proposal: computed values are in units of T_q / vscale (T_q = row sum)
    const double *R64, *C64, *mu64, *colnorm;
    const uint8_t *labels;
    const float *cand_v;      // candidate lists, structure of arrays (score_lists.h: cand_at / candu_at)
    const uint32_t *cand_i;
    const float *cand_u;
    uint64_t fb_rec_cap = 0;         // general D: records the brute-force workspace holds (phk_fallback_group_kernel cuts the
                                     // reference into 64 chunks per query when they fit, else FB_CHUNKS); 0: FB_CHUNKS
    const float *cand_a = nullptr;   // general D: largest |accumulator| a lane saw at the (block, chunk) item boundaries of its
                                     // sweep, [2 halves][N] (the running sums the chain's charges scale with; see ErrBound)
    double *pend = nullptr;          // general-D decision kernel: [2 N] exact d^2 to the nearest positive / negative centroid of the
                                     // queries it decided; phk_finish_cen_kernel turns them into the proximity metric
    double eb_cAmax = 0.0;           // coefficient of that observed running sum (PHK_MFMA_ACC x instructions per value)
    void *fb_rec;           // fallback partial records
    double *scores;
    uint32_t *status;       // NaN-row counter (may be null)
    uint32_t *fb_count;     // fallback queue length
    uint32_t *fb_list;      // fallback queue (query indices)
    uint32_t *slow_list;    // queries the one-lane-per-query decision kernel could not certify (fb_count[2] of them)
    uint64_t q_base;        // index of this batch's first query within the caller's arrays
    const uint8_t *col_mask = nullptr;   // train columns excluded from the search (cross-validation folds), or null
    // "second chance" pass (phk_rerank16_kernel MODE 2): the queries are rows map[0 .. *map_count) of the batch, their
    // candidate lists sit at the dense positions 0 .. min(*map_count, N) of a second list set of capacity N
    const uint32_t *map = nullptr;
    const uint32_t *map_count = nullptr;
    uint32_t *counters = nullptr;           // the batch's counter words (see phk_score_fast)
    // Statistics that most waves of a large grid increment -- decisions by exact distances, the reasons a query is handed on --
    // are counted in PHK_STRIPES copies of the counter words, each in a cache line of its own, chosen by the workgroup number,
    // and summed by phk_fallback_merge_kernel.  As atomics on the batch's ONE line of counters they were a serial resource the
    // whole grid queued for: ~9 ns apiece, 0.31 ms of phk_decide_h_kernel's 0.79 and 0.19 ms of phk_rerank16_kernel's 0.35 on
    // configs[1] (end of round 3; profiles/r03/README.md).
    uint32_t *stripes = nullptr;            // [PHK_STRIPES][32] words, or null: count in `counters` / `fb_count`
    int slow_back = 0;                      // phk_rerank16_kernel MODE 1: 3 = both of the following in one launch; 0 = slow_list[0 ..) counted by fb_count[2],
                                            // 1 = the list that grows down from slow_list[slow_cap - 1], counted by counters[12]
    uint64_t slow_cap = 0;
    // phk_decide_h_kernel -> phk_rerank16_kernel (MODE 1, slow_back == 3): the hand-over lists as `sub_lists` separate pairs of
    // lists, workgroup b of the decision kernel appending to pair b % sub_lists.  Pair s owns slow_list[s sub_cap, (s + 1) sub_cap)
    // (front list up from its start, back list down from its end; sub_cap = 64 ceil(workgroups / sub_lists) bounds what its
    // workgroups can hand over) and counts in words 2 / 12 of stripe s -- a cache line of its own, where the two returning
    // atomics per wave of the decision kernel no longer queue behind every other wave's (0: the single pair of lists above)
    uint32_t sub_lists = 0;
    uint64_t sub_cap = 0;
    uint32_t *stat_total = nullptr;         // [0] += fallback queue length, [1] += orderings decided by exact distances
    const uint32_t *exact_extra = nullptr;  // exact-distance decisions of an earlier pass of the same batch
    // lists of the two-part int8 sweep (score_i8.hip; phk_rerank_kernel<.., I8H>): a value lacks g_j S_L, S_L = the exact
    // integer product of c - c0 with the column's L digits -- |.| <= |c - c0| lam8[segment] per unit of row sum
    // Per-row routing at general D (phk_score_fast): a pass over a SUB-BATCH -- rows of the batch gathered into a dense
    // count matrix -- works on dense indices; out_map[i] is row i's index within the batch, used wherever a result
    // leaves the pass: the score, the centroid distances in `pend`, the brute-force queue.  In the first pass (q2_count
    // set) a row the lists cannot decide is handed on instead of brute-forced: a row beyond the int8 operand (sentinel
    // lists) to q2_big, counted by q2_count[1] -- the f16 count-exact sweep takes it -- any other to q2_wide, counted by
    // q2_count[0] -- re-swept with all three digits (null: straight to the brute-force queue).
    const uint32_t *out_map = nullptr;
    const uint32_t *rowsum = nullptr;       // row sums of the count rows (phk_decide_gen_kernel), or null
    uint32_t *q2_count = nullptr;
    uint32_t *q2_wide = nullptr;
    uint32_t *q2_big = nullptr;
    // phk_fallback_merge_kernel, the last kernel of a batch: its last workgroup to finish zeroes the batch's counter words
    // (word 15 = the ticket) and striped statistics words, so that no memset precedes the next use of the set
    uint32_t *clean_counters = nullptr;
    uint32_t *clean_stripes = nullptr;
    double eb_babs = 0.0;     // k = 4 high-parts-only lists: what the bias as the sweep's three float16 pieces can be off by, in v units (added to habs)
    double eb_cM = 0.0, eb_M = 0.0;   // ... and the rounding of the bias step: coefficient of |mu - 1/D| (ErrBound::cM, M)
    const int8_t *L8 = nullptr;             // [M + n_cpos + n_cneg][D] L digits, row-major
    const float *T8 = nullptr;              // per 32-column block: 32 quanta g_j (+ 32 bias terms)
    uint32_t t8_blk[3] = {0, 0, 0};         // first block of each segment
    double lam8[3] = {0, 0, 0};
};

// the counter word `k` of this workgroup's stripe (see RerankParams::stripes)
__device__ __forceinline__ uint32_t *phk_stat_word(const RerankParams &p, uint32_t *plain, int k) {
    return p.stripes ? p.stripes + (blockIdx.x & (PHK_STRIPES - 1)) * 32u + (uint32_t)k : plain;
}

// Sums inside each group of 16 lanes = one DPP row: rotations by 8, 4, 2, 1 (row_ror) leave the total on every lane, in
// the VALU (a __shfl_xor butterfly is 4 dependent ds_bpermute round trips per sum -- with eleven sums per pass that chain
// was most of phk_decide_h_kernel's time).  Same operand pairs as the xor butterfly, so the same bits.
template <int ROR>
__device__ __forceinline__ int row_ror_i32(int v) {
    return __builtin_amdgcn_mov_dpp(v, 0x120 + ROR, 0xF, 0xF, true);
}
template <int ROR>
__device__ __forceinline__ double row_ror_f64(double x) {
    return __hiloint2double(row_ror_i32<ROR>(__double2hiint(x)), row_ror_i32<ROR>(__double2loint(x)));
}
__device__ __forceinline__ double group16_sum(double x) {
    x += row_ror_f64<8>(x);
    x += row_ror_f64<4>(x);
    x += row_ror_f64<2>(x);
    x += row_ror_f64<1>(x);
    return x;
}
__device__ __forceinline__ float group16_sum(float x) {
    x += __int_as_float(row_ror_i32<8>(__float_as_int(x)));
    x += __int_as_float(row_ror_i32<4>(__float_as_int(x)));
    x += __int_as_float(row_ror_i32<2>(__float_as_int(x)));
    x += __int_as_float(row_ror_i32<1>(__float_as_int(x)));
    return x;
}
__device__ __forceinline__ uint32_t group16_sum(uint32_t x) {
    x += (uint32_t)row_ror_i32<8>((int)x);
    x += (uint32_t)row_ror_i32<4>((int)x);
    x += (uint32_t)row_ror_i32<2>((int)x);
    x += (uint32_t)row_ror_i32<1>((int)x);
    return x;
}

__device__ __forceinline__ uint32_t group16_max(uint32_t x) {
    x = max(x, (uint32_t)row_ror_i32<8>((int)x));
    x = max(x, (uint32_t)row_ror_i32<4>((int)x));
    x = max(x, (uint32_t)row_ror_i32<2>((int)x));
    x = max(x, (uint32_t)row_ror_i32<1>((int)x));
    return x;
}
__device__ __forceinline__ uint32_t group16_min(uint32_t x) {
    x = min(x, (uint32_t)row_ror_i32<8>((int)x));
    x = min(x, (uint32_t)row_ror_i32<4>((int)x));
    x = min(x, (uint32_t)row_ror_i32<2>((int)x));
    x = min(x, (uint32_t)row_ror_i32<1>((int)x));
    return x;
}
__device__ __forceinline__ double group16_max(double x) {
    x = fmax(x, row_ror_f64<8>(x));
    x = fmax(x, row_ror_f64<4>(x));
    x = fmax(x, row_ror_f64<2>(x));
    x = fmax(x, row_ror_f64<1>(x));
    return x;
}
__device__ __forceinline__ double wave_max(double x) {
    x = group16_max(x);
    x = fmax(x, __shfl_xor(x, 16));
    x = fmax(x, __shfl_xor(x, 32));
    return x;
}

// The query operand of a count-exact MFMA chain: the counts minus their centre c0 = phk_row_center(T, D).  From the row's
// sum of squares, sum, largest and smallest count: Q = |c - c0| / T, I = |c - c0|_inf / T, and the residue of the centring
// habs = |c0 - T/D| hsum / T (see ErrBound); also |q' - (c0/T - 1/D) 1|^2 = |q'|^2 + D (c0/T - 1/D)^2, the operand the
// low parts of a high-parts-only value multiply (sum_i q'_i = 0).
struct CenteredOperand {
    double Q, I, habs, shift2;   // shift2 = D (c0 / T - 1 / D)^2
};
__device__ __forceinline__ CenteredOperand phk_centered_operand(double sumsq, double T, double cmax, double cmin, double D,
                                                                double hsum) {
    const double c0 = (double)phk_row_center((uint32_t)T, (uint32_t)D);
    CenteredOperand o;
    const double ss = fmax(sumsq - 2.0 * c0 * T + D * c0 * c0, 0.0);
    o.Q = sqrt(ss) / T * (1.0 + 1e-12);
    o.I = fmax(cmax - c0, c0 - cmin) / T;
    const double dl = c0 / T - 1.0 / D;
    o.habs = fabs(dl) * hsum;
    o.shift2 = D * dl * dl;
    return o;
}

// The same from the row's reciprocal sum rT = RN(1 / T), with float32 square root: every output is an upper bound with
// slack, at a fifth of the instructions (three float64 divisions and a float64 square root otherwise) -- for the
// wave-per-query decision kernel, whose per-query scalar arithmetic is what bounds it.
__device__ __forceinline__ CenteredOperand phk_centered_operand_fast(double sumsq, double T, double rT, double cmax, double cmin,
                                                                     double D, double hsum) {
    const double c0 = (double)phk_row_center((uint32_t)T, (uint32_t)D);
    CenteredOperand o;
    const double ss = fmax(sumsq - 2.0 * c0 * T + D * c0 * c0, 0.0);
    o.Q = (double)__builtin_sqrtf((float)ss) * rT * (1.0 + 1.0e-6);
    o.I = fmax(cmax - c0, c0 - cmin) * rT * (1.0 + 1.0e-9);
    const double dl = fabs(c0 * rT - 1.0 / D) * (1.0 + 1.0e-9) + 1.0e-18;
    o.habs = dl * hsum;
    o.shift2 = D * dl * dl;
    return o;
}

// Exact squared distance of the group's query to `row`.  The query is held UNNORMALISED: qd = the integer
// counts (or the float64 row with Tq = 1), Tq = their sum, and
//     |q - r|^2 = sum_i (c_i - Tq r_i)^2 / Tq^2        (one rounding per difference, inside the fma)
// which needs no per-element division and is at least as accurate as forming q = c / Tq first.
__device__ __forceinline__ uint32_t wave_sum(uint32_t x) {
    x = group16_sum(x);
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}
__device__ __forceinline__ double wave_sum(double x) {   // rows in the VALU, the four row totals through two bpermute steps
    x = group16_sum(x);
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}

// exact direct-difference squared distance of the wave's query to `row`.  D = 256 * DSUB; lane l holds dimensions
// 256*sub + 4l .. +3 of the query for sub = 0 .. DSUB-1.  The query is held UNNORMALISED, as at k = 4 (exact_d2_g16):
// qd = the integer counts with Tq = their sum and invT2 = 1 / Tq^2 (float64 rows: qd = the row, Tq = invT2 = 1, and the
// expression below is q_i - r_i exactly), |q - r|^2 = sum_i (c_i - Tq r_i)^2 / Tq^2 -- one rounding per difference, inside
// the fma, and no per-element division: forming q = c / Tq first cost the decision kernel 5 multiply-adds per element
// (round 4; every route of a query -- decision kernel, exact candidate distances, brute force -- uses this one form).
template <int DSUB>
__device__ __forceinline__ double exact_d2(const double (&qd)[4 * DSUB], double Tq, double invT2, const double *row, int lane) {
    double acc = 0.0;
#pragma unroll
    for (int sub = 0; sub < DSUB; ++sub) {
        const double2 a = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane];
        const double2 b = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane + 1];
        const double d0 = fma(-Tq, a.x, qd[4 * sub + 0]), d1 = fma(-Tq, a.y, qd[4 * sub + 1]);
        const double d2 = fma(-Tq, b.x, qd[4 * sub + 2]), d3 = fma(-Tq, b.y, qd[4 * sub + 3]);
        acc = fma(d0, d0, fma(d1, d1, fma(d2, d2, fma(d3, d3, acc))));
    }
    return wave_sum(acc) * invT2;
}

// Bound on |computed v - true v| of the proposal pass for a column with |r'| <= R.  u = 2^-24;
// A = |q| + |mu|; P = |q'|; Q, I: Euclidean and maximum norm of the query operand as the MFMA chain sees it:
//     eps(R) = u R (cA A + cQ Q + cI I + cP P + cR R) + c_abs (R + P) + habs
//   fp32 MFMA  : cA 6, cP 264, cR 4 -- q' to fp32 (<= 4uA), r' to fp32 (<= uR), 258 fused roundings each <= u |partial|
//                (the instruction is a k-ordered fmaf chain), product partials <= (P + 4uA) R, the norm step LAST.
//   f16 MFMA   : a chain of n instructions on operands x (query side) and y (column side) errs by at most
//                n u (PHK_MFMA_ACC |x| |y| + PHK_MFMA_PROD |x|_inf |y|_inf)  (score_lists.h: two halves of aligned,
//                truncated terms and one rounding each; running sums <= |x| |y| by Cauchy-Schwarz).
//     split f16   (n = 3D/16; x = q' S as hi + lo, y = r' S as hi + lo): cA 6, cP 11 n + 24, cI 18 n on I = |q'|_inf,
//                cR 6, c_abs sqrt(D) 2^-24/S -- operands carry 22 bits (|x - hi - lo| <= 2^-22 |x| + one fp16 subnormal
//                quantum), the dropped lo.lo term <= 2^-22 P R.
//     count-exact (n = 2D/16; high parts only: n = D/16; x = the integer counts minus their centre c0 -- exact in fp16 --
//                y = r~' S): cA 1 (bias -> fp32), cQ 11 n + 3 on Q = |c - c0| / T, cI 18 n on I = |c - c0|_inf / T,
//                cP 4 + 1 + 62, cR 4 + 1 + 31 (r' -> r~': 2^-22; the final fma: u |v|; the 5 index bits embedded in the
//                value at k = 4: 31 ulp <= 62 u |v|), c_abs as split f16, habs = |c0 - T/D| max_j |sum_i r~'_ji| / T
//                (what centring the counts leaves behind; the column sums of r' vanish up to the split's rounding).
//     D > 256    the sweep runs in chunks of 256 dimensions and the kernel records the largest |accumulator| a lane met at
//                the chunk boundaries (cand_a): inside a chunk a running sum is within |x_c| |y_c| of the sum at its start,
//                so Q is the largest CHUNK norm of the query operand and habs gains cAmax u a_observed -- the bound follows
//                the sums that occurred instead of the Cauchy-Schwarz worst case over all D dimensions.
//     int8       (score_i8.hip; x = c - c0 as int8, y = r' in 24-bit fixed point, three int8 parts): the part sums are exact
//                integers, so no chain term: cQ 4 on the full |c - c0| / T (three int -> float conversions and two fused
//                multiply-adds on sums <= 1.26 / 0.26 / 0.27 / 1.0 |x| |y|: |x|_1 <= sqrt(D) |x|, 2^15 g <= |y|_inf / 253),
//                cA 2 (bias -> fp32, T b), cP kappa/u + 2, cR kappa (1 + kappa)/u + 3 with kappa = max_j |r'_j - r~'_j| / |r'_j| of
//                the quantisation (computed at build; the final fma: u |v|), c_abs 0, habs as count-exact.
//     cI carries rho_inf = max_j |r~'_j|_inf / |r'_j| of the model, so that |x|_inf |y|_inf <= I rho_inf R.
//     Subnormal float16 operands (round 4; found by the bulk fuzz of tests/mfma_fuzz_worker.py, confirmed by the probe in
//     tools/diag/mfma_emulate.py): the instruction aligns a term by the operands' exponent FIELDS, so a non-zero subnormal
//     (|x| < 2^-14: the low parts of small reference elements) counts as 2^-14 whatever its leading zeros, and the `p` of
//     the per-instruction charge u (11 A + 18 p) is the largest NOMINAL product:  p <= (|x|_inf + 2^-14)(|y|_inf + 2^-14).
//     In v units, with f = 2^-14 / S:  p <= (I + f)(rho_inf R + f), i.e. the chain is charged cIf u f (I + R + f) on top of
//     the cI term (cIf = 18 n) -- six orders of magnitude below it for any real reference (f = 1.5e-8 against R ~ 1e-3), but
//     without it the bound is not a bound.
struct ErrBound {
    double A, P, cA, cP, cR, cabs;
    double Q = 0.0, cQ = 0.0;
    double I = 0.0, cI = 0.0;
    double habs = 0.0;
    double cIf = 0.0;   // f16 chains: PHK_MFMA_PROD x instructions, WITHOUT rho_inf -- the floor of the nominal products, see above
    double M = 0.0, cM = 0.0;   // |mu - 1/D| and its coefficient: the bias step of the k = 4 sweep (round 5; see phk_score_fast)
    __device__ double operator()(double R) const {
        const double f = 1.4901161193847656e-08;   // 2^-14 / S, S = 2^12: a float16 subnormal's nominal magnitude in operand units
        return 5.9604644775390625e-08 * (R * (cA * A + cQ * Q + cI * I + cP * P + cR * R + cM * M) + cIf * f * (I + R + f)) + cabs * (R + P) + habs;
    }
};

// An upper bound of sqrt(x) from the float32 instruction (1 instruction, |error| < 2e-7 relative with the conversion) --
// for norms and radii that only enter the error bound or the triangle radius, where larger is the safe side; a float64
// square root is ~30 instructions of this kernel's budget.
__device__ __forceinline__ double phk_sqrt_up(double x) { return (double)__builtin_sqrtf((float)x) * (1.0 + 1.0e-6); }

// The margin test of resolve_segment for the three segments AT ONCE: lanes 8 g .. 8 g + 7 hold segment g's candidates and
// carry out its ranking, its triangle radius and its margin test side by side (the conditions are resolve_segment's, word
// for word; what differs per segment -- columns, `need`, the norms' base -- is per-lane data).  One after the other the
// three tests were two thirds of the decision kernel's per-query instructions, every lane of the wave computing the same
// scalars.  Returns bit 8 g set where segment g is certified; ri[0..2] = the lane's segment's best columns.
// A segment that is not certified goes through resolve_segment (exact candidate distances) as before.
__device__ __forceinline__ uint64_t certify_segments(const RerankParams &p, double nqp2, double nqp, const ErrBound &eb, double vs,
                                                     int lane, float pre_v, uint32_t pre_i, uint32_t (&ri)[3]) {
    const int g = lane >> 3, gb = lane & 56;
    const uint32_t ncols = g == 0 ? (uint32_t)p.M : g == 1 ? (uint32_t)p.n_cpos : g == 2 ? (uint32_t)p.n_cneg : 0u;
    const int need = g == 0 ? p.kn : 1;
    const double *colnorm = p.colnorm + (g == 0 ? 0 : g == 1 ? p.M : p.M + p.n_cpos);
    float v = pre_v;
    const uint32_t ix = pre_i;
    if (ix >= ncols) v = -3.0e38f;  // padding / empty slot (and lanes >= 24)
    int rank = 0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const float ov = __shfl(v, gb | m);
        rank += (ov > v || (ov == v && m < (lane & 7))) ? 1 : 0;
    }
    float rv[4];
    uint32_t rx[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned long long bal = __ballot(rank == r);
        const int srcl = gb + __ffsll((long long)((bal >> gb) & 0xFFull)) - 1;
        rv[r] = __shfl(v, srcl);
        rx[r] = __shfl(ix, srcl);
    }
    ri[0] = rx[0]; ri[1] = rx[1]; ri[2] = rx[2];
    const double eps_g = eb(p.rmax);  // holds for every column
    const float rvn1 = need == 1 ? rv[0] : need == 2 ? rv[1] : rv[2], rvn = need == 1 ? rv[1] : need == 2 ? rv[2] : rv[3];
    const uint32_t rxn1 = need == 1 ? rx[0] : need == 2 ? rx[1] : rx[2];
    bool cert = false;
    if (rxn1 < ncols) {
        const double d2up = fmax(nqp2 - 2.0 * ((double)rvn1 * vs - eps_g), 0.0);
        const double R0 = fmin(p.rmax, (nqp + phk_sqrt_up(d2up)) * (1.0 + 1e-6));
        bool near = true;  // the top `need` computed candidates all lie within R0
#pragma unroll
        for (int r = 0; r < 3; ++r)
            if (r < need) near = near && colnorm[rx[r]] <= R0;
        const double eps_m = near ? eb(R0) : eps_g;
        cert = ((double)rvn1 - (double)rvn) * vs > 2.0 * eps_m;
    }
    return __ballot(cert && (lane & 7) == 0 && lane < 24);
}

__device__ __forceinline__ int wave_sum_i32(int x) { return (int)wave_sum((uint32_t)x); }

// G16 ownership: the 16 lanes of a group share a 256-element row; lane t holds elements 32 i + 2 t + j (i < 8, j < 2) as
// qd[2 i + j], so that every load instruction of the group covers ONE contiguous piece (256 B of a float64 row, 128 B of a
// uint32 row).  With 16 consecutive elements per lane -- the first layout -- each instruction touched 16 lines per query
// (64 per wave) for 16 B each, and the L1's line rate, not latency or HBM, set these kernels' time (clock64 phase timers:
// 16 k cycles per pass with every operand cache-resident; profiles/r02/README.md).
__device__ __forceinline__ double exact_d2_g16(const double (&qd)[16], double Tq, double invT2, const double *row, int t) {
    const double2 *r = reinterpret_cast<const double2 *>(row) + t;   // G16 ownership: one contiguous 256 B per load
    double2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = r[16 * i];
    // all eight loads in flight before the first use: left alone, the scheduler trades them for registers and emits
    // load, wait, 4 FMAs, load, wait, ... -- eight exposed round trips per row
    __builtin_amdgcn_sched_barrier(0);
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double d0 = fma(-Tq, v[i].x, qd[2 * i]), d1 = fma(-Tq, v[i].y, qd[2 * i + 1]);
        acc = fma(d0, d0, fma(d1, d1, acc));
    }
    return group16_sum(acc) * invT2;
}
// two rows at once (the nearest centroid of either class): sixteen loads in flight, one round trip
__device__ __forceinline__ void exact_d2_pair_g16(const double (&qd)[16], double Tq, double invT2, const double *rowa,
                                                  const double *rowb, int t, double &da, double &db) {
    const double2 *ra = reinterpret_cast<const double2 *>(rowa) + t, *rb = reinterpret_cast<const double2 *>(rowb) + t;
    double2 va[8], vb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        va[i] = ra[16 * i];
        vb[i] = rb[16 * i];
    }
    __builtin_amdgcn_sched_barrier(0);
    double acca = 0.0, accb = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double a0 = fma(-Tq, va[i].x, qd[2 * i]), a1 = fma(-Tq, va[i].y, qd[2 * i + 1]);
        const double b0 = fma(-Tq, vb[i].x, qd[2 * i]), b1 = fma(-Tq, vb[i].y, qd[2 * i + 1]);
        acca = fma(a0, a0, fma(a1, a1, acca));
        accb = fma(b0, b0, fma(b1, b1, accb));
    }
    da = group16_sum(acca) * invT2;
    db = group16_sum(accb) * invT2;
}

// element i of a small register array, by selects: a run-time index into a local array sends it to scratch memory
template <typename T>
__device__ __forceinline__ T pick4(const T (&a)[4], int i) { return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : a[3]; }
template <typename T>
__device__ __forceinline__ T pick3(const T (&a)[3], int i) { return i == 0 ? a[0] : i == 1 ? a[1] : a[2]; }

// the high-parts-only decision (phk_decide_h_kernel, phk_rerank_h_kernel): low parts and the lam* table, see score_decide.hip 2d
struct HiParams {
    const _Float16 *lo16;     // [columns][D] low parts (D = 256: in G16 order, see lo_pos() in score_f16.hip)
    double lam_tab[3][65];    // per segment
    double lam_r0[3], lam_inv_step[3];
};

__device__ __forceinline__ double phk_lam_of(const HiParams &hp, int sg, double R) {
    int i = (int)ceil((R - hp.lam_r0[sg]) * hp.lam_inv_step[sg]);
    i = i < 0 ? 0 : (i > 64 ? 64 : i);
    return hp.lam_tab[sg][i];
}

// the brute force's partial record: 3 nearest train columns of a chunk + nearest positive / negative centroid of the chunk
struct FbRecord {
    double d[3];
    uint32_t i[3];
    uint32_t pad;
    double minpos, minneg;
};

__device__ __forceinline__ bool fb_less(double da, uint64_t ia, double db, uint64_t ib) {
    return da < db || (da == db && ia < ib);
}

// chunks the reference is cut into per queued query: 64 while the record workspace holds them (a short queue -- the usual
// two or three rows of a batch -- is then spread over 64 workgroups per row instead of 16: the kernel is a chain of dependent
// passes over a chunk's columns, 25 -> 10 us at configs[1]), else FB_CHUNKS
__host__ __device__ __forceinline__ uint32_t fb_group_chunks(uint64_t count, uint64_t rec_cap) {
    return (rec_cap && count * 64 <= rec_cap) ? 64u : (uint32_t)FB_CHUNKS;
}


// ------------------------------------------------------------------------------------
// launchers -- src_kind: 0 = uint32 count rows, 1 = normalised float64 rows (the SRC template argument of the kernels);
// every one enqueues on ctx->stream
// ------------------------------------------------------------------------------------
int phk_launch_rerank_wave(phk_ctx *ctx, int src_kind, int dsub, bool i8h, unsigned blocks, const void *src, const RerankParams &p);  // score_rerank.hip
int phk_launch_rerank16(phk_ctx *ctx, int src_kind, int mode, unsigned blocks, const void *src, const RerankParams &p);
int phk_launch_rerank_h(phk_ctx *ctx, int dsub, unsigned blocks, const uint32_t *counts, const RerankParams &p, const HiParams &hp);
int phk_launch_decide(phk_ctx *ctx, int src_kind, const void *src, const RerankParams &p);                                             // score_decide.hip
int phk_launch_decide_gen(phk_ctx *ctx, int dsub, const uint32_t *counts, const RerankParams &p);
int phk_launch_decide_h(phk_ctx *ctx, bool knn, bool cen, dim3 grid, dim3 block, const uint32_t *counts, const RerankParams &p, const HiParams &hp);
int phk_launch_fallback_partial(phk_ctx *ctx, int src_kind, size_t lds, const void *src, const RerankParams &p);                       // score_fallback.hip
int phk_launch_fallback_group(phk_ctx *ctx, int src_kind, int dsub, const void *src, const RerankParams &p);
int phk_launch_fallback_merge(phk_ctx *ctx, const RerankParams &p);
int phk_launch_finish_cen(phk_ctx *ctx, uint64_t nb, const double *pend, double *scores);
int phk_launch_gather_list_rows(phk_ctx *ctx, const uint32_t *counts, const uint32_t *rowsum, const uint32_t *list, uint64_t n, uint64_t D,
                                uint32_t *out, uint32_t *out_sum);
int phk_launch_append_queue(phk_ctx *ctx, const uint32_t *list, uint32_t n, uint32_t *fb_list, uint32_t *fb_count, uint32_t *q_count);
int phk_score_fallback_init_device(phk_ctx *ctx);
