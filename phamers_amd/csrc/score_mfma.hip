// score_mfma.hip -- the fast scoring path for D = 256 (k = 4):
//
//   1. phk_knn_mfma_kernel   fp32-input MFMA (v_mfma_f32_32x32x2_f32) "distance GEMM" of the
//      centred queries against the centred train rows + centroids, with a fused running
//      top-4 per (query, column segment, half-list) kept in registers -- nothing of the
//      N x (M+C) product ever reaches memory.  This is the dense contraction behind
//      scikit-learn's brute-force k-NN (scripts/learning.py:127) and behind the
//      nearest-centroid search (scripts/learning.py:59-66).
//   2. phk_rerank_kernel     per query: certify the candidate order against a rigorous fp32
//      error bound; where certified take the vote from the labels, otherwise (and always for
//      the two nearest-centroid distances the proximity metric needs, scripts/phamer.py:
//      198-210) recompute direct-difference float64 distances for the candidates.  A query
//      whose candidate set cannot be certified to contain the true neighbours is queued for
//   3. phk_knn_fallback_kernel  exact float64 brute force over every column.
//
// So the MFMA pass only ever PROPOSES candidates; every emitted number is decided by float64
// arithmetic of the same form as the reference's (direct differences), or by a certified
// ordering.
//
// Ranking quantity.  With mu = mean train row, r' = r - mu, q' = q - mu (distances are
// translation invariant) the kernel maximises  v = q'.r' - |r'|^2/2  = (|q'|^2 - |q-r|^2)/2.
// The -|r'|^2/2 term rides through the MFMA as a 129th k-step (A = -|r'|^2/2, B = 1).
//
// Operand layout ("fragment order").  v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31]
// [kk = l>>5] and B[kk = l>>5][j = l&31].  The summation index is relabelled so that lane
// half h = l>>5 covers dimensions 128h .. 128h+127: MFMA step s (0..127) contracts
// dimensions {s, 128+s}.  Column block cb (32 train rows / centroids) is stored as
//     Bf[cb][g = 0..32][lane][4]  (float4 per lane per group, 1 KiB per wave-load):
//     g < 32 : element e = r'[32cb + (lane&31)][128*(lane>>5) + 4g + e]
//     g = 32 : element 0 = (lane < 32) ? -|r'|^2/2 : 0        (norm step), rest 0
// and a wave keeps its 32 queries' q' in 128 VGPRs for the whole sweep (q'[j][128h + s]).
#include <stdlib.h>

#include "phk_common.h"
#include "score_model.h"

#include <cmath>
#include <vector>

#include "score_lists.h"

#define NG 33          // 32 k-groups + the norm group
#ifndef PHK_HI_REFINE
#define PHK_HI_REFINE 6  // candidates per query whose low product the high-parts-only decision stage evaluates
#endif
#define FB_CHUNKS 16   // column chunks per queued query in the exact brute-force fallback
#define FB_LDS_MAX (160u * 1024u - 1024u)   // its dynamic LDS: one chunk of distances + the query, float64

__global__ void phk_rowsum_kernel(const uint32_t *__restrict__ counts, uint64_t N, uint64_t D, uint32_t *__restrict__ out);   // score_f16.hip

bool phk_fast_supports_dim(uint64_t D);

// ------------------------------------------------------------------------------------
// model build (host): centre, round to fp32, fragment-order, upload
// ------------------------------------------------------------------------------------
int phk_model_build_fast(phk_ctx *ctx, phk_model *m, const double *pos, const double *neg,
                         const double *cpos, const double *cneg) {
    (void)ctx;
    m->fast = false;
    const uint64_t D = m->D;
    // MFMA proposal paths: D a multiple of 256 up to 4096 (k = 4, 5, 6) and up to 3 neighbours
    if (!phk_fast_supports_dim(D) || m->kn > CAND - 1) return PHK_OK;  // exact path serves other shapes
    if (m->M >= (1ull << 31)) return PHK_OK;
    // the MFMA path's last resort (phk_fallback_partial_kernel) keeps one chunk of float64 distances + the query in LDS
    if (((m->M + m->n_cpos + m->n_cneg + FB_CHUNKS - 1) / FB_CHUNKS + D) * sizeof(double) > FB_LDS_MAX) return PHK_OK;
    std::vector<double> mu(D, 0.0);
    for (uint64_t r = 0; r < m->n_pos; ++r)
        for (uint64_t d = 0; d < D; ++d) mu[d] += pos[r * D + d];
    for (uint64_t r = 0; r < m->n_neg; ++r)
        for (uint64_t d = 0; d < D; ++d) mu[d] += neg[r * D + d];
    double mu2 = 0.0;
    std::vector<float> mu32(D);
    for (uint64_t d = 0; d < D; ++d) {
        mu[d] /= (double)m->M;
        if (!(mu[d] == mu[d]) || std::isinf(mu[d])) return PHK_OK;  // NaN/inf train data: exact path
        mu32[d] = (float)mu[d];
        mu[d] = (double)mu32[d];  // centre by the fp32-representable vector: q' is then formed alike on both paths
        mu2 += mu[d] * mu[d];
    }
    m->n_rblk_ref = (uint32_t)phk_div_up(m->M, 32);
    m->n_rblk_pos = (uint32_t)phk_div_up(m->n_cpos, 32);
    m->n_rblk_neg = (uint32_t)phk_div_up(m->n_cneg, 32);
    double max_norm = 0.0;
    std::vector<double> colnorm(m->M + m->n_cpos + m->n_cneg + 1, 0.0);  // |r'| of every real column
    {   // |r'| of every column as the float32-rounded centred row gives it
        auto norms = [&](const double *rows, uint64_t n, double *out) {
            phk_parallel_for(n, [&](uint64_t r) {
                double s2 = 0.0;
                for (uint64_t d = 0; d < D; ++d) {
                    const double v = (double)(float)(rows[r * D + d] - mu[d]);
                    s2 += v * v;
                }
                out[r] = std::sqrt(s2);
            });
            for (uint64_t r = 0; r < n; ++r)
                if (out[r] > max_norm) max_norm = out[r];
        };
        norms(pos, m->n_pos, colnorm.data());
        norms(neg, m->n_neg, colnorm.data() + m->n_pos);
        if (m->n_cpos) norms(cpos, m->n_cpos, colnorm.data() + m->M);
        if (m->n_cneg) norms(cneg, m->n_cneg, colnorm.data() + m->M + m->n_cpos);
    }
    if (!(max_norm == max_norm) || std::isinf(max_norm)) return PHK_OK;
    if (hipMalloc(&m->d_mu32, D * sizeof(float)) != hipSuccess) return PHK_ERR_NOMEM;
    if (hipMalloc(&m->d_mu64, D * sizeof(double)) != hipSuccess) return PHK_ERR_NOMEM;
    if (hipMalloc(&m->d_colnorm, colnorm.size() * sizeof(double)) != hipSuccess) return PHK_ERR_NOMEM;
    if (hipMemcpy(m->d_colnorm, colnorm.data(), colnorm.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->d_mu32, mu32.data(), D * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->d_mu64, mu.data(), D * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        return PHK_ERR_HIP;
    PHK_TRY(phk_model_build_f16(m, pos, neg, cpos, cneg, mu.data(), colnorm.data()));
    PHK_TRY(phk_model_build_i8(m, pos, neg, cpos, cneg, mu.data(), colnorm.data()));
    m->h_mu = mu;
    m->max_colnorm_train = 0.0;
    for (uint64_t c = 0; c < m->M; ++c)
        if (colnorm[c] > m->max_colnorm_train) m->max_colnorm_train = colnorm[c];
    m->max_colnorm = max_norm;
    m->mu_norm = std::sqrt(mu2);
    {
        double t2 = 0.0;
        for (uint64_t d = 0; d < D; ++d) t2 += (mu[d] - 1.0 / (double)D) * (mu[d] - 1.0 / (double)D);
        m->mu_tilde_norm = std::sqrt(t2) * (1.0 + 1.0e-12);
    }
    m->fast = true;
    return PHK_OK;
}

void phk_model_free_fast(phk_model *m) {
    if (m->d_colnorm) (void)hipFree(m->d_colnorm);
    if (m->d_Af16) (void)hipFree(m->d_Af16);
    m->d_Af16 = nullptr;
    if (m->d_A8) (void)hipFree(m->d_A8);
    if (m->d_A8h) (void)hipFree(m->d_A8h);
    if (m->d_L8) (void)hipFree(m->d_L8);
    if (m->d_T8) (void)hipFree(m->d_T8);
    if (m->d_T8h) (void)hipFree(m->d_T8h);
    m->d_T8h = nullptr;
    m->d_A8 = m->d_A8h = nullptr;
    m->d_L8 = nullptr;
    m->d_T8 = nullptr;
    if (m->d_term_orig) (void)hipFree(m->d_term_orig);
    if (m->d_col_mask) (void)hipFree(m->d_col_mask);
    m->d_term_orig = nullptr;
    m->d_col_mask = nullptr;
    if (m->d_betah16) (void)hipFree(m->d_betah16);
    m->d_betah16 = nullptr;
    if (m->d_Af16h) (void)hipFree(m->d_Af16h);
    if (m->d_lo16) (void)hipFree(m->d_lo16);
    m->d_Af16h = nullptr;
    m->d_lo16 = nullptr;
    if (m->d_cn16) (void)hipFree(m->d_cn16);
    if (m->d_beta16) (void)hipFree(m->d_beta16);
    m->d_cn16 = nullptr;
    if (m->d_mu32) (void)hipFree(m->d_mu32);
    if (m->d_mu64) (void)hipFree(m->d_mu64);
    m->d_colnorm = nullptr;
    m->d_mu32 = nullptr;
    m->d_mu64 = nullptr;
}

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

// Resolve one segment for the wave's query: find the `need` best columns.
//   returns false if the candidate set cannot be certified (-> fallback queue);
//   out_idx[0..need) = column indices of the best; out_d2 = exact d^2 of the best (computed when
//   want_d2, or when the order had to be decided by exact distances).
// Columns with |r'| > |q'| + d_need cannot be among the `need` nearest (triangle inequality), so the
// error bound only has to hold for columns with |r'| <= R0 = |q'| + (upper bound of d_need).
template <int DSUB>
__device__ bool resolve_segment(const RerankParams &p, uint64_t q, int seg, uint32_t ncols, int need,
                                const double (&qd)[4 * DSUB], double Tq, double invT2, double nqp2, const ErrBound &eb, const double vs,
                                const double *rows, const double *colnorm, bool want_d2, int lane,
                                uint32_t (&out_idx)[3], double &out_d2, float pre_v, uint32_t pre_i, float pre_u,
                                bool allow_margin = true, double u_extra = 0.0) {
    // lanes 0..7 hold the 8 candidates (half = lane>>2, slot = lane&3), out of the lists the caller loaded up front
    float v = __shfl(pre_v, seg * 8 + (lane & 7));
    uint32_t ix = __shfl(pre_i, seg * 8 + (lane & 7));
    if (lane >= 8 || ix >= ncols) v = -3.0e38f;  // padding / empty slot
    if (lane >= 8) ix = 0xFFFFFFFFu;
    // vs: computed values -> v units (a power of two, divided by the row sum for the count-exact proposal)
    // every column the two half-lists dropped has a computed value <= the larger of their
    // best-dropped values (-3e38 when nothing real was dropped)
    // (u_extra: lists whose values are short of the true ones by up to that much -- the two-part int8 sweep)
    const double U = fmax((double)__shfl(pre_u, seg * 2), (double)__shfl(pre_u, seg * 2 + 1)) * vs + u_extra;
    // rank of each candidate among the 8 (descending v, ties by lane)
    int rank = 0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const float ov = __shfl(v, m);
        rank += (lane < 8 && (ov > v || (ov == v && m < lane))) ? 1 : 0;
    }
    float rv[4];
    uint32_t ri[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned long long bal = __ballot(lane < 8 && rank == r);
        const int srcl = __ffsll((long long)bal) - 1;
        rv[r] = __shfl(v, srcl);
        ri[r] = __shfl(ix, srcl);
    }
    const double nqp = sqrt(nqp2);
    const double eps_g = eb(p.rmax);  // holds for every column
    if (allow_margin && ri[need - 1] < ncols) {
        // upper bound of the true need-th nearest distance from the computed candidates
        const double d2up = fmax(nqp2 - 2.0 * ((double)rv[need - 1] * vs - eps_g), 0.0);
        const double R0 = fmin(p.rmax, (nqp + sqrt(d2up)) * (1.0 + 1e-6));
        bool near = true;  // the top `need` computed candidates all lie within R0
        for (int r = 0; r < need; ++r) near = near && colnorm[ri[r]] <= R0;
        const double eps_m = near ? eb(R0) : eps_g;
        // certified by margin: the need-th and (need+1)-th computed values are > 2 eps apart
        if (((double)rv[need - 1] - (double)rv[need]) * vs > 2.0 * eps_m) {
#pragma unroll
            for (int r = 0; r < 3; ++r) out_idx[r] = ri[r];
            if (want_d2) out_d2 = exact_d2<DSUB>(qd, Tq, invT2, rows + (uint64_t)ri[0] * (256 * DSUB), lane);
            return true;
        }
    }
    if (lane == 0) atomicAdd(phk_stat_word(p, p.fb_count + 1, 1), 1u);  // statistics: resolved by exact distances
    // not certified: exact float64 distances for all (valid) candidates
    double best[3] = {INFINITY, INFINITY, INFINITY};
    uint32_t bidx[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (int m = 0; m < 8; ++m) {
        const uint32_t c = __shfl(ix, m);
        if (c >= ncols) continue;
        const double d2 = exact_d2<DSUB>(qd, Tq, invT2, rows + (uint64_t)c * (256 * DSUB), lane);
        // insert (d2, c) ascending; ties to the lower column index
        if (d2 < best[2] || (d2 == best[2] && c < bidx[2])) {
            best[2] = d2; bidx[2] = c;
            if (best[2] < best[1] || (best[2] == best[1] && bidx[2] < bidx[1])) {
                double t = best[1]; best[1] = best[2]; best[2] = t;
                uint32_t ti = bidx[1]; bidx[1] = bidx[2]; bidx[2] = ti;
                if (best[1] < best[0] || (best[1] == best[0] && bidx[1] < bidx[0])) {
                    t = best[0]; best[0] = best[1]; best[1] = t;
                    ti = bidx[0]; bidx[0] = bidx[1]; bidx[1] = ti;
                }
            }
        }
    }
    if (bidx[need - 1] == 0xFFFFFFFFu) return false;
    // the need-th best (exact) must beat what any dropped column within reach could be
    const double R0x = fmin(p.rmax, (nqp + sqrt(best[need - 1])) * (1.0 + 1e-6));
    const double tv = 0.5 * (nqp2 - best[need - 1]);
    if (!(tv > U + eb(R0x))) return false;
#pragma unroll
    for (int r = 0; r < 3; ++r) out_idx[r] = bidx[r];
    out_d2 = best[0];
    return true;
}

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

// one query, one wave (the body of phk_rerank_kernel)
// MU_LDS: the training mean is read from LDS at byte offset mu_lds (address space 3: a generic pointer would turn every
// read into a flat load, which also counts on the vector-memory counter and serialises the kernel's other loads)
__device__ __forceinline__ int wave_sum_i32(int x) { return (int)wave_sum((uint32_t)x); }

// I8H: the lists come from the two-part int8 sweep (see RerankParams::L8 and the refinement step below)
template <int SRC, int DSUB, bool MU_LDS, bool I8H>
__device__ __forceinline__ void rerank_one_query(const void *__restrict__ src, const RerankParams &p, uint64_t q, uint32_t mu_lds,
                                                 int lane) {
    constexpr int D = 256 * DSUB;
    typedef __attribute__((address_space(3))) const double lds_cdouble;
    // this lane's query elements, dims 256*sub + 4*lane .. +3: the integer counts with their sum Tq (count rows) or the
    // float64 row with Tq = 1 -- see exact_d2
    double qd[4 * DSUB];
    double Tq = 1.0, invT2 = 1.0;
    double vs = p.vscale;
    bool nan_row = false;
    double ssq = 0.0, rtq = 1.0;   // count rows: sum of squares, reciprocal of the row sum
    CenteredOperand cop = {0.0, 0.0, 0.0, 0.0};
    uint32_t xq[I8H ? DSUB : 1];   // I8H: this lane's centred counts as four int8 per 256-dimension chunk (the sweep's operand)
    // The candidate lists of all three segments are requested first, beside the query row: lane l < 24 holds candidate l & 7
    // of segment l >> 3, lane l < 6 the best dropped value of half-list l.  Loaded where they are used, each segment's
    // lists were one more dependent round trip in a kernel that is a chain of them (10 M queries at configs[2]).
    float pre_v = -3.0e38f, pre_u = -3.0e38f;
    uint32_t pre_i = 0xFFFFFFFFu;
    if (lane < 24) {
        const uint64_t o = cand_at(lane >> 3, (lane & 7) >> 2, lane & 3, q, p.N);
        pre_v = p.cand_v[o];
        pre_i = p.cand_i[o];
    }
    if (lane < 6) pre_u = p.cand_u[candu_at(lane >> 1, lane & 1, q, p.N)];
    if (SRC == 0) {
        const uint32_t *row = static_cast<const uint32_t *>(src) + q * D;
        uint4 c[DSUB];
        uint32_t s = 0;
#pragma unroll
        for (int sub = 0; sub < DSUB; ++sub) {
            c[sub] = reinterpret_cast<const uint4 *>(row + 256 * sub)[lane];
            s += c[sub].x + c[sub].y + c[sub].z + c[sub].w;
        }
        s = wave_sum(s);
        nan_row = s == 0;
        const double ds = (double)s, ry = 1.0 / ds;
        if (p.per_row_scale) vs = p.vscale * ry;   // (= vscale / ds up to the reciprocal's rounding; vs scales margins and bounds, never a score)
        uint32_t cmx = 0, cmn = 0xFFFFFFFFu;
        double sq = 0.0, qc2 = 0.0;
        const double rcen = (double)phk_row_center(s, D);
        rtq = ry;
        Tq = ds;
        invT2 = 1.0 / (ds * ds);
        if (I8H) {
            const uint32_t cen = phk_row_center(s, D);
#pragma unroll
            for (int sub = 0; sub < DSUB; ++sub)   // (rows beyond the int8 range have empty lists: their bytes are never used)
                xq[sub] = ((c[sub].x - cen) & 0xFFu) | (((c[sub].y - cen) & 0xFFu) << 8) | (((c[sub].z - cen) & 0xFFu) << 16) |
                          ((c[sub].w - cen) << 24);
        }
#pragma unroll
        for (int sub = 0; sub < DSUB; ++sub) {
            const double x0 = (double)c[sub].x, x1 = (double)c[sub].y, x2 = (double)c[sub].z, x3 = (double)c[sub].w;
            qd[4 * sub + 0] = x0; qd[4 * sub + 1] = x1; qd[4 * sub + 2] = x2; qd[4 * sub + 3] = x3;
            sq = fma(x0, x0, fma(x1, x1, fma(x2, x2, fma(x3, x3, sq))));
            if (!I8H) {   // (the largest / smallest count: for the maximum-norm term of the f16 chains' bound only)
                cmx = max(max(cmx, max(c[sub].x, c[sub].y)), max(c[sub].z, c[sub].w));
                cmn = min(min(cmn, min(c[sub].x, c[sub].y)), min(c[sub].z, c[sub].w));
            }
            if (!I8H && DSUB > 1 && p.per_row_scale && p.eb_cAmax > 0.0) {   // norm of this 256-dimension chunk of c - c0 (f16 chains)
                const double y0 = x0 - rcen, y1 = x1 - rcen, y2 = x2 - rcen, y3 = x3 - rcen;
                qc2 = fmax(qc2, wave_sum(fma(y0, y0, fma(y1, y1, fma(y2, y2, y3 * y3)))));
            }
        }
        ssq = wave_sum(sq);   // sum of squared counts, exact: |q|^2 = ssq / T^2
        if (p.per_row_scale && !nan_row) {  // the operand of the count-exact chain: the counts minus their centre
            // (largest / smallest count: only where the bound has a maximum-norm term -- the f16 chains)
            const double cmax = (!I8H && p.eb_cI > 0.0) ? wave_max((double)cmx) : rcen, cmin = (!I8H && p.eb_cI > 0.0) ? -wave_max(-(double)cmn) : rcen;
            cop = phk_centered_operand_fast(ssq, ds, ry, cmax, cmin, (double)D, p.eb_hsum);
            if (!I8H && DSUB > 1 && p.eb_cAmax > 0.0) cop.Q = phk_sqrt_up(qc2) * ry * (1.0 + 1e-9);   // (with the observed running sums)
        }
    } else {
        const double *row = static_cast<const double *>(src) + q * D;
        bool bad = false;
#pragma unroll
        for (int sub = 0; sub < DSUB; ++sub) {
            const double2 a = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane];
            const double2 b = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane + 1];
            qd[4 * sub + 0] = a.x; qd[4 * sub + 1] = a.y; qd[4 * sub + 2] = b.x; qd[4 * sub + 3] = b.y;
            bad |= a.x != a.x || a.y != a.y || b.x != b.x || b.y != b.y;
        }
        nan_row = __any(bad);
    }
    if (nan_row) {  // zero-count contig: the reference's normalised row is NaN
        if (lane == 0) {
            p.scores[p.q_base + q] = __builtin_nan("");
            if (p.status) atomicAdd(p.status, 1u);
        }
        return;
    }
    double aq = 0.0, ap = 0.0, am = 0.0, pc2 = 0.0;
#pragma unroll
    for (int sub = 0; sub < DSUB; ++sub) {
        double2 m0, m1;
        if (MU_LDS) {
            lds_cdouble *lm = (lds_cdouble *)(uintptr_t)mu_lds + 256 * sub + 4 * lane;
            m0.x = lm[0]; m0.y = lm[1]; m1.x = lm[2]; m1.y = lm[3];
        } else {
            m0 = reinterpret_cast<const double2 *>(p.mu64 + 256 * sub)[2 * lane];
            m1 = reinterpret_cast<const double2 *>(p.mu64 + 256 * sub)[2 * lane + 1];
        }
        // q' in units of 1 / Tq (count rows: c_i - Tq mu_i; float64 rows: Tq = 1, q_i - mu_i exactly)
        const double c0 = fma(-Tq, m0.x, qd[4 * sub + 0]), c1 = fma(-Tq, m0.y, qd[4 * sub + 1]);
        const double c2 = fma(-Tq, m1.x, qd[4 * sub + 2]), c3 = fma(-Tq, m1.y, qd[4 * sub + 3]);
        if (SRC != 0)
            aq = fma(qd[4 * sub + 0], qd[4 * sub + 0], fma(qd[4 * sub + 1], qd[4 * sub + 1],
                     fma(qd[4 * sub + 2], qd[4 * sub + 2], fma(qd[4 * sub + 3], qd[4 * sub + 3], aq))));
        const double apc = fma(c0, c0, fma(c1, c1, fma(c2, c2, c3 * c3)));
        ap += apc;
        if (!(SRC == 0 && p.per_row_scale)) {   // split-f16 lists only: maximum norm and chunk norms of q'
            am = fmax(fmax(am, fmax(fabs(c0), fabs(c1))), fmax(fabs(c2), fabs(c3)));
            if (DSUB > 1) pc2 = fmax(pc2, wave_sum(apc));
        }
    }
    // |q|^2 enters the error bound only: for count rows from the exact sum of squares (+ slack for the two roundings)
    const double nq2 = SRC == 0 ? ssq * (rtq * rtq) * (1.0 + 1e-12) : wave_sum(aq);
    const double nqp2 = wave_sum(ap) * invT2;
    if (SRC == 0) {   // (split-f16 lists of count rows: the maximum norm and the chunk norms in units of q)
        am *= rtq;
        pc2 *= invT2;
    }
    ErrBound eb;   // (its norms: upper bounds, phk_sqrt_up)
    eb.A = phk_sqrt_up(nq2) + p.mu_norm;
    const double nqp_up = phk_sqrt_up(nqp2);
    eb.P = nqp_up;
    eb.cA = p.eb_cA; eb.cP = p.eb_cP; eb.cR = p.eb_cR; eb.cabs = p.eb_abs; eb.cQ = p.eb_cQ; eb.cI = p.eb_cI; eb.cIf = p.eb_cIf;
    if (SRC == 0 && p.per_row_scale) {   // count-exact lists: the chain's query operand is c - c0
        eb.Q = cop.Q; eb.I = cop.I; eb.habs = cop.habs;
        eb.P = phk_sqrt_up(nqp2 + cop.shift2);   // (high-parts-only lists: what the low parts multiply)
    } else {                             // split-f16 lists: the chain's query operand is q' (largest chunk norm)
        eb.Q = DSUB > 1 ? sqrt(pc2) * (1.0 + 1e-12) : eb.P;
        eb.I = wave_max(am);
    }
    if (p.cand_a)   // the running sums this query's sweep met at its chunk boundaries, in v units
        eb.habs += p.eb_cAmax * 5.9604644775390625e-08 * (double)fmaxf(p.cand_a[q], p.cand_a[p.N + q]) * vs * 1.001;

    bool ok = true;
    double knn = 0.0, cen = 0.0;
    uint32_t idx[3];
    double d2;
    // Lists of the two-part int8 sweep: a value w^h lacks g_j S_L, at most el = |c - c0| / T * lam8[segment] in v units.  With
    // h_need the need-th best list value of a segment and e_h = el + eps: every column that can be among the `need` nearest
    // has w^h >= h_need - 2 e_h (the leaders' true values are >= h_need - e_h; below the window a true value is < that).  The
    // window's members that are list members get g_j S_L added -- an exact integer dot product with the column's L digits,
    // 4 DSUB bytes per lane -- and then carry the three-part sweep's value; the window must end above everything the
    // half-lists dropped (uok), else the segment is left to the exact candidate distances.  Lane l < 24 works for candidate
    // l & 7 of segment l >> 3, as in certify_segments.
    uint64_t uok = ~0ull;
    // (a row beyond the int8 operand: the sweep stored its sentinel, 3e38 as the best dropped value)
    const bool big_row = p.q2_count ? __any(lane < 6 && pre_u > 1.0e38f) != 0 : false;
    if (I8H) {
        const int g3 = lane >> 3, gb = lane & 56;
        const uint32_t ncols = g3 == 0 ? (uint32_t)p.M : g3 == 1 ? (uint32_t)p.n_cpos : g3 == 2 ? (uint32_t)p.n_cneg : 0u;
        const int need = g3 == 0 ? p.kn : 1;
        const bool valid = pre_i < ncols;
        const float gq = valid ? p.T8[(uint64_t)(p.t8_blk[g3 < 3 ? g3 : 0] + (pre_i >> 5)) * 64 + (pre_i & 31u)] : 0.0f;
        const uint32_t gcol = pre_i + (g3 == 0 ? 0u : g3 == 1 ? (uint32_t)p.M : (uint32_t)(p.M + p.n_cpos));
        const float v = valid ? pre_v : -3.0e38f;
        int rank = 0;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float ov = __shfl(v, gb | m);
            rank += (ov > v || (ov == v && m < (lane & 7))) ? 1 : 0;
        }
        const unsigned long long bal = __ballot(valid && rank == need - 1);
        const unsigned long long mine = (bal >> gb) & 0xFFull;
        const float hneed = __shfl(v, gb + (mine ? __ffsll((long long)mine) - 1 : 0));
        const double el = cop.Q * p.lam8[g3 < 3 ? g3 : 0] * (1.0 + 1.0e-6);
        const double thr = (double)hneed * vs - 2.0 * (el + eb(p.rmax));
        const bool inwin = valid && mine != 0 && (double)v * vs >= thr;
        const double Ug = (double)fmaxf(__shfl(pre_u, 2 * (g3 < 3 ? g3 : 0)), __shfl(pre_u, 2 * (g3 < 3 ? g3 : 0) + 1)) * vs;
        uok = __ballot(mine != 0 && Ug < thr && (lane & 7) == 0 && lane < 24);
        // a window of exactly `need` members is decided as it stands (the members ARE the nearest; the margin test below
        // passes on the list values, the next one lying 2 e_h lower): only wider windows are refined
        unsigned long long wm = __ballot(inwin && lane < 24);
        const bool wide = __popcll((wm >> gb) & 0xFFull) > need;
        wm = __ballot(inwin && wide && lane < 24);
        while (wm) {   // wave-uniform; two members per round trip
            const int m0 = __ffsll((long long)wm) - 1;
            wm &= wm - 1;
            const int m1 = wm ? __ffsll((long long)wm) - 1 : m0;
            wm &= wm - 1;
            const uint32_t *r0 = reinterpret_cast<const uint32_t *>(p.L8 + (uint64_t)__shfl(gcol, m0) * D) + lane;
            const uint32_t *r1 = reinterpret_cast<const uint32_t *>(p.L8 + (uint64_t)__shfl(gcol, m1) * D) + lane;
            uint32_t w0[DSUB], w1[DSUB];
#pragma unroll
            for (int sub = 0; sub < DSUB; ++sub) {
                w0[sub] = r0[64 * sub];
                w1[sub] = r1[64 * sub];
            }
            int a0 = 0, a1 = 0;
#pragma unroll
            for (int sub = 0; sub < DSUB; ++sub) {
                a0 = __builtin_amdgcn_sdot4((int)xq[sub], (int)w0[sub], a0, false);
                a1 = __builtin_amdgcn_sdot4((int)xq[sub], (int)w1[sub], a1, false);
            }
            a0 = wave_sum_i32(a0);
            a1 = wave_sum_i32(a1);
            if (lane == m0) pre_v = fmaf((float)a0, gq, pre_v);
            if (lane == m1 && m1 != m0) pre_v = fmaf((float)a1, gq, pre_v);
        }
    }
    // all three margin tests side by side; what they certify needs no further list work
    uint32_t cri[3];
    uint64_t certified = certify_segments(p, nqp2, nqp_up, eb, vs, lane, pre_v, pre_i, cri);
    if (I8H) {
        certified &= uok;
        // statistics: a window of a segment the method uses reached past the lists (rare: one atomic per such query)
        const uint64_t used = ((p.method & PHK_METHOD_KNN) ? 0x01ull : 0ull) | ((p.method & PHK_METHOD_KMEANS) ? 0x010100ull : 0ull);
        if (lane == 0 && (uok & used) != used && p.counters && !big_row) atomicAdd(p.counters + 9, 1u);
    }
    const double uex0 = I8H ? cop.Q * p.lam8[0] * (1.0 + 1.0e-6) : 0.0, uex1 = I8H ? cop.Q * p.lam8[1] * (1.0 + 1.0e-6) : 0.0,
                 uex2 = I8H ? cop.Q * p.lam8[2] * (1.0 + 1.0e-6) : 0.0;
    if (p.method & PHK_METHOD_KNN) {
        if (certified & 1ull) {
#pragma unroll
            for (int r = 0; r < 3; ++r) idx[r] = __shfl(cri[r], 0);
        } else {
            ok = resolve_segment<DSUB>(p, q, 0, (uint32_t)p.M, p.kn, qd, Tq, invT2, nqp2, eb, vs, p.R64, p.colnorm, false, lane, idx, d2, pre_v, pre_i, pre_u, !I8H, uex0);
        }
        if (ok) {
            int votes = 0;
            for (int r = 0; r < p.kn; ++r) votes += p.labels[idx[r]] ? 1 : 0;
            knn = (2 * votes > p.kn) ? 1.0 : -1.0;
        }
    }
    if (ok && (p.method & PHK_METHOD_KMEANS)) {
        double dp2 = 0.0, dn2 = 0.0;
        if (certified & (1ull << 8))
            dp2 = exact_d2<DSUB>(qd, Tq, invT2, p.C64 + (uint64_t)__shfl(cri[0], 8) * D, lane);
        else
            ok = resolve_segment<DSUB>(p, q, 1, (uint32_t)p.n_cpos, 1, qd, Tq, invT2, nqp2, eb, vs, p.C64, p.colnorm + p.M, true, lane, idx, dp2, pre_v, pre_i, pre_u, !I8H, uex1);
        if (ok && (certified & (1ull << 16)))
            dn2 = exact_d2<DSUB>(qd, Tq, invT2, p.C64 + (p.n_cpos + (uint64_t)__shfl(cri[0], 16)) * D, lane);
        else if (ok)
            ok = resolve_segment<DSUB>(p, q, 2, (uint32_t)p.n_cneg, 1, qd, Tq, invT2, nqp2, eb, vs, p.C64 + p.n_cpos * D,
                                 p.colnorm + p.M + p.n_cpos, true, lane, idx, dn2, pre_v, pre_i, pre_u, !I8H, uex2);
        if (ok) {
            if (p.pend) {   // two square roots, a division and a tanh in float64 are ~180 instructions of this wave, for one
                            // number: left to a lane-per-query kernel (phk_finish_cen_kernel, the same expressions)
                if (lane == 0) {
                    const uint64_t oq = p.out_map ? (uint64_t)p.out_map[q] : q;
                    p.pend[2 * oq] = dp2;
                    p.pend[2 * oq + 1] = dn2;
                }
            } else {
                const double ep = sqrt(dp2), en = sqrt(dn2);
                cen = tanh((en - ep) / (ep + en));  // scripts/phamer.py:206-209
            }
        }
    }
    if (lane == 0) {
        const uint64_t oq = p.out_map ? (uint64_t)p.out_map[q] : q;
        if (ok) {
            p.scores[p.q_base + oq] = knn + cen;  // scripts/phamer.py:313 (cen: see pend)
        } else if (p.q2_count && big_row) {
            p.q2_big[atomicAdd(p.q2_count + 1, 1u)] = (uint32_t)q;
        } else if (p.q2_count && p.q2_wide) {
            p.q2_wide[atomicAdd(p.q2_count, 1u)] = (uint32_t)q;
        } else {
            const uint32_t slot = atomicAdd(p.fb_count, 1u);
            p.fb_list[slot] = (uint32_t)oq;
        }
    }
}

// One wave per query.  D >= 2048: a workgroup walks its share of the queries (grid-stride) with the training mean in LDS,
// loaded once -- read from memory per query it was 8 D bytes through the vector cache, a quarter of the kernel's traffic
// (configs[4]: 11.4 -> 9.6 ms).  Smaller D: one query per wave and launch slot, the mean from the cache (the 8 KB of
// D = 1024 stay resident there, and the walk was measured slower: configs[2] 37.5 -> 44.7 ms).
#ifndef PHK_RERANK_WAVES
#define PHK_RERANK_WAVES 4
#endif
template <int SRC, int DSUB, bool I8H = false>
__global__ __launch_bounds__(256, (DSUB <= 4 ? PHK_RERANK_WAVES : 1)) void phk_rerank_kernel(const void *__restrict__ src, RerankParams p) {
    constexpr int D = 256 * DSUB;
    constexpr bool WALK = DSUB >= 8;
    const int lane = threadIdx.x & 63;
    __shared__ double s_mu[WALK ? D : 2];
    if (WALK) {
        for (int i = threadIdx.x; i < D / 2; i += 256)
            reinterpret_cast<double2 *>(s_mu)[i] = reinterpret_cast<const double2 *>(p.mu64)[i];
        __syncthreads();
    }
    // listed (slow_back == 2): only the queries phk_rerank_h_kernel passed on (slow_list, fb_count[2] of them)
    const uint64_t nq = p.slow_back == 2 ? (uint64_t)phk_uniform_load(p.fb_count + 2) : p.N;
    const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t mu_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)s_mu;
    uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (WALK) {
        for (; w < nq; w += stride)
            rerank_one_query<SRC, DSUB, true, I8H>(src, p, p.slow_back == 2 ? (uint64_t)(p.slow_list[w] & 0x3FFFFFFFu) : w, mu_lds, lane);
    } else if (w < nq) {   // (no loop: its live state costs the registers that keep four waves per SIMD)
        rerank_one_query<SRC, DSUB, false, I8H>(src, p, p.slow_back == 2 ? (uint64_t)(p.slow_list[w] & 0x3FFFFFFFu) : w, 0u, lane);
    }
}

// ------------------------------------------------------------------------------------
// 2b. the same decision stage for D = 256 with FOUR queries per wavefront (16 lanes each, 16
//     dimensions per lane).  The one-wave-per-query kernel above is latency bound (a chain of ~6
//     dependent memory round trips per query); packing 4 queries into a wave quarters the number of
//     such chains per SIMD.  Control flow is uniform per wave: a group that does not need a step
//     runs it predicated on safe addresses.
// ------------------------------------------------------------------------------------
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

// per-group version of resolve_segment; `live` = this group still needs an answer.  Returns ok.
// element i of a small register array, by selects: a run-time index into a local array sends it to scratch memory
template <typename T>
__device__ __forceinline__ T pick4(const T (&a)[4], int i) { return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : a[3]; }
template <typename T>
__device__ __forceinline__ T pick3(const T (&a)[3], int i) { return i == 0 ? a[0] : i == 1 ? a[1] : a[2]; }
__device__ bool resolve_segment_g16(const RerankParams &p, uint64_t q, int seg, uint32_t ncols, int need,
                                    const double (&qd)[16], double Tq, double invT2, double nqp2, const ErrBound &eb,
                                    const double vs, const double *rows, const double *colnorm, bool want_d2, bool live,
                                    int lane, float v_in, uint32_t ix_in, float u_in, uint32_t (&out_idx)[3],
                                    double &out_d2) {
    // v_in / ix_in: this lane's candidate of the segment (lanes t < 8: half t >> 2, slot t & 3), u_in: the
    // best-dropped value of half-list t & 1 -- loaded by the caller together with the query row, so that the
    // three segments' lists cost one memory round trip, not three
    const int t = lane & 15, base = lane & 48;
    float v = -3.0e38f;
    uint32_t ix = 0xFFFFFFFFu;
    if (t < 8) {
        v = ix_in >= ncols ? -3.0e38f : v_in;
        ix = ix_in;
    }
    const double U = fmax((double)u_in, (double)__shfl_xor(u_in, 1)) * vs;
    // rank among the group's 8 candidates, then values / indices by rank
    int rank = 0;
    float cvv[8];
    uint32_t cix[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        cvv[m] = __shfl(v, base + m);
        cix[m] = __shfl(ix, base + m);
        rank += (t < 8 && (cvv[m] > v || (cvv[m] == v && m < t))) ? 1 : 0;
    }
    float rv[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    uint32_t ri[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int rm = __shfl(rank, base + m);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rv[r] = rm == r ? cvv[m] : rv[r];
            ri[r] = rm == r ? cix[m] : ri[r];
        }
    }
    const double nqp = sqrt(nqp2);
    const double eps_g = eb(p.rmax);
    double eps_w = eps_g;   // the bound the window of the exact route uses (the margin test's)
    bool certified = false;
    if (pick4(ri, need - 1) < ncols) {
        const double d2up = fmax(nqp2 - 2.0 * ((double)pick4(rv, need - 1) * vs - eps_g), 0.0);
        const double R0 = fmin(p.rmax, (nqp + sqrt(d2up)) * (1.0 + 1e-6));
        double cnr[3] = {0.0, 0.0, 0.0};  // independent loads (no short-circuit chain of round trips)
#pragma unroll
        for (int r = 0; r < 3; ++r)
            if (r < need) cnr[r] = colnorm[ri[r]];
        bool near = true;
#pragma unroll
        for (int r = 0; r < 3; ++r) near = near && (r >= need || cnr[r] <= R0);
        const double eps_m = near ? eb(R0) : eps_g;
        eps_w = eps_m;
        certified = ((double)pick4(rv, need - 1) - (double)pick4(rv, need)) * vs > 2.0 * eps_m;
    }
    bool ok = certified;
    double best[3] = {INFINITY, INFINITY, INFINITY};
    uint32_t bidx[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const bool need_exact = live && !certified;
    if (__any(need_exact)) {  // wave-uniform: some group must decide by exact float64 distances
        if (need_exact && t == 0) atomicAdd(phk_stat_word(p, p.fb_count + 1, 1), 1u);
        // only candidates inside the window can be among the `need` nearest: one whose computed value lies more than
        // 2 eps below the need-th best has a true value below the true need-th best (same argument as the margin test).
        // The lists are sorted within each half, so the late slots are skipped by whole waves most of the time.
        const double wthr = (double)pick4(rv, need - 1) * vs - 2.0 * eps_w;
#pragma unroll   // (fully: cix[m] / cvv[m] under a run-time m would live in scratch memory)
        for (int m = 0; m < 8; ++m) {
            const uint32_t c = cix[m];
            const bool valid = need_exact && c < ncols && (double)cvv[m] * vs >= wthr;
            if (!__any(valid)) continue;
            const double d2 = exact_d2_g16(qd, Tq, invT2, rows + (uint64_t)(valid ? c : 0u) * FAST_D, t);
            if (valid && (d2 < best[2] || (d2 == best[2] && c < bidx[2]))) {
                best[2] = d2; bidx[2] = c;
                if (best[2] < best[1] || (best[2] == best[1] && bidx[2] < bidx[1])) {
                    double td = best[1]; best[1] = best[2]; best[2] = td;
                    uint32_t ti = bidx[1]; bidx[1] = bidx[2]; bidx[2] = ti;
                    if (best[1] < best[0] || (best[1] == best[0] && bidx[1] < bidx[0])) {
                        td = best[0]; best[0] = best[1]; best[1] = td;
                        ti = bidx[0]; bidx[0] = bidx[1]; bidx[1] = ti;
                    }
                }
            }
        }
        if (need_exact) {
            ok = false;
            if (pick3(bidx, need - 1) != 0xFFFFFFFFu) {
                const double bneed = pick3(best, need - 1);
                const double R0x = fmin(p.rmax, (nqp + sqrt(bneed)) * (1.0 + 1e-6));
                const double tv = 0.5 * (nqp2 - bneed);
                ok = tv > U + eb(R0x);
            }
        }
    }
    if (certified) {
#pragma unroll
        for (int r = 0; r < 3; ++r) out_idx[r] = ri[r];
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) out_idx[r] = bidx[r];
    }
    if (want_d2) {  // distance to the best column (centroid segments): one exact evaluation when certified
        const uint32_t c0 = ri[0];  // speculative: the row is fetched alongside the column norms, not after the verdict
        const double d2c = exact_d2_g16(qd, Tq, invT2, rows + (uint64_t)(c0 < ncols ? c0 : 0u) * FAST_D, t);
        out_d2 = certified ? d2c : best[0];
    }
    return ok;
}

// MODE 0: every query of the batch; 1: the queries phk_decide_kernel handed over (slow_list); 2: second chance -- rows
// map[0 .. *map_count) with their lists at dense positions (see RerankParams)
template <int SRC, int MODE>
__global__ __launch_bounds__(256, 3) void phk_rerank16_kernel(const void *__restrict__ src, RerankParams p) {
    const int lane = threadIdx.x & 63, t = lane & 15;
    uint64_t qraw = (((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 4 + (lane >> 4);
    bool inrange;
    uint64_t q, ql;   // the query's row in src / scores; the position of its candidate lists
    uint32_t todo = 3u;
    if (MODE == 1) {
        // slow_back == 3: both lists in one launch -- waves from the front of the grid take the front list (counted by
        // fb_count[2]), waves from its end the back list (counters[12]); front + back <= N, so the two never meet, and a
        // wave serves one kind of list.  (Two launches, each over the whole grid, spent 0.09 ms apiece on waves that
        // read a count and left.)
        uint64_t cnt, base = 0, cap = p.slow_cap;
        bool back = p.slow_back == 1;
        if (p.slow_back == 3 && p.sub_lists) {
            // the grid's waves in sub_lists runs of nlw: run s serves pair s, wave lw of the run from the front of its numbering
            // the pair's front list, from its end the back list (front + back <= sub_cap <= 4 (nlw - 2): the two never meet);
            // neighbouring waves work on neighbouring queries of one list
            const uint64_t wave_id = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
            const uint64_t nlw = ((((uint64_t)gridDim.x * blockDim.x) >> 6)) / p.sub_lists;
            if (wave_id >= nlw * p.sub_lists) return;   // (the waves beyond a whole number of runs)
            const uint32_t sl = __builtin_amdgcn_readfirstlane((uint32_t)(wave_id / nlw));   // (wave-uniform: scalar loads below)
            const uint64_t lw = wave_id % nlw;
            base = (uint64_t)sl * p.sub_cap;
            cap = p.sub_cap;
            uint64_t q4 = lw * 4;
            cnt = phk_uniform_load(p.stripes + sl * 32u + 2u);
            if (q4 >= cnt) {
                q4 = (nlw - 1 - lw) * 4;
                cnt = phk_uniform_load(p.stripes + sl * 32u + 12u);
                back = true;
            }
            qraw = q4 + (uint64_t)(lane >> 4);
        } else {
            cnt = phk_uniform_load(p.slow_back == 1 ? p.counters + 12 : p.fb_count + 2);
            if (p.slow_back == 3 && (qraw & ~3ull) >= cnt) {
                const uint64_t nwave4 = (((uint64_t)gridDim.x * blockDim.x) >> 6) * 4;
                qraw = nwave4 - 4 - (qraw & ~3ull) + (qraw & 3ull);   // wave k from the end, same lane group
                cnt = phk_uniform_load(p.counters + 12);
                back = true;
            }
        }
        if ((qraw & ~3ull) >= cnt) return;
        inrange = qraw < cnt;
        // entry = query | todo << 30: which parts are still open (bit 0 the k-NN vote, bit 1 the centroid metric; 0 = both).
        // A part the sender has decided already sits in scores[q] and is only added to.
        const uint64_t pos = inrange ? qraw : cnt - 1;
        const uint32_t entry = p.slow_list[base + (back ? cap - 1 - pos : pos)];
        q = ql = entry & 0x3FFFFFFFu;
        todo = entry >> 30 ? entry >> 30 : 3u;
    } else if (MODE == 2) {
        const uint64_t cnt_all = phk_uniform_load(p.map_count);
        // a handful of rows is cheaper to brute-force than to sweep (one workgroup's sweep is ~0.2 ms of latency):
        // the second proposal pass stands down below PHK_SECOND_MIN rows (score_model.h) and so does this kernel
        const uint64_t cnt = cnt_all < PHK_SECOND_MIN ? 0 : (cnt_all < p.N ? cnt_all : p.N);
        // rows without a second list set (beyond its capacity, or all of them when the pass stood down) go straight to
        // the brute-force queue
        for (uint64_t i = cnt + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt_all; i += (uint64_t)gridDim.x * blockDim.x)
            p.fb_list[atomicAdd(p.fb_count, 1u)] = p.map[i];
        if ((qraw & ~3ull) >= cnt) return;
        inrange = qraw < cnt;
        ql = inrange ? qraw : cnt - 1;
        q = p.map[ql];
    } else {
        if ((qraw & ~3ull) >= p.N) return;  // whole wave past the end
        inrange = qraw < p.N;
        q = ql = inrange ? qraw : p.N - 1;
    }
    double qd[16];
    double vs = p.vscale;
    bool nan_row = false;
    double Tq = 1.0, invT2 = 1.0;  // row sum and 1 / Tq^2 (counts); 1 for float64 rows
    uint32_t cmx = 0, cmn = 0xFFFFFFFFu;   // largest / smallest count of the row
    // the query's six half-lists, fetched with the row
    float lv[NSEG], lu[NSEG];
    uint32_t lix[NSEG];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        const uint64_t e = cand_at(sg, (t >> 2) & 1, t & 3, ql, p.N);
        lv[sg] = p.cand_v[e];
        lix[sg] = p.cand_i[e];
        lu[sg] = p.cand_u[candu_at(sg, t & 1, ql, p.N)];
    }
    if (SRC == 0) {
        const uint2 *row = reinterpret_cast<const uint2 *>(static_cast<const uint32_t *>(src) + q * FAST_D) + t;
        uint2 c[8];
        uint32_t sum = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c[i] = row[16 * i];
            sum += c[i].x + c[i].y;
        }
        sum = group16_sum(sum);
        nan_row = sum == 0;
        const double ds = (double)sum;
        if (p.per_row_scale) vs = p.vscale / ds;
        Tq = ds;
        invT2 = 1.0 / (ds * ds);
#pragma unroll
        for (int i = 0; i < 8; ++i) {   // the counts themselves: see exact_d2_g16
            qd[2 * i + 0] = (double)c[i].x;
            qd[2 * i + 1] = (double)c[i].y;
            cmx = max(cmx, max(c[i].x, c[i].y));
            cmn = min(cmn, min(c[i].x, c[i].y));
        }
        cmx = group16_max(cmx);
        cmn = group16_min(cmn);
    } else {
        const double2 *row = reinterpret_cast<const double2 *>(static_cast<const double *>(src) + q * FAST_D) + t;
        bool bad = false;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double2 v = row[16 * i];
            qd[2 * i] = v.x;
            qd[2 * i + 1] = v.y;
            bad |= v.x != v.x || v.y != v.y;
        }
        // any NaN in the group's row
        unsigned b = bad ? 1u : 0u;
        b |= __shfl_xor(b, 8); b |= __shfl_xor(b, 4); b |= __shfl_xor(b, 2); b |= __shfl_xor(b, 1);
        nan_row = b != 0;
    }
    double aq = 0.0, ap = 0.0, am = 0.0;
    {
        const double2 *mp = reinterpret_cast<const double2 *>(p.mu64) + t;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double2 m = mp[16 * i];
            const double c0 = fma(-Tq, m.x, qd[2 * i]), c1 = fma(-Tq, m.y, qd[2 * i + 1]);
            aq = fma(qd[2 * i], qd[2 * i], fma(qd[2 * i + 1], qd[2 * i + 1], aq));
            ap = fma(c0, c0, fma(c1, c1, ap));
            am = fmax(am, fmax(fabs(c0), fabs(c1)));
        }
    }
    const double sumsq = group16_sum(aq);
    const double nq2 = sumsq * invT2, nqp2 = group16_sum(ap) * invT2;
    ErrBound eb;
    eb.A = sqrt(nq2) + p.mu_norm;
    eb.P = sqrt(nqp2);
    eb.cA = p.eb_cA; eb.cP = p.eb_cP; eb.cR = p.eb_cR; eb.cabs = p.eb_abs; eb.cQ = p.eb_cQ; eb.cI = p.eb_cI; eb.cIf = p.eb_cIf;
    eb.cM = p.eb_cM; eb.M = p.eb_M;
    if (SRC == 0 && p.per_row_scale) {   // count-exact lists: the chain's query operand is c - c0 (see ErrBound)
        const CenteredOperand cop = phk_centered_operand(sumsq, nan_row ? 1.0 : Tq, (double)cmx, (double)cmn, (double)FAST_D, p.eb_hsum);
        eb.Q = cop.Q; eb.I = cop.I; eb.habs = cop.habs + p.eb_babs;
        eb.P = sqrt(nqp2 + cop.shift2);   // (high-parts-only lists: what the low parts multiply)
    } else {                             // split-f16 lists: the chain's query operand is q' (Tq = 1 for float64 rows)
        eb.Q = 0.0; eb.I = group16_max(am) / Tq;
    }

    bool live = inrange && !nan_row;   // NaN rows: every comparison below is false; they are answered separately
    bool ok = true;
    double knn = 0.0, cen = 0.0;
    uint32_t idx[3];
    double d2 = 0.0;
    const bool do_knn = (p.method & PHK_METHOD_KNN) && (todo & 1u), do_cen = (p.method & PHK_METHOD_KMEANS) && (todo & 2u);
    // wave-uniform skips: a segment nobody in the wave needs is not touched
    if (__any(do_knn)) {
        const bool okk = resolve_segment_g16(p, q, 0, (uint32_t)p.M, p.kn, qd, Tq, invT2, nqp2, eb, vs, p.R64, p.colnorm, false,
                                             live && do_knn, lane, lv[0], lix[0], lu[0], idx, d2);
        int votes = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) votes += (r < p.kn && idx[r] < p.M && p.labels[idx[r] < p.M ? idx[r] : 0u]) ? 1 : 0;
        knn = do_knn ? ((2 * votes > p.kn) ? 1.0 : -1.0) : 0.0;
        ok = okk || !do_knn;
    }
    if (__any(do_cen)) {
        double dp2 = 0.0, dn2 = 0.0;
        const bool ok1 = resolve_segment_g16(p, q, 1, (uint32_t)p.n_cpos, 1, qd, Tq, invT2, nqp2, eb, vs, p.C64, p.colnorm + p.M, true,
                                             live && ok && do_cen, lane, lv[1], lix[1], lu[1], idx, dp2);
        const bool ok2 = resolve_segment_g16(p, q, 2, (uint32_t)p.n_cneg, 1, qd, Tq, invT2, nqp2, eb, vs, p.C64 + p.n_cpos * FAST_D,
                                             p.colnorm + p.M + p.n_cpos, true, live && ok && ok1 && do_cen, lane, lv[2], lix[2],
                                             lu[2], idx, dn2);
        ok = ok && ((ok1 && ok2) || !do_cen);
        const double ep = sqrt(dp2), en = sqrt(dn2);
        cen = do_cen ? tanh((en - ep) / (ep + en)) : 0.0;  // scripts/phamer.py:206-209
    }
    // parts the sender had decided already are in scores[q]
    const double prev = (MODE == 1 && todo != 3u && inrange && !nan_row) ? p.scores[p.q_base + q] : 0.0;
    if (t == 0 && inrange) {
        if (nan_row) {  // zero-count contig: the reference's normalised row is NaN
            p.scores[p.q_base + q] = __builtin_nan("");
            if (p.status) atomicAdd(p.status, 1u);
        } else if (ok) {
            p.scores[p.q_base + q] = prev + knn + cen;  // scripts/phamer.py:313
        } else {
            const uint32_t slot = atomicAdd(p.fb_count, 1u);
            p.fb_list[slot] = (uint32_t)q;
        }
    }
}

// ------------------------------------------------------------------------------------
// 2c. D = 256: the decision for the (large) majority of queries that the margin test certifies, with the
//     per-query scalar arithmetic -- ranking of the 8 candidates of a segment, error bounds, square roots,
//     tanh -- done by ONE lane per query instead of redundantly by the 16 lanes that share a query in
//     phk_rerank16_kernel (that kernel is VALU-bound on exactly this redundancy: ~345 instructions per query).
//     A 256-thread block handles 256 queries in three phases:
//       A  lane = query : read its six half-lists, merge them, fetch column norms / labels of the leaders
//       B  16 lanes = query, 16 queries at a time: the row itself -- |q|^2, |q'|^2 and the exact float64
//          distances to the leading positive / negative centroid (exact_d2_g16)
//       C  lane = query : margin tests (the conditions of resolve_segment_g16), vote, proximity metric
//     A query some segment of which is not certified by margin goes to `slow_list`; phk_rerank16_kernel
//     (LISTED) then treats it exactly as before (exact candidate distances, fallback queue).
// ------------------------------------------------------------------------------------
template <int SRC>
__global__ __launch_bounds__(256) void phk_decide_kernel(const void *__restrict__ src, RerankParams p) {
    __shared__ uint32_t s_ix[2][256];
    __shared__ double s_T[256], s_nq2[256], s_nqp2[256], s_dp2[256], s_dn2[256], s_opQ[256], s_opI[256], s_opH[256];
    const int tid = threadIdx.x, lane = tid & 63, t = lane & 15;
    const int QB = blockDim.x;   // queries per block (64: one wave per block, no cross-wave waiting at the phase changes)
    const uint64_t qb = (uint64_t)blockIdx.x * QB;
    const bool want_knn = (p.method & PHK_METHOD_KNN) != 0, want_cen = (p.method & PHK_METHOD_KMEANS) != 0;

    // the count rows of phase B are fetched two passes ahead (explicit register double buffer: hipcc does not
    // software-pipeline that loop by itself and each pass would otherwise start with a full memory round trip);
    // the first two are requested here, ahead of phase A's list reads
    auto rowptr = [&](int pass) {
        const int ql = pass * (QB / 16) + (tid >> 4);
        const uint64_t q = qb + ql < p.N ? qb + ql : p.N - 1;
        return reinterpret_cast<const uint2 *>(static_cast<const uint32_t *>(src) + q * FAST_D) + t;
    };
    uint2 pre[2][8];
    if (SRC == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            pre[0][i] = rowptr(0)[16 * i];
            pre[1][i] = rowptr(1)[16 * i];
        }
    }

    // ---- phase A: one lane per query ----
    const uint64_t qa = qb + tid;
    const bool in_a = qa < p.N;
    const uint64_t qc = in_a ? qa : p.N - 1;
    float gap_hi[NSEG], gap_lo[NSEG], Useg[NSEG];   // need-th and (need+1)-th computed value, best dropped value
    uint32_t lead[NSEG][3];
    bool filled[NSEG];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        const uint32_t ncols = sg == 0 ? (uint32_t)p.M : sg == 1 ? (uint32_t)p.n_cpos : (uint32_t)p.n_cneg;
        const int need = sg == 0 ? p.kn : 1;
        Useg[sg] = fmaxf(p.cand_u[candu_at(sg, 0, qc, p.N)], p.cand_u[candu_at(sg, 1, qc, p.N)]);
        // the 4 best of the 8 candidates by insertion (descending; an equal value stays behind: a tie at the
        // decisive position fails the margin test anyway); padding / empty slots never enter
        float v[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        uint32_t ix[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        float w[8];
        uint32_t wx[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {   // consecutive lanes = consecutive queries: coalesced
            w[c] = p.cand_v[cand_at(sg, c >> 2, c & 3, qc, p.N)];
            wx[c] = p.cand_i[cand_at(sg, c >> 2, c & 3, qc, p.N)];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float x = wx[c] >= ncols ? -3.0e38f : w[c];
            uint32_t xi = wx[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool up = x > v[k];
                const float tv = v[k];
                const uint32_t ti = ix[k];
                v[k] = up ? x : tv;
                ix[k] = up ? xi : ti;
                x = up ? tv : x;
                xi = up ? ti : xi;
            }
        }
        gap_hi[sg] = need == 1 ? v[0] : need == 2 ? v[1] : v[2];
        gap_lo[sg] = need == 1 ? v[1] : need == 2 ? v[2] : v[3];
        filled[sg] = (need == 1 ? ix[0] : need == 2 ? ix[1] : ix[2]) < ncols;
#pragma unroll
        for (int r = 0; r < 3; ++r) lead[sg][r] = ix[r];
    }
    s_ix[0][tid] = lead[1][0] < (uint32_t)p.n_cpos ? lead[1][0] : 0u;
    s_ix[1][tid] = lead[2][0] < (uint32_t)p.n_cneg ? lead[2][0] : 0u;
    // speculative gathers, consumed in phase C
    double cn0[3];
    uint8_t lab0[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint32_t c = lead[0][r] < (uint32_t)p.M ? lead[0][r] : 0u;
        cn0[r] = p.colnorm[c];
        lab0[r] = p.labels[c];
    }
    const double cnp = p.colnorm[p.M + s_ix[0][tid]], cnn = p.colnorm[p.M + p.n_cpos + s_ix[1][tid]];
    __syncthreads();

    // ---- phase B: 16 lanes per query, 16 queries per pass ----
    {
        double mu[16];
        const double2 *mp = reinterpret_cast<const double2 *>(p.mu64) + t;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double2 m2 = mp[16 * i];
            mu[2 * i] = m2.x;
            mu[2 * i + 1] = m2.y;
        }
#pragma unroll 2
        for (int pass = 0; pass < 16; ++pass) {
            const int ql = pass * (QB / 16) + (tid >> 4);
            const uint64_t q = qb + ql < p.N ? qb + ql : p.N - 1;
            double qd[16], Tq = 1.0, invT2 = 1.0;
            bool bad = false;
            uint32_t cmx = 0, cmn = 0xFFFFFFFFu;
            if (SRC == 0) {
                uint2 cur[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) cur[i] = pre[pass & 1][i];
                if (pass + 2 < 16) {
                    const uint2 *nrow = rowptr(pass + 2);
#pragma unroll
                    for (int i = 0; i < 8; ++i) pre[pass & 1][i] = nrow[16 * i];
                }
                uint32_t sum = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint2 c = cur[i];
                    sum += c.x + c.y;
                    qd[2 * i + 0] = (double)c.x;
                    qd[2 * i + 1] = (double)c.y;
                    cmx = max(cmx, max(c.x, c.y));
                    cmn = min(cmn, min(c.x, c.y));
                }
                sum = group16_sum(sum);
                cmx = group16_max(cmx);
                cmn = group16_min(cmn);
                bad = sum == 0;
                Tq = (double)sum;
                invT2 = 1.0 / (Tq * Tq);
            } else {
                const double2 *row = reinterpret_cast<const double2 *>(static_cast<const double *>(src) + q * FAST_D) + t;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double2 v2 = row[16 * i];
                    qd[2 * i] = v2.x;
                    qd[2 * i + 1] = v2.y;
                    bad |= v2.x != v2.x || v2.y != v2.y;
                }
                unsigned bb = bad ? 1u : 0u;
                bb |= __shfl_xor(bb, 8); bb |= __shfl_xor(bb, 4); bb |= __shfl_xor(bb, 2); bb |= __shfl_xor(bb, 1);
                bad = bb != 0;
            }
            double aq = 0.0, ap = 0.0, am = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const double c0 = fma(-Tq, mu[i], qd[i]);
                aq = fma(qd[i], qd[i], aq);
                ap = fma(c0, c0, ap);
                am = fmax(am, fabs(c0));
            }
            const double sumsq = group16_sum(aq);
            const double nq2 = sumsq * invT2, nqp2 = group16_sum(ap) * invT2;
            // the query operand of the proposal's MFMA chain (see ErrBound): c - c0 for count-exact lists, q' otherwise
            double opQ = 0.0, opI = group16_max(am) / Tq, opH = 0.0;
            if (SRC == 0 && p.per_row_scale && !bad) {
                const CenteredOperand cop = phk_centered_operand(sumsq, Tq, (double)cmx, (double)cmn, (double)FAST_D, p.eb_hsum);
                opQ = cop.Q; opI = cop.I; opH = cop.habs;
            }
            double dp2 = 0.0, dn2 = 0.0;
            if (want_cen) {
                exact_d2_pair_g16(qd, Tq, invT2, p.C64 + (uint64_t)s_ix[0][ql] * FAST_D,
                                  p.C64 + (p.n_cpos + (uint64_t)s_ix[1][ql]) * FAST_D, t, dp2, dn2);
            }
            if (t == 0) {
                s_T[ql] = bad ? 0.0 : Tq;   // 0 marks a NaN row
                s_opQ[ql] = opQ;
                s_opI[ql] = opI;
                s_opH[ql] = opH;
                s_nq2[ql] = nq2;
                s_nqp2[ql] = nqp2;
                s_dp2[ql] = dp2;
                s_dn2[ql] = dn2;
            }
        }
    }
    __syncthreads();

    // ---- phase C: one lane per query ----
    if (!in_a) return;
    const double Tq = s_T[tid];
    if (Tq == 0.0) {  // zero-count contig / NaN input: the reference's normalised row is NaN
        p.scores[p.q_base + qa] = __builtin_nan("");
        if (p.status) atomicAdd(p.status, 1u);
        return;
    }
    const double nq2 = s_nq2[tid], nqp2 = s_nqp2[tid];
    const double vs = p.per_row_scale ? p.vscale / Tq : p.vscale;
    ErrBound eb;
    eb.A = sqrt(nq2) + p.mu_norm;
    eb.P = sqrt(nqp2);
    eb.Q = s_opQ[tid]; eb.I = s_opI[tid]; eb.habs = s_opH[tid];
    eb.cA = p.eb_cA; eb.cP = p.eb_cP; eb.cR = p.eb_cR; eb.cabs = p.eb_abs; eb.cQ = p.eb_cQ; eb.cI = p.eb_cI; eb.cIf = p.eb_cIf;
    const double nqp = eb.P;
    const double eps_g = eb(p.rmax);
    auto certify = [&](int sg, int need, const double *cnorms) {   // resolve_segment_g16's margin test
        if (!filled[sg]) return false;
        const double d2up = fmax(nqp2 - 2.0 * ((double)gap_hi[sg] * vs - eps_g), 0.0);
        const double R0 = fmin(p.rmax, (nqp + sqrt(d2up)) * (1.0 + 1e-6));
        bool near = true;
        for (int r = 0; r < need; ++r) near = near && cnorms[r] <= R0;
        const double eps_m = near ? eb(R0) : eps_g;
        return ((double)gap_hi[sg] - (double)gap_lo[sg]) * vs > 2.0 * eps_m;
    };
    bool cert = true;
    double knn = 0.0, cen = 0.0;
    if (want_knn) {
        cert = certify(0, p.kn, cn0);
        int votes = 0;
        for (int r = 0; r < p.kn; ++r) votes += lab0[r] ? 1 : 0;
        knn = (2 * votes > p.kn) ? 1.0 : -1.0;
    }
    if (want_cen) {
        cert = cert && certify(1, 1, &cnp) && certify(2, 1, &cnn);
        const double ep = sqrt(s_dp2[tid]), en = sqrt(s_dn2[tid]);
        cen = tanh((en - ep) / (ep + en));  // scripts/phamer.py:206-209
    }
    if (cert) {
        p.scores[p.q_base + qa] = knn + cen;  // scripts/phamer.py:313
    } else {
        p.slow_list[atomicAdd(p.fb_count + 2, 1u)] = (uint32_t)qa;
    }
}

// ------------------------------------------------------------------------------------
// 2c'. The same idea at general D for the lists of the two-part int8 sweep (round 4).  phk_rerank_kernel spends one wave per
//     query, and most of that wave's ~650 instructions are per-query scalar work that all 64 lanes repeat (ranking, the
//     error bound, the window and margin tests); a third of the step at configs[2].  Here a 64-thread block takes 64
//     queries through the k = 4 kernel's three phases:
//       A  lane = query : its six half-lists, the 4 best of each segment's 8 candidates, labels of the leaders
//       B  16 lanes = query, 4 queries per pass: the count row in 1024-dimension chunks (G16 ownership: every load
//          covers one contiguous 256 B piece) -- sum of squares, |q'|^2 against the training mean in LDS, and the exact
//          distances to the leading positive / negative centroid in the canonical form of exact_d2
//       C  lane = query : e_l, eps, the two-part window; a query is decided HERE when, in every segment the method uses,
//          the window holds exactly `need` columns and ends above everything the half-lists dropped -- the case
//          rerank_one_query decides "as it stands" (96 % of configs[2]); the margin test is implied:
//          gap_lo vs < thr = gap_hi vs - 2 (e_l + eps_g)  =>  (gap_hi - gap_lo) vs > 2 eps.
//     Everything else -- wider windows (they need the L product), windows past the lists, rows beyond the int8 operand,
//     NaN rows excepted -- goes to slow_list, and phk_rerank_kernel (listed) treats those queries exactly as before.
//     The row sum comes from p.rowsum (the count kernel's / the launcher's), so a chunk's c - T mu needs no second pass.
// ------------------------------------------------------------------------------------
template <int DSUB>
__global__ __launch_bounds__(64, 3) void phk_decide_gen_kernel(const uint32_t *__restrict__ counts, RerankParams p) {
    constexpr int D = 256 * DSUB;
    constexpr int CHUNK = D < 1024 ? D : 1024;     // dimensions per chunk of phase B
    constexpr int NCH = D / CHUNK;
    constexpr int LPC = CHUNK / 64;                // uint4 loads per lane and chunk (4 dimensions each)
    static_assert(LPC % 4 == 0, "loads in groups of four");
    __shared__ double s_mu[D];
    __shared__ uint32_t s_ix[2][64];
    __shared__ double s_ssq[64], s_nqp2[64], s_dp2[64], s_dn2[64];
    __shared__ float s_gap[NSEG][3][64];   // phase A -> C: need-th / (need+1)-th list value, best dropped value (not kept in registers across phase B)
    __shared__ uint32_t s_flag[64];        // ... bits 0-2: segment filled, bits 4-6: labels of the three leading train columns
    const int tid = threadIdx.x, t = tid & 15, grp = tid >> 4;
    const uint64_t qb = (uint64_t)blockIdx.x * 64;
    const bool want_knn = (p.method & PHK_METHOD_KNN) != 0, want_cen = (p.method & PHK_METHOD_KMEANS) != 0;
    for (int i = tid; i < D / 2; i += 64) reinterpret_cast<double2 *>(s_mu)[i] = reinterpret_cast<const double2 *>(p.mu64)[i];

    // ---- phase A: one lane per query ----
    const uint64_t qa = qb + tid;
    const bool in_a = qa < p.N;
    const uint64_t qc = in_a ? qa : p.N - 1;
    float gap_hi[NSEG], gap_lo[NSEG], Useg[NSEG];   // need-th and (need+1)-th list value, best dropped value
    uint32_t lead[NSEG][3];
    bool filled[NSEG];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        const uint32_t ncols = sg == 0 ? (uint32_t)p.M : sg == 1 ? (uint32_t)p.n_cpos : (uint32_t)p.n_cneg;
        const int need = sg == 0 ? p.kn : 1;
        Useg[sg] = fmaxf(p.cand_u[candu_at(sg, 0, qc, p.N)], p.cand_u[candu_at(sg, 1, qc, p.N)]);
        float v[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        uint32_t ix[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        float w[8];
        uint32_t wx[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {   // consecutive lanes = consecutive queries: coalesced
            w[c] = p.cand_v[cand_at(sg, c >> 2, c & 3, qc, p.N)];
            wx[c] = p.cand_i[cand_at(sg, c >> 2, c & 3, qc, p.N)];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {   // the 4 best by insertion (descending; an equal value stays behind)
            float x = wx[c] >= ncols ? -3.0e38f : w[c];
            uint32_t xi = wx[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool up = x > v[k];
                const float tv = v[k];
                const uint32_t ti = ix[k];
                v[k] = up ? x : tv;
                ix[k] = up ? xi : ti;
                x = up ? tv : x;
                xi = up ? ti : xi;
            }
        }
        gap_hi[sg] = need == 1 ? v[0] : need == 2 ? v[1] : v[2];
        gap_lo[sg] = need == 1 ? v[1] : need == 2 ? v[2] : v[3];
        filled[sg] = (need == 1 ? ix[0] : need == 2 ? ix[1] : ix[2]) < ncols;
#pragma unroll
        for (int r = 0; r < 3; ++r) lead[sg][r] = ix[r];
    }
    s_ix[0][tid] = lead[1][0] < (uint32_t)p.n_cpos ? lead[1][0] : 0u;
    s_ix[1][tid] = lead[2][0] < (uint32_t)p.n_cneg ? lead[2][0] : 0u;
    {
        uint32_t fl = 0;
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
            s_gap[sg][0][tid] = gap_hi[sg];
            s_gap[sg][1][tid] = gap_lo[sg];
            s_gap[sg][2][tid] = Useg[sg];
            fl |= filled[sg] ? 1u << sg : 0u;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) fl |= p.labels[lead[0][r] < (uint32_t)p.M ? lead[0][r] : 0u] ? 16u << r : 0u;
        s_flag[tid] = fl;
    }
    __syncthreads();

    // ---- phase B: 16 lanes per query, 4 queries per pass ----
#pragma unroll 1
    for (int pass = 0; pass < 16; ++pass) {
        const int ql = pass * 4 + grp;
        const uint64_t q = qb + ql < p.N ? qb + ql : p.N - 1;
        const double Tq = (double)p.rowsum[q];
        const uint4 *row = reinterpret_cast<const uint4 *>(counts + q * D) + t;
        const double *cp = p.C64 + (uint64_t)s_ix[0][ql] * D, *cn = p.C64 + (p.n_cpos + (uint64_t)s_ix[1][ql]) * D;
        // The exact distances are formed EXACTLY as exact_d2 forms them in a 64-lane wave, so that a score does not depend on
        // which kernel decided it: that wave's lane 16 g + t owns dimensions 256 sub + 64 g + 4 t .. + 3 -- here load i of a
        // chunk, g = i & 3, sub = 4 ch + (i >> 2) -- and accumulates them over sub with the same nesting; its reduction is
        // the row sums of the four lane groups (the same DPP rotations over the same 16 positions), then (G0 + G1) + (G2 + G3).
        double ssq = 0.0, ap = 0.0, dp[4] = {0.0, 0.0, 0.0, 0.0}, dn[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll 1
            for (int io = 0; io < LPC / 4; ++io) {   // four loads at a time (one per lane group of the canonical form): ~80 registers in flight
            uint4 c[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = row[ch * (CHUNK / 4) + 16 * (4 * io + i)];   // dimensions CHUNK ch + 64 (4 io + i) + 4 t .. + 3
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d0 = ch * CHUNK + 64 * (4 * io + i) + 4 * t;
                const double x[4] = {(double)c[i].x, (double)c[i].y, (double)c[i].z, (double)c[i].w};
                const double2 m0 = *reinterpret_cast<const double2 *>(s_mu + d0), m1 = *reinterpret_cast<const double2 *>(s_mu + d0 + 2);
                const double mu4[4] = {m0.x, m0.y, m1.x, m1.y};
                double2 a0 = {0.0, 0.0}, a1 = {0.0, 0.0}, b0 = {0.0, 0.0}, b1 = {0.0, 0.0};
                if (want_cen) {
                    a0 = *reinterpret_cast<const double2 *>(cp + d0); a1 = *reinterpret_cast<const double2 *>(cp + d0 + 2);
                    b0 = *reinterpret_cast<const double2 *>(cn + d0); b1 = *reinterpret_cast<const double2 *>(cn + d0 + 2);
                }
                const double ca[4] = {a0.x, a0.y, a1.x, a1.y}, cb[4] = {b0.x, b0.y, b1.x, b1.y};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ssq = fma(x[e], x[e], ssq);
                    const double qm = fma(-Tq, mu4[e], x[e]);
                    ap = fma(qm, qm, ap);
                }
                if (want_cen) {
                    const double e0 = fma(-Tq, ca[0], x[0]), e1 = fma(-Tq, ca[1], x[1]), e2 = fma(-Tq, ca[2], x[2]), e3 = fma(-Tq, ca[3], x[3]);
                    dp[i & 3] = fma(e0, e0, fma(e1, e1, fma(e2, e2, fma(e3, e3, dp[i & 3]))));
                    const double f0 = fma(-Tq, cb[0], x[0]), f1 = fma(-Tq, cb[1], x[1]), f2 = fma(-Tq, cb[2], x[2]), f3 = fma(-Tq, cb[3], x[3]);
                    dn[i & 3] = fma(f0, f0, fma(f1, f1, fma(f2, f2, fma(f3, f3, dn[i & 3]))));
                }
            }
            }
        }
        ssq = group16_sum(ssq);
        ap = group16_sum(ap);
        double dps = 0.0, dns = 0.0;
        if (want_cen) {
            dps = (group16_sum(dp[0]) + group16_sum(dp[1])) + (group16_sum(dp[2]) + group16_sum(dp[3]));
            dns = (group16_sum(dn[0]) + group16_sum(dn[1])) + (group16_sum(dn[2]) + group16_sum(dn[3]));
        }
        if (t == 0) {
            s_ssq[ql] = ssq;
            s_nqp2[ql] = ap;      // (x T^2)
            s_dp2[ql] = dps;      // (x T^2)
            s_dn2[ql] = dns;
        }
    }
    __syncthreads();

    // ---- phase C: one lane per query ----
    bool slow = false;
    if (in_a) {
        const uint32_t Tu = p.rowsum[qa];
        const uint32_t fl = s_flag[tid];
        if (Tu == 0u) {  // zero-count contig: the reference's normalised row is NaN
            p.scores[p.q_base + qa] = __builtin_nan("");
            if (p.status) atomicAdd(p.status, 1u);
        } else {
            const double Tq = (double)Tu, ry = 1.0 / Tq, invT2 = 1.0 / (Tq * Tq);
            const double ssq = s_ssq[tid], nqp2 = s_nqp2[tid] * invT2;
            const double vs = p.vscale * ry;
            const double rcen = (double)phk_row_center(Tu, D);
            const CenteredOperand cop = phk_centered_operand_fast(ssq, Tq, ry, rcen, rcen, (double)D, p.eb_hsum);
            ErrBound eb;
            eb.A = phk_sqrt_up(ssq * (ry * ry) * (1.0 + 1e-12)) + p.mu_norm;
            eb.P = phk_sqrt_up(nqp2 + cop.shift2);
            eb.Q = cop.Q; eb.I = cop.I; eb.habs = cop.habs;
            eb.cA = p.eb_cA; eb.cP = p.eb_cP; eb.cR = p.eb_cR; eb.cabs = p.eb_abs; eb.cQ = p.eb_cQ; eb.cI = p.eb_cI; eb.cIf = p.eb_cIf;
            const double eps_g = eb(p.rmax);
            // a segment is decided as it stands: the window [h_need - 2 (e_l + eps), ..] holds exactly `need` list values
            // and ends above everything the half-lists dropped
            auto as_it_stands = [&](int sg) {
                if (!((fl >> sg) & 1u)) return false;
                const double el = cop.Q * p.lam8[sg] * (1.0 + 1.0e-6);
                const double thr = (double)s_gap[sg][0][tid] * vs - 2.0 * (el + eps_g);
                return (double)s_gap[sg][2][tid] * vs < thr && (double)s_gap[sg][1][tid] * vs < thr;
            };
            bool ok = true;
            double knn = 0.0;
            if (want_knn) {
                ok = as_it_stands(0);
                int votes = 0;
                for (int r = 0; r < p.kn; ++r) votes += (fl >> (4 + r)) & 1u;
                knn = (2 * votes > p.kn) ? 1.0 : -1.0;
            }
            if (want_cen) ok = ok && as_it_stands(1) && as_it_stands(2);
            if (ok) {
                if (want_cen) {
                    const uint64_t oq = p.out_map ? (uint64_t)p.out_map[qa] : qa;
                    if (p.pend) {   // (phk_finish_cen_kernel turns them into the proximity metric)
                        p.pend[2 * oq] = s_dp2[tid] * invT2;
                        p.pend[2 * oq + 1] = s_dn2[tid] * invT2;
                        p.scores[p.q_base + oq] = knn;
                    } else {
                        const double ep = sqrt(s_dp2[tid] * invT2), en = sqrt(s_dn2[tid] * invT2);
                        p.scores[p.q_base + oq] = knn + tanh((en - ep) / (ep + en));  // scripts/phamer.py:206-209, 313
                    }
                } else {
                    p.scores[p.q_base + (p.out_map ? (uint64_t)p.out_map[qa] : qa)] = knn;
                }
            } else {
                slow = true;
            }
        }
    }
    // hand-over: one atomic per wave, not per query (same-line atomics retire one after the other)
    const unsigned long long sm = __ballot(slow);
    if (sm) {
        uint32_t base = 0;
        if (tid == 0) base = atomicAdd(p.fb_count + 2, (uint32_t)__popcll(sm));
        base = __shfl(base, 0);
        if (slow) p.slow_list[base + (uint32_t)__popcll(sm & ((1ull << tid) - 1ull))] = (uint32_t)qa;
    }
}

// ------------------------------------------------------------------------------------
// 2d. decision stage of the high-parts-only proposal (phk_knn_f16h_kernel; D = 256, uint32 counts).
//     The lists hold HIGH-PART values  w^h_j  whose distance from the count-exact value  w_j  is the low product
//         w_j - w^h_j = sum_i (c_i - T mu_i) lo_ji,      |.| <= T S |q'| lam_j,   lam_j = |lo_j| / S   (Cauchy-Schwarz).
//     With  e_h = |q'| lam*(R0) + e22  (lam* = the largest lam_j among the columns within reach R0, HiParams.lam_tab;
//     e22 = the count-exact error model, which also covers this pass's fewer MFMA roundings):
//       window   every column whose true value can be among the `need` best has  w^h >= h_need - 2 e_h,  h_need = the
//                need-th best high-part value.  Columns in the window must all be list members: the best value either
//                half-list dropped has to be below the window, else the query takes the second chance.
//       refine   the window's members (3 to 8 columns, typically 3 or 4) get the low product -- float32 v_fma_mix on the
//                float64-centred counts, 16 lanes per query, from the low parts stored in G16 order (512 B per column);
//                its 20 roundings are bounded inside the test (2^-19 |q'| lam*).  They then carry count-exact
//                values and the count-exact margin test decides their order exactly as phk_decide_kernel does.
//       centroid segments (need = 1): the leader is certified by its high-part margin (h_1 - h_2 > 2 e_h; both are list
//                members: each half-list keeps its 4 best) and its exact float64 distance is computed as before; a
//                leader that is not certified sends the query to the second chance.
//     Phases as in phk_decide_kernel: A one lane per query (lists, window), B 16 lanes per query (row norms, low
//     products, exact centroid distances), C one lane per query (margin tests, vote, metric).
// ------------------------------------------------------------------------------------
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

template <bool KNN, bool CEN>
__global__ __launch_bounds__(64) void phk_decide_h_kernel(const uint32_t *__restrict__ counts, RerankParams p, HiParams hp) {
    __shared__ uint32_t s_c0[8][64];        // train-segment candidates by descending high-part value
    __shared__ uint32_t s_ix[2][64];        // centroid-segment leaders
    __shared__ float s_corr[PHK_HI_REFINE][64];
    // phase A's per-query results wait in LDS while phase B (the register-hungry part) runs
    __shared__ float s_v8[8][64], s_u0[64], s_ch[4][64];
    __shared__ double s_cn[5][64];
    __shared__ uint32_t s_flags[64];
    // phase B leaves the row's RAW sums here; everything one lane per query can finish -- the division by T^2, the centred
    // operand's norms (a float64 square root and three divisions) -- is phase C's: in phase B the 16 lanes of a query, 64 lanes of
    // a wave, each repeated it in every pass (a fifth of that phase's instructions, and the kernel is bound by its own
    // instruction stream at two waves per SIMD)
    __shared__ double s_T[64], s_sumsq[64], s_apsum[64], s_dp2[64], s_dn2[64];
    __shared__ uint32_t s_cmx[64], s_cmn[64];
    __shared__ double s_mu[FAST_D];         // the training mean (LDS reads keep vmcnt for the row / column loads)
    const int tid = threadIdx.x, t = tid & 15;
    const uint64_t qb = (uint64_t)blockIdx.x * 64;
    constexpr bool want_knn = KNN, want_cen = CEN;
    // Every load of this kernel is issued in batches that do not depend on each other, with a scheduling barrier between
    // a batch and its first use: left alone, the compiler trades loads in flight for registers and emits load, wait,
    // use, load, wait, ... (24 exposed round trips in phase A and 16 per pass in phase B, by the ISA), and a load under
    // a run-time condition makes every later wait conservative (vmcnt completes in order) -- hence the template
    // parameters instead of `method` tests.
    auto rowptr = [&](int pass) {
        const int ql = pass * 4 + (tid >> 4);
        const uint64_t q = qb + ql < p.N ? qb + ql : p.N - 1;
        return reinterpret_cast<const uint2 *>(counts + q * FAST_D) + t;
    };
    uint2 cur[8];   // the count row of the pass at hand; the next one is requested a whole pass ahead
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = rowptr(0)[16 * i];
    {
        const double2 m2a = reinterpret_cast<const double2 *>(p.mu64)[2 * tid], m2b = reinterpret_cast<const double2 *>(p.mu64)[2 * tid + 1];
        reinterpret_cast<double2 *>(s_mu)[2 * tid] = m2a;
        reinterpret_cast<double2 *>(s_mu)[2 * tid + 1] = m2b;
    }

    // ---- phase A: one lane per query ----
    const uint64_t qa = qb + tid;
    const bool in_a = qa < p.N;
    const uint64_t qc = in_a ? qa : p.N - 1;
    float v8[8];
    uint32_t i8[8];
    float U0 = 0.f;
    bool ok0 = true;          // the need-th list position holds a real column
    float ch1[2] = {0.f, 0.f}, ch2[2] = {0.f, 0.f};
    uint32_t cl[2] = {0u, 0u};
    bool cfill[2] = {true, true};
    {
        float lv[NSEG][8], lu[2] = {0.f, 0.f};
        uint32_t li[NSEG][8];
        if (KNN) {
            lu[0] = p.cand_u[candu_at(0, 0, qc, p.N)];
            lu[1] = p.cand_u[candu_at(0, 1, qc, p.N)];
        }
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg)
            if (sg == 0 ? KNN : CEN) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    li[sg][c] = p.cand_i[cand_at(sg, c >> 2, c & 3, qc, p.N)];
                    lv[sg][c] = p.cand_v[cand_at(sg, c >> 2, c & 3, qc, p.N)];
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        // train segment: the 8 candidates sorted by high-part value (descending; empty / padding slots last)
        if (KNN) {
            U0 = fmaxf(lu[0], lu[1]);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                v8[c] = li[0][c] >= (uint32_t)p.M ? -3.0e38f : lv[0][c];
                i8[c] = li[0][c];
            }
#pragma unroll
            for (int a = 1; a < 8; ++a)          // insertion sort network, fully unrolled (descending)
#pragma unroll
                for (int b = a; b > 0; --b) {
                    const bool sw = v8[b] > v8[b - 1];
                    const float tv = v8[b]; const uint32_t ti = i8[b];
                    v8[b] = sw ? v8[b - 1] : v8[b]; i8[b] = sw ? i8[b - 1] : i8[b];
                    v8[b - 1] = sw ? tv : v8[b - 1]; i8[b - 1] = sw ? ti : i8[b - 1];
                }
        }
        // centroid segments: leader and runner-up by high-part value
        if (CEN) {
#pragma unroll
            for (int sg = 1; sg <= 2; ++sg) {
                const uint32_t ncols = sg == 1 ? (uint32_t)p.n_cpos : (uint32_t)p.n_cneg;
                float b1 = -3.0e38f, b2 = -3.0e38f;
                uint32_t bi = 0xFFFFFFFFu;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t ix = li[sg][c];
                    const float w = ix >= ncols ? -3.0e38f : lv[sg][c];
                    const bool up = w > b1;
                    b2 = up ? b1 : fmaxf(b2, w);
                    bi = up ? ix : bi;
                    b1 = up ? w : b1;
                }
                ch1[sg - 1] = b1; ch2[sg - 1] = b2; cl[sg - 1] = bi;
                cfill[sg - 1] = bi < ncols;
            }
        }
    }
    const uint32_t ixp = cl[0] < (uint32_t)p.n_cpos ? cl[0] : 0u, ixn = cl[1] < (uint32_t)p.n_cneg ? cl[1] : 0u;
    s_ix[0][tid] = ixp;
    s_ix[1][tid] = ixn;
    {
        // gathers consumed in phase C: one batch
        double cn0[3];
        uint8_t lab[8];
        uint32_t c8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) c8[r] = (KNN && i8[r] < (uint32_t)p.M) ? i8[r] : 0u;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < 3) cn0[r] = p.colnorm[c8[r]];
            lab[r] = p.labels[c8[r]];
        }
        const double cnp = p.colnorm[p.M + ixp], cnn = p.colnorm[p.M + p.n_cpos + ixn];
        __builtin_amdgcn_sched_barrier(0);
        uint32_t labbits = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            labbits |= (uint32_t)(lab[r] ? 1u : 0u) << r;
            s_c0[r][tid] = c8[r];
            s_v8[r][tid] = (KNN && i8[r] < (uint32_t)p.M) ? v8[r] : -3.0e38f;
        }
        if (KNN) ok0 = i8[p.kn - 1] < (uint32_t)p.M;
        s_u0[tid] = U0;
        s_ch[0][tid] = ch1[0]; s_ch[1][tid] = ch2[0]; s_ch[2][tid] = ch1[1]; s_ch[3][tid] = ch2[1];
        s_cn[0][tid] = cn0[0]; s_cn[1][tid] = cn0[1]; s_cn[2][tid] = cn0[2]; s_cn[3][tid] = cnp; s_cn[4][tid] = cnn;
        s_flags[tid] = labbits | (ok0 ? 0x100u : 0u) | (cfill[0] ? 0x200u : 0u) | (cfill[1] ? 0x400u : 0u);
    }
    __syncthreads();

    // ---- phase B: 16 lanes per query, 4 queries per pass ----
    // A pass issues all its column loads -- the two centroid rows (float64) and the low parts of the first
    // PHK_HI_REFINE train candidates, whose addresses depend on the lists only -- and, youngest, the count row of the
    // NEXT pass; then it reduces its own row (in registers since the pass before).  One exposed L2 round trip per pass,
    // and no wait ever covers the count row's HBM miss.  The low products are fetched for all PHK_HI_REFINE candidates
    // (a window-sized fetch, tried, has to wait for the row first and cost 60 % more time); phase C, which knows the
    // window, only uses the members.
#pragma unroll 2
    for (int pass = 0; pass < 16; ++pass) {
        const int ql = pass * 4 + (tid >> 4);
        double2 ca[8], cb[8];
        uint4 l0[PHK_HI_REFINE], l1[PHK_HI_REFINE];
        uint2 nxt[8];
        if (CEN) {
            const double2 *ra = reinterpret_cast<const double2 *>(p.C64 + (uint64_t)s_ix[0][ql] * FAST_D) + t;
            const double2 *rb = reinterpret_cast<const double2 *>(p.C64 + (p.n_cpos + (uint64_t)s_ix[1][ql]) * FAST_D) + t;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                ca[i] = ra[16 * i];
                cb[i] = rb[16 * i];
            }
        }
        if (KNN) {
#pragma unroll
            for (int r = 0; r < PHK_HI_REFINE; ++r) {
                const uint4 *lp = reinterpret_cast<const uint4 *>(hp.lo16 + (uint64_t)s_c0[r][ql] * FAST_D) + t;   // G16-ordered rows
                l0[r] = lp[0];
                l1[r] = lp[16];
            }
        }
        {
            const uint2 *nrow = rowptr((pass + 1) & 15);   // (the last pass re-reads row 0: a load under a condition would
#pragma unroll                                             //  make the waits below conservative)
            for (int i = 0; i < 8; ++i) nxt[i] = nrow[16 * i];
        }
        __builtin_amdgcn_sched_barrier(0);
        double qd[16];
        uint32_t sum = 0, cmx = 0, cmn = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint2 c = cur[i];
            sum += c.x + c.y;
            cmx = max(cmx, max(c.x, c.y));
            cmn = min(cmn, min(c.x, c.y));
            qd[2 * i + 0] = (double)c.x;
            qd[2 * i + 1] = (double)c.y;
        }
        sum = group16_sum(sum);
        cmx = group16_max(cmx);
        cmn = group16_min(cmn);
        const bool bad = sum == 0;
        const double Tq = (double)sum;
        double dp2 = 0.0, dn2 = 0.0;   // (the sums; phase C multiplies by 1 / T^2)
        if (CEN) {   // exact float64 distances to the two leading centroids (as exact_d2_g16)
            double acca = 0.0, accb = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double a0 = fma(-Tq, ca[i].x, qd[2 * i]), a1 = fma(-Tq, ca[i].y, qd[2 * i + 1]);
                const double b0 = fma(-Tq, cb[i].x, qd[2 * i]), b1 = fma(-Tq, cb[i].y, qd[2 * i + 1]);
                acca = fma(a0, a0, fma(a1, a1, acca));
                accb = fma(b0, b0, fma(b1, b1, accb));
            }
            dp2 = group16_sum(acca);
            dn2 = group16_sum(accb);
        }
        double aq = 0.0, ap = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {   // qd becomes c - T mu (the centred counts) in place
            const double2 m2 = reinterpret_cast<const double2 *>(s_mu)[16 * i + t];
            aq = fma(qd[2 * i], qd[2 * i], fma(qd[2 * i + 1], qd[2 * i + 1], aq));
            qd[2 * i] = fma(-Tq, m2.x, qd[2 * i]);
            qd[2 * i + 1] = fma(-Tq, m2.y, qd[2 * i + 1]);
            ap = fma(qd[2 * i], qd[2 * i], fma(qd[2 * i + 1], qd[2 * i + 1], ap));
        }
        const double sumsq = group16_sum(aq), apsum = group16_sum(ap);
        // The proposal kernel's query operand is c - c0 (phk_row_center), so its value is the high product of the
        // UNcentred counts minus (c0 - T/D) sum_i hi_ji; the low product that completes it is therefore taken with
        // c - T mu - (c0 - T/D): sum_i (c_i - T mu_i - dlt) lo_ji = sum_i (c_i - T mu_i) lo_ji + dlt sum_i hi_ji - dlt sum_i r~'_ji,
        // the last term being the model's hsum residue (see ErrBound).
        if (KNN) {
            // float32 products (v_fma_mix takes the half operand as it is): 16 + 4 roundings per sum, bounded in
            // phase C by 2^-19 |x| lam* -- 1e-6 of the low product's own bound
            const double dlt = (double)phk_row_center(sum, FAST_D) - Tq * (1.0 / FAST_D);
            float qf[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) qf[i] = (float)(qd[i] - dlt);
#pragma unroll
            for (int r = 0; r < PHK_HI_REFINE; ++r) {
                const _Float16 *lh0 = reinterpret_cast<const _Float16 *>(&l0[r]), *lh1 = reinterpret_cast<const _Float16 *>(&l1[r]);
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc = fmaf(qf[i], (float)lh0[i], acc);
                    acc = fmaf(qf[8 + i], (float)lh1[i], acc);
                }
                acc = group16_sum(acc);
                if (t == 0) s_corr[r][ql] = acc;
            }
        }
        if (t == 0) {
            s_T[ql] = bad ? 0.0 : Tq;   // 0 marks a NaN row
            s_sumsq[ql] = sumsq;
            s_apsum[ql] = apsum;
            s_dp2[ql] = dp2;
            s_dn2[ql] = dn2;
            s_cmx[ql] = cmx;
            s_cmn[ql] = cmn;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
    }
    __syncthreads();

    // ---- phase C: one lane per query ----
    if (!in_a) return;
    const double Tq = s_T[tid];
    if (Tq == 0.0) {  // zero-count contig: the reference's normalised row is NaN
        p.scores[p.q_base + qa] = __builtin_nan("");
        if (p.status) atomicAdd(p.status, 1u);
        return;
    }
    const double invT2 = 1.0 / (Tq * Tq);
    const double nq2 = s_sumsq[tid] * invT2, nqp2 = s_apsum[tid] * invT2;
    const CenteredOperand cop = phk_centered_operand(s_sumsq[tid], Tq, (double)s_cmx[tid], (double)s_cmn[tid], (double)FAST_D, p.eb_hsum);
    const double vs = p.vscale / Tq;
    double cn0[3];
    uint32_t labbits;
    {   // phase A's results back from LDS
#pragma unroll
        for (int r = 0; r < 8; ++r) v8[r] = s_v8[r][tid];
        U0 = s_u0[tid];
        ch1[0] = s_ch[0][tid]; ch2[0] = s_ch[1][tid]; ch1[1] = s_ch[2][tid]; ch2[1] = s_ch[3][tid];
        cn0[0] = s_cn[0][tid]; cn0[1] = s_cn[1][tid]; cn0[2] = s_cn[2][tid];
        const uint32_t fl = s_flags[tid];
        labbits = fl & 0xFFu;
        ok0 = (fl & 0x100u) != 0;
        cfill[0] = (fl & 0x200u) != 0;
        cfill[1] = (fl & 0x400u) != 0;
    }
    const double cnp2 = s_cn[3][tid], cnn2 = s_cn[4][tid];
    ErrBound eb;
    eb.A = sqrt(nq2) + p.mu_norm;
    eb.P = sqrt(nqp2);
    eb.Q = cop.Q; eb.I = cop.I; eb.habs = cop.habs + p.eb_babs;
    eb.cA = p.eb_cA; eb.cP = p.eb_cP; eb.cR = p.eb_cR; eb.cabs = p.eb_abs; eb.cQ = p.eb_cQ; eb.cI = p.eb_cI; eb.cIf = p.eb_cIf;
    eb.cM = p.eb_cM; eb.M = p.eb_M;
    const double nqp = eb.P;
    const double nqx = sqrt(nqp2 + cop.shift2);   // |q' - (c0/T - 1/D) 1|: what the low parts multiply (see phase B)
    auto e_hi = [&](int sg, double R) { return nqx * phk_lam_of(hp, sg, R) + eb(R); };
    bool cert = true;      // the k-NN part
    bool cert_c = true;    // the centroid part
    double knn = 0.0, cen = 0.0;
    if (want_knn) {
        const int need = p.kn;
        cert = ok0;
        if (cert) {
            // reach of the need nearest columns from the need-th high-part value; error bounds at that reach when the
            // leaders lie within it
            const double eg = e_hi(0, p.rmax);
            const double d2up = fmax(nqp2 - 2.0 * ((double)v8[need - 1] * vs - eg), 0.0);
            const double R0 = fmin(p.rmax, (nqp + sqrt(d2up)) * (1.0 + 1e-6));
            bool near = true;
            for (int r = 0; r < need; ++r) near = near && cn0[r] <= R0;
            const double Rw = near ? R0 : p.rmax;
            // count-exact error model + the float32 rounding of the low products (22 roundings x 2^-24 < 2^-19)
            const double eh = near ? e_hi(0, R0) : eg, e22 = eb(Rw) + 0x1p-19 * nqx * phk_lam_of(hp, 0, Rw);
            // window members: list positions 0 .. nw-1 (sorted by high-part value)
            const double thr = (double)v8[need - 1] * vs - 2.0 * eh;
            int nw = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) nw += ((double)v8[r] * vs >= thr && v8[r] > -1.0e38f) ? 1 : 0;
            // every column of the window has to be a list member with a refined value
            cert = nw <= PHK_HI_REFINE && (double)U0 * vs < thr;
            // diagnostics: window too wide / reaches past the lists (one uniform address per statement, so that the
            // compiler folds a wave's increments into one atomic; a per-lane address costs ~10 ns per lane)
            if (!cert && nw > PHK_HI_REFINE) atomicAdd(phk_stat_word(p, p.counters + 8, 8), 1u);
            if (!cert && nw <= PHK_HI_REFINE) atomicAdd(phk_stat_word(p, p.counters + 9, 9), 1u);
            if (cert) {
                // refined values of the window's members, descending
                double rv[PHK_HI_REFINE];
                uint32_t rl[PHK_HI_REFINE];
#pragma unroll
                for (int r = 0; r < PHK_HI_REFINE; ++r) {
                    const bool in = r < nw;
                    rv[r] = in ? ((double)v8[r] + (double)s_corr[r][tid]) * vs : -1.0e300;
                    rl[r] = (labbits >> r) & 1u;
                }
#pragma unroll
                for (int a = 1; a < PHK_HI_REFINE; ++a)
#pragma unroll
                    for (int b = a; b > 0; --b) {
                        const bool sw = rv[b] > rv[b - 1];
                        const double tv = rv[b]; const uint32_t tl = rl[b];
                        rv[b] = sw ? rv[b - 1] : rv[b]; rl[b] = sw ? rl[b - 1] : rl[b];
                        rv[b - 1] = sw ? tv : rv[b - 1]; rl[b - 1] = sw ? tl : rl[b - 1];
                    }
                // the need-th and (need+1)-th refined values decide (a window of exactly `need` members is decided)
                const double hi_v = need == 1 ? rv[0] : need == 2 ? rv[1] : rv[2];
                const double lo_v = need == 1 ? rv[1] : need == 2 ? rv[2] : rv[3];
                cert = nw == need || hi_v - lo_v > 2.0 * e22;
                if (!cert) atomicAdd(phk_stat_word(p, p.counters + 10, 10), 1u);   // diagnostics: refined values too close
                int votes = 0;
                for (int r = 0; r < need; ++r) votes += (int)rl[r];
                knn = (2 * votes > need) ? 1.0 : -1.0;
            }
        }
    }
    if (want_cen) {
        auto leader_ok = [&](int k2, double cnorm) {
            if (!cfill[k2]) return false;
            const double eg = e_hi(1 + k2, p.rmax);
            const double d2up = fmax(nqp2 - 2.0 * ((double)ch1[k2] * vs - eg), 0.0);
            const double R0 = fmin(p.rmax, (nqp + sqrt(d2up)) * (1.0 + 1e-6));
            const double eh = cnorm <= R0 ? e_hi(1 + k2, R0) : eg;
            return ((double)ch1[k2] - (double)ch2[k2]) * vs > 2.0 * eh;
        };
        cert_c = leader_ok(0, cnp2) && leader_ok(1, cnn2);
        if (!cert_c) atomicAdd(phk_stat_word(p, p.counters + 11, 11), 1u);   // diagnostics: centroid leader not certified
        const double ep = sqrt(s_dp2[tid] * invT2), en = sqrt(s_dn2[tid] * invT2);   // (the squared distances as exact_d2_g16 forms them)
        cen = tanh((en - ep) / (ep + en));  // scripts/phamer.py:206-209
    }
    if (cert && cert_c) {
        p.scores[p.q_base + qa] = knn + cen;  // scripts/phamer.py:313
    } else {
        // the open part(s) go to phk_rerank16_kernel (MODE 1), which takes the query's lists as they are -- high-part
        // values under the high-part error model -- and decides by exact float64 candidate distances where that
        // suffices; the decided part waits in scores[q]  (entry = query | open parts << 30)
        p.scores[p.q_base + qa] = (cert ? knn : 0.0) + (cert_c ? cen : 0.0);
        const uint32_t open_parts = ((want_knn && !cert) ? 1u : 0u) | ((want_cen && !cert_c) ? 2u : 0u);
        // two lists in one array, so that a wave of the next kernel works on one kind of segment: queries with the k-NN
        // part open from the front, those with only the centroid part open from the back
        if (p.sub_lists) {   // this workgroup's pair of lists (see RerankParams::sub_lists)
            const uint32_t sl = blockIdx.x % p.sub_lists;
            uint32_t *cw = p.stripes + sl * 32u;
            const uint64_t base = (uint64_t)sl * p.sub_cap;
            if (open_parts == 2u) p.slow_list[base + p.sub_cap - 1 - atomicAdd(cw + 12, 1u)] = (uint32_t)qa | (open_parts << 30);
            else p.slow_list[base + atomicAdd(cw + 2, 1u)] = (uint32_t)qa | (open_parts << 30);
        } else if (open_parts == 2u) p.slow_list[p.slow_cap - 1 - atomicAdd(p.counters + 12, 1u)] = (uint32_t)qa | (open_parts << 30);
        else p.slow_list[atomicAdd(p.fb_count + 2, 1u)] = (uint32_t)qa | (open_parts << 30);
    }
}

// ------------------------------------------------------------------------------------
// 2e. the same decision for general D (k = 5, 6; phk_knn_f16_general_kernel with HI): one wave per query, as
//     phk_rerank_kernel.  Window, low products (float64, 4 DSUB dimensions per lane, 2 D bytes of lo16 per member),
//     count-exact margin test among the refined members; centroid leaders certified by their high-part margin and
//     given their exact distance.  A query it cannot decide goes to phk_rerank_kernel (listed), which works from the
//     same lists under the high-part error model with exact candidate distances.
// ------------------------------------------------------------------------------------
template <int DSUB>
__global__ __launch_bounds__(256) void phk_rerank_h_kernel(const uint32_t *__restrict__ counts, RerankParams p, HiParams hp) {
    constexpr int D = 256 * DSUB;
    const int lane = threadIdx.x & 63;
    const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (q >= p.N) return;
    const uint32_t *row = counts + q * D;
    uint4 c[DSUB];
    uint32_t s = 0;
#pragma unroll
    for (int sub = 0; sub < DSUB; ++sub) {
        c[sub] = reinterpret_cast<const uint4 *>(row + 256 * sub)[lane];
        s += c[sub].x + c[sub].y + c[sub].z + c[sub].w;
    }
    s = wave_sum(s);
    if (s == 0) {  // zero-count contig: the reference's normalised row is NaN
        if (lane == 0) {
            p.scores[p.q_base + q] = __builtin_nan("");
            if (p.status) atomicAdd(p.status, 1u);
        }
        return;
    }
    const double Tq = (double)s;
    double qd[4 * DSUB];   // normalised row (exact distances), then reused
    double aq = 0.0, ap = 0.0, sq = 0.0, qc2 = 0.0;
    uint32_t cmx = 0, cmn = 0xFFFFFFFFu;
    const double rcen = (double)phk_row_center(s, D);
#pragma unroll
    for (int sub = 0; sub < DSUB; ++sub) {
        const double x0 = (double)c[sub].x, x1 = (double)c[sub].y, x2 = (double)c[sub].z, x3 = (double)c[sub].w;
        sq = fma(x0, x0, fma(x1, x1, fma(x2, x2, fma(x3, x3, sq))));
        {   // norm of this 256-dimension chunk of c - c0 (see ErrBound, D > 256)
            const double y0 = x0 - rcen, y1 = x1 - rcen, y2 = x2 - rcen, y3 = x3 - rcen;
            qc2 = fmax(qc2, wave_sum(fma(y0, y0, fma(y1, y1, fma(y2, y2, y3 * y3)))));
        }
        cmx = max(max(cmx, max(c[sub].x, c[sub].y)), max(c[sub].z, c[sub].w));
        cmn = min(min(cmn, min(c[sub].x, c[sub].y)), min(c[sub].z, c[sub].w));
        qd[4 * sub + 0] = (double)c[sub].x / Tq;
        qd[4 * sub + 1] = (double)c[sub].y / Tq;
        qd[4 * sub + 2] = (double)c[sub].z / Tq;
        qd[4 * sub + 3] = (double)c[sub].w / Tq;
        const double2 m0 = reinterpret_cast<const double2 *>(p.mu64 + 256 * sub)[2 * lane];
        const double2 m1 = reinterpret_cast<const double2 *>(p.mu64 + 256 * sub)[2 * lane + 1];
        const double c0 = qd[4 * sub + 0] - m0.x, c1 = qd[4 * sub + 1] - m0.y, c2 = qd[4 * sub + 2] - m1.x, c3 = qd[4 * sub + 3] - m1.y;
        aq = fma(qd[4 * sub + 0], qd[4 * sub + 0], fma(qd[4 * sub + 1], qd[4 * sub + 1],
                 fma(qd[4 * sub + 2], qd[4 * sub + 2], fma(qd[4 * sub + 3], qd[4 * sub + 3], aq))));
        ap = fma(c0, c0, fma(c1, c1, fma(c2, c2, fma(c3, c3, ap))));
    }
    const double nq2 = wave_sum(aq), nqp2 = wave_sum(ap);
    const double vs = p.vscale / Tq;
    // the proposal's query operand is c - c0 (see phk_decide_h_kernel): Q, I of ErrBound, and the low product is taken
    // with c - T mu - (c0 - T/D)
    const CenteredOperand cop = phk_centered_operand(wave_sum(sq), Tq, wave_max((double)cmx), -wave_max(-(double)cmn), (double)D, p.eb_hsum);
    const double dlt = (double)phk_row_center(s, D) - Tq / (double)D;
    ErrBound eb;
    eb.A = sqrt(nq2) + p.mu_norm;
    eb.P = sqrt(nqp2);
    eb.Q = sqrt(qc2) / Tq * (1.0 + 1e-12); eb.I = cop.I; eb.habs = cop.habs;
    if (p.cand_a) eb.habs += p.eb_cAmax * 5.9604644775390625e-08 * (double)fmaxf(p.cand_a[q], p.cand_a[p.N + q]) * vs * 1.001;
    eb.cA = p.eb_cA; eb.cP = p.eb_cP; eb.cR = p.eb_cR; eb.cabs = p.eb_abs; eb.cQ = p.eb_cQ; eb.cI = p.eb_cI; eb.cIf = p.eb_cIf;
    const double nqp = eb.P;
    const double nqx = sqrt(nqp2 + cop.shift2);
    auto e_hi = [&](int sg, double R) { return nqx * phk_lam_of(hp, sg, R) + eb(R); };
    // low product of column `col` (global column index): sum_i (c_i - T mu_i - dlt) lo_i, the whole wave
    auto low_product = [&](uint64_t col) {
        const _Float16 *lr = hp.lo16 + col * D;
        double acc = 0.0;
#pragma unroll
        for (int sub = 0; sub < DSUB; ++sub) {
            const uint2 l = reinterpret_cast<const uint2 *>(lr + 256 * sub)[lane];
            const _Float16 *lh = reinterpret_cast<const _Float16 *>(&l);
            const double2 m0 = reinterpret_cast<const double2 *>(p.mu64 + 256 * sub)[2 * lane];
            const double2 m1 = reinterpret_cast<const double2 *>(p.mu64 + 256 * sub)[2 * lane + 1];
            acc = fma(fma(-Tq, m0.x, (double)c[sub].x) - dlt, (double)lh[0], acc);
            acc = fma(fma(-Tq, m0.y, (double)c[sub].y) - dlt, (double)lh[1], acc);
            acc = fma(fma(-Tq, m1.x, (double)c[sub].z) - dlt, (double)lh[2], acc);
            acc = fma(fma(-Tq, m1.y, (double)c[sub].w) - dlt, (double)lh[3], acc);
        }
        return wave_sum(acc);
    };
    // the 8 candidates of a segment sorted by high-part value (descending), on every lane
    auto sorted8 = [&](int seg, uint32_t ncols, float (&rv)[8], uint32_t (&ri)[8], double &U) {
        float v = -3.0e38f;
        uint32_t ix = 0xFFFFFFFFu;
        if (lane < 8) {
            const uint64_t o = cand_at(seg, lane >> 2, lane & 3, q, p.N);
            v = p.cand_v[o];
            ix = p.cand_i[o];
            if (ix >= ncols) v = -3.0e38f;
        }
        U = fmax((double)p.cand_u[candu_at(seg, 0, q, p.N)], (double)p.cand_u[candu_at(seg, 1, q, p.N)]) * vs;
        int rank = 0;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float ov = __shfl(v, m);
            rank += (lane < 8 && (ov > v || (ov == v && m < lane))) ? 1 : 0;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const unsigned long long bal = __ballot(lane < 8 && rank == r);
            const int srcl = __ffsll((long long)bal) - 1;
            rv[r] = __shfl(v, srcl);
            ri[r] = __shfl(ix, srcl);
        }
    };
    bool ok = true;
    double knn = 0.0, cen = 0.0;
    if (p.method & PHK_METHOD_KNN) {
        const int need = p.kn;
        float rv[8];
        uint32_t ri[8];
        double U;
        sorted8(0, (uint32_t)p.M, rv, ri, U);
        ok = ri[need - 1] < (uint32_t)p.M;
        if (ok) {
            const double eg = e_hi(0, p.rmax);
            const double d2up = fmax(nqp2 - 2.0 * ((double)rv[need - 1] * vs - eg), 0.0);
            const double R0 = fmin(p.rmax, (nqp + sqrt(d2up)) * (1.0 + 1e-6));
            bool near = true;
            for (int r = 0; r < need; ++r) near = near && p.colnorm[ri[r]] <= R0;
            const double eh = near ? e_hi(0, R0) : eg, e22 = near ? eb(R0) : eb(p.rmax);
            const double thr = (double)rv[need - 1] * vs - 2.0 * eh;
            int nw = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) nw += ((double)rv[r] * vs >= thr && ri[r] < (uint32_t)p.M) ? 1 : 0;
            ok = nw <= PHK_HI_REFINE && U < thr;
            if (ok) {
                double fv[PHK_HI_REFINE];
                uint32_t fl[PHK_HI_REFINE];
#pragma unroll
                for (int r = 0; r < PHK_HI_REFINE; ++r) {
                    fv[r] = -1.0e300;
                    fl[r] = 0;
                    if (r < nw) {   // wave-uniform
                        fv[r] = ((double)rv[r] + (nw > need ? low_product(ri[r]) : 0.0)) * vs;
                        fl[r] = p.labels[ri[r]] ? 1u : 0u;
                    }
                }
#pragma unroll
                for (int a = 1; a < PHK_HI_REFINE; ++a)
#pragma unroll
                    for (int b = a; b > 0; --b) {
                        const bool sw = fv[b] > fv[b - 1];
                        const double tv = fv[b]; const uint32_t tl = fl[b];
                        fv[b] = sw ? fv[b - 1] : fv[b]; fl[b] = sw ? fl[b - 1] : fl[b];
                        fv[b - 1] = sw ? tv : fv[b - 1]; fl[b - 1] = sw ? tl : fl[b - 1];
                    }
                const double hi_v = need == 1 ? fv[0] : need == 2 ? fv[1] : fv[2];
                const double lo_v = need == 1 ? fv[1] : need == 2 ? fv[2] : fv[3];
                ok = nw == need || hi_v - lo_v > 2.0 * e22;
                int votes = 0;
                for (int r = 0; r < need; ++r) votes += (int)fl[r];
                knn = (2 * votes > need) ? 1.0 : -1.0;
            }
        }
    }
    if (ok && (p.method & PHK_METHOD_KMEANS)) {
        double d2[2] = {0.0, 0.0};
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const uint32_t ncols = k2 == 0 ? (uint32_t)p.n_cpos : (uint32_t)p.n_cneg;
            float rv[8];
            uint32_t ri[8];
            double U;
            sorted8(1 + k2, ncols, rv, ri, U);
            bool good = ok && ri[0] < ncols;
            if (good) {
                const double eg = e_hi(1 + k2, p.rmax);
                const double d2up = fmax(nqp2 - 2.0 * ((double)rv[0] * vs - eg), 0.0);
                const double R0 = fmin(p.rmax, (nqp + sqrt(d2up)) * (1.0 + 1e-6));
                const double cn = p.colnorm[p.M + (k2 ? p.n_cpos : 0) + ri[0]];
                const double eh = cn <= R0 ? e_hi(1 + k2, R0) : eg;
                good = ((double)rv[0] - (double)rv[1]) * vs > 2.0 * eh;
                if (good) {   // (the canonical form of exact_d2: raw counts and their sum)
                    double qraw[4 * DSUB];
#pragma unroll
                    for (int sub = 0; sub < DSUB; ++sub) {
                        qraw[4 * sub + 0] = (double)c[sub].x; qraw[4 * sub + 1] = (double)c[sub].y;
                        qraw[4 * sub + 2] = (double)c[sub].z; qraw[4 * sub + 3] = (double)c[sub].w;
                    }
                    d2[k2] = exact_d2<DSUB>(qraw, Tq, 1.0 / (Tq * Tq), p.C64 + ((k2 ? p.n_cpos : 0) + (uint64_t)ri[0]) * D, lane);
                }
            }
            ok = good;
        }
        const double ep = sqrt(d2[0]), en = sqrt(d2[1]);
        cen = tanh((en - ep) / (ep + en));  // scripts/phamer.py:206-209
    }
    if (lane == 0) {
        if (ok) p.scores[p.q_base + q] = knn + cen;  // scripts/phamer.py:313
        else p.slow_list[atomicAdd(p.fb_count + 2, 1u)] = (uint32_t)q;
    }
}

// ------------------------------------------------------------------------------------
// 3. exact brute force for queued queries.  Work item = (queued query, column chunk): a block
//    computes the direct-difference float64 distances of its chunk (one thread per column), then
//    reduces them to a partial record (3 nearest train columns of the chunk + nearest positive /
//    negative centroid of the chunk).  A second kernel merges the FB_CHUNKS records of a query.
// ------------------------------------------------------------------------------------
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

template <int SRC>
__global__ __launch_bounds__(256) void phk_fallback_partial_kernel(const void *__restrict__ src, RerankParams p) {
    extern __shared__ double fb_lds[];  // [0, 256): the query; then one chunk of distances
    const uint64_t D = p.D;
    double *fb_q = fb_lds, *fb_dist = fb_lds + D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t count = phk_uniform_load(p.fb_count);
    const uint64_t ncols = p.M + p.n_cpos + p.n_cneg;
    const uint64_t nch = fb_group_chunks(count, p.fb_rec_cap);
    const uint64_t cw = (ncols + nch - 1) / nch;
    FbRecord *rec = static_cast<FbRecord *>(p.fb_rec);
    const uint64_t items = (uint64_t)count * nch;
    for (uint64_t it = blockIdx.x; it < items; it += gridDim.x) {
        const uint64_t qi = it / nch, ch = it % nch;
        const uint64_t q = p.fb_list[qi];
        const uint64_t c0 = ch * cw < ncols ? ch * cw : ncols, c1 = (c0 + cw < ncols) ? c0 + cw : ncols;
        if (c0 >= c1) {   // (more chunks than columns: an empty record)
            if (threadIdx.x == 0) {
                FbRecord e;
                for (int k = 0; k < 3; ++k) { e.d[k] = INFINITY; e.i[k] = 0xFFFFFFFFu; }
                e.minpos = e.minneg = INFINITY; e.pad = 0;
                rec[it] = e;
            }
            continue;
        }
        // The distances are evaluated in the SAME float64 form, element ownership and summation order as every other
        // exact evaluation of this model shape, so that a query's score does not depend on the route that decided it
        // (which depends on how many rows its batch queued): D = 256 -- exact_d2_g16 (raw counts c and the row sum T,
        // sum (c_i - T r_i)^2 / T^2, 16 lanes per column in G16 ownership).  (Other D: phk_fallback_group_kernel; the branch
        // below for them is not reached by phk_score_fast.)
        const bool g16 = D == FAST_D;
        double Tq = 1.0, invT2 = 1.0;
        if (SRC == 0) {
            const uint32_t *row = static_cast<const uint32_t *>(src) + q * D;
            uint32_t s = 0;
            for (uint64_t d = lane; d < D; d += 64) s += row[d];  // every wave sums the whole row
            s = wave_sum(s);
            if (g16) {
                Tq = (double)s;
                invT2 = 1.0 / (Tq * Tq);
                for (uint64_t d = threadIdx.x; d < D; d += 256) fb_q[d] = (double)row[d];
            } else {
                for (uint64_t d = threadIdx.x; d < D; d += 256) fb_q[d] = (double)row[d] / (double)s;
            }
        } else {
            for (uint64_t d = threadIdx.x; d < D; d += 256) fb_q[d] = static_cast<const double *>(src)[q * D + d];
        }
        __syncthreads();
        if (g16) {
            // 16 lanes per column (contiguous 256-byte pieces of its row per load), 16 columns per pass
            for (uint64_t cb = c0; cb < c1; cb += 16) {
                const uint64_t c = cb + (threadIdx.x >> 4);
                const int t16 = threadIdx.x & 15;
                const uint64_t cc = c < c1 ? c : c1 - 1;
                const double2 *row = reinterpret_cast<const double2 *>(cc < p.M ? p.R64 + cc * D : p.C64 + (cc - p.M) * D) + t16;
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double2 r = row[16 * i];
                    const double d0 = fma(-Tq, r.x, fb_q[32 * i + 2 * t16]), d1 = fma(-Tq, r.y, fb_q[32 * i + 2 * t16 + 1]);
                    acc = fma(d0, d0, fma(d1, d1, acc));
                }
                acc = group16_sum(acc) * invT2;
                if (t16 == 0 && c < c1) fb_dist[c - c0] = (p.col_mask && c < p.M && p.col_mask[c]) ? INFINITY : acc;
            }
        } else {
            // one wave per column, 4 columns per pass
            for (uint64_t cb = c0; cb < c1; cb += 4) {
                const uint64_t c = cb + wave;
                const uint64_t cc = c < c1 ? c : c1 - 1;
                const double *row = cc < p.M ? p.R64 + cc * D : p.C64 + (cc - p.M) * D;
                double acc = 0.0;
                for (uint64_t sub = 0; sub < D / 256; ++sub) {
                    const double2 a = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane];
                    const double2 b = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane + 1];
                    const double *qv = fb_q + 256 * sub + 4 * lane;
                    const double d0 = qv[0] - a.x, d1 = qv[1] - a.y, d2 = qv[2] - b.x, d3 = qv[3] - b.y;
                    acc = fma(d0, d0, fma(d1, d1, fma(d2, d2, fma(d3, d3, acc))));
                }
                acc = wave_sum(acc);
                if (lane == 0 && c < c1) fb_dist[c - c0] = (p.col_mask && c < p.M && p.col_mask[c]) ? INFINITY : acc;
            }
        }
        __syncthreads();
        if (wave == 0) {
            FbRecord r;
            double last_d = -1.0;
            uint64_t last_i = 0;
            bool first = true;
            for (int k = 0; k < 3; ++k) {  // (distance, index)-ordered selection among train columns
                double bd = INFINITY;
                uint64_t bi = ~0ull;
                for (uint64_t c = c0 + lane; c < c1 && c < p.M; c += 64) {
                    const double d = fb_dist[c - c0];
                    const bool after = first || fb_less(last_d, last_i, d, c);
                    if (after && fb_less(d, c, bd, bi)) { bd = d; bi = c; }
                }
#pragma unroll
                for (int sft = 32; sft > 0; sft >>= 1) {
                    const double od = __shfl_xor(bd, sft);
                    const uint64_t oi = __shfl_xor(bi, sft);
                    if (fb_less(od, oi, bd, bi)) { bd = od; bi = oi; }
                }
                r.d[k] = bd;
                r.i[k] = (uint32_t)bi;  // 0xFFFFFFFF when the chunk has fewer train columns
                last_d = bd; last_i = bi; first = false;
            }
            double bp = INFINITY, bn = INFINITY;
            for (uint64_t c = c0 + lane; c < c1; c += 64) {
                if (c >= p.M && c < p.M + p.n_cpos) bp = fmin(bp, fb_dist[c - c0]);
                if (c >= p.M + p.n_cpos) bn = fmin(bn, fb_dist[c - c0]);
            }
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) {
                bp = fmin(bp, __shfl_xor(bp, sft));
                bn = fmin(bn, __shfl_xor(bn, sft));
            }
            r.minpos = bp; r.minneg = bn; r.pad = 0;
            if (lane == 0) rec[it] = r;
        }
        __syncthreads();
    }
}

// General D: the same brute force with the queued queries taken EIGHT at a time.  A queued query of the one-query kernel
// above streams the whole float64 reference through its CU (50 000 x 32 KiB at configs[4]: 1.6 GB per query, 22 ms for
// a hundred queries); here a workgroup of 8 waves holds 8 queries -- one per wave, the normalised row in registers -- and
// all of them meet every column of the item's chunk while it passes through the caches once.  Each wave evaluates its
// query exactly as phk_rerank_kernel does (exact_d2<DSUB>: same operands, element ownership and summation order), so
// a score does not depend on the route that produced it.  Items are numbered chunk-major: the workgroups that run
// together share a chunk of the reference.
template <int SRC, int DSUB>
__global__ __launch_bounds__(512) void phk_fallback_group_kernel(const void *__restrict__ src, RerankParams p) {
    constexpr int D = 256 * DSUB;
    // two reference rows (float64) in LDS: the row every wave works on and the next one on its way in.  Read straight from
    // memory by eight waves a row crossed the CU's 32 KiB L1 eight times (184 GB of L2 -> L1 traffic for a hundred queries
    // at configs[4]); through LDS it crosses once.
    extern __shared__ __attribute__((aligned(16))) uint8_t fbg_lds[];   // 2 x 8 D bytes
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t count = phk_uniform_load(p.fb_count);
    const uint64_t ncols = p.M + p.n_cpos + p.n_cneg;
    const uint64_t nch = fb_group_chunks(count, p.fb_rec_cap);
    const uint64_t cw = (ncols + nch - 1) / nch;
    const uint64_t ngroups = ((uint64_t)count + 7) / 8;
    FbRecord *rec = static_cast<FbRecord *>(p.fb_rec);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)fbg_lds;
    // row `c` -> buffer `buf`: wave w moves the 1 KiB pieces w, w + 8, ..  (LDS-DMA: lane l's 16 bytes land at piece + 16 l)
    auto stage = [&](uint64_t c, int buf) {
        const uint8_t *row = reinterpret_cast<const uint8_t *>(c < p.M ? p.R64 + c * D : p.C64 + (c - p.M) * D);
#pragma unroll
        for (int pc = 0; pc < (D * 8) / 8192 + 1; ++pc) {
            const uint32_t piece = (uint32_t)wave + 8u * (uint32_t)pc;
            if (piece * 1024u < (uint32_t)(D * 8)) {
                const uint8_t *gp = row + piece * 1024u + (uint32_t)lane * 16u;
                const uint32_t lp = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)buf * (uint32_t)(D * 8) + piece * 1024u);
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(lp) : "memory");
            }
        }
    };
    for (uint64_t it = blockIdx.x; it < ngroups * nch; it += gridDim.x) {   // (uniform over the workgroup)
        const uint64_t ch = it / ngroups, g8 = (it % ngroups) * 8;
        const uint64_t qi = g8 + (uint64_t)wave;
        const bool active = qi < count;                 // a group's last waves may have no query: they keep the barriers
        const uint64_t q = p.fb_list[active ? qi : g8];
        const uint64_t c0 = ch * cw < ncols ? ch * cw : ncols, c1 = (c0 + cw < ncols) ? c0 + cw : ncols;
        if (c0 >= c1) {   // (more chunks than columns)
            if (active && lane == 0) {
                FbRecord e;
                for (int k = 0; k < 3; ++k) { e.d[k] = INFINITY; e.i[k] = 0xFFFFFFFFu; }
                e.minpos = e.minneg = INFINITY; e.pad = 0;
                rec[qi * nch + ch] = e;
            }
            continue;
        }
        stage(c0, 0);
        double qd[4 * DSUB];
        double Tq = 1.0, invT2 = 1.0;
        if (SRC == 0) {
            const uint32_t *row = static_cast<const uint32_t *>(src) + q * D;
            uint4 c[DSUB];
            uint32_t sm = 0;
#pragma unroll
            for (int sub = 0; sub < DSUB; ++sub) {
                c[sub] = reinterpret_cast<const uint4 *>(row + 256 * sub)[lane];
                sm += c[sub].x + c[sub].y + c[sub].z + c[sub].w;
            }
            const double ds = (double)wave_sum(sm);
            Tq = ds;
            invT2 = 1.0 / (ds * ds);
#pragma unroll
            for (int sub = 0; sub < DSUB; ++sub) {   // the counts themselves: see exact_d2
                qd[4 * sub + 0] = (double)c[sub].x; qd[4 * sub + 1] = (double)c[sub].y;
                qd[4 * sub + 2] = (double)c[sub].z; qd[4 * sub + 3] = (double)c[sub].w;
            }
        } else {
            const double *row = static_cast<const double *>(src) + q * D;
#pragma unroll
            for (int sub = 0; sub < DSUB; ++sub) {
                const double2 a = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane];
                const double2 b = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane + 1];
                qd[4 * sub + 0] = a.x; qd[4 * sub + 1] = a.y; qd[4 * sub + 2] = b.x; qd[4 * sub + 3] = b.y;
            }
        }
        FbRecord r;
#pragma unroll
        for (int k = 0; k < 3; ++k) { r.d[k] = INFINITY; r.i[k] = 0xFFFFFFFFu; }
        r.minpos = r.minneg = INFINITY;
        r.pad = 0;
        for (uint64_t c = c0; c < c1; ++c) {
            const int buf = (int)((c - c0) & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of row c have landed ...
            __syncthreads();                                   // ... and everybody's; everybody is done with the other buffer
            if (c + 1 < c1) stage(c + 1, buf ^ 1);
            // exact_d2<DSUB> with the row read from LDS: same operands, element ownership and summation order
            const double *row = reinterpret_cast<const double *>(fbg_lds + (size_t)buf * (D * 8));
            double acc = 0.0;
#pragma unroll
            for (int sub = 0; sub < DSUB; ++sub) {
                const double2 a = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane];
                const double2 b = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane + 1];
                const double d0 = fma(-Tq, a.x, qd[4 * sub + 0]), d1 = fma(-Tq, a.y, qd[4 * sub + 1]);
                const double d2 = fma(-Tq, b.x, qd[4 * sub + 2]), d3 = fma(-Tq, b.y, qd[4 * sub + 3]);
                acc = fma(d0, d0, fma(d1, d1, fma(d2, d2, fma(d3, d3, acc))));
            }
            double dist = wave_sum(acc) * invT2;   // the same value on every lane
            if (c < p.M) {
                if (p.col_mask && p.col_mask[c]) dist = INFINITY;
                double d = dist;
                uint64_t ix = c;
#pragma unroll
                for (int k = 0; k < 3; ++k) {   // (distance, index)-ordered
                    const uint64_t cur = r.i[k] == 0xFFFFFFFFu ? ~0ull : (uint64_t)r.i[k];
                    if (fb_less(d, ix, r.d[k], cur)) {
                        const double td = r.d[k];
                        r.d[k] = d; r.i[k] = (uint32_t)ix; d = td; ix = cur;
                    }
                }
            } else if (c < p.M + p.n_cpos) {
                r.minpos = fmin(r.minpos, dist);
            } else {
                r.minneg = fmin(r.minneg, dist);
            }
        }
        __syncthreads();   // the last row is read: the next item's first row may overwrite buffer 0
        if (active && lane == 0) rec[qi * nch + ch] = r;
    }
}

// one thread per queued query: merge its FB_CHUNKS partial records and emit the score
__global__ __launch_bounds__(256) void phk_fallback_merge_kernel(RerankParams p) {
    const uint32_t count = *p.fb_count;
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.stat_total) {   // statistics: this batch's counters into the call's totals
        // (atomic: the striped copies are added by another workgroup of this launch)
        atomicAdd(p.stat_total + 0, p.fb_count[0]);
        atomicAdd(p.stat_total + 1, p.fb_count[1] + (p.exact_extra ? *p.exact_extra : 0u));
        if (p.map_count) atomicAdd(p.stat_total + 2, *p.map_count);   // queries that took the second chance
        if (p.q2_count) {   // general D: rows re-swept with three digits / swept by the f16 kernel (both are second chances)
            atomicAdd(p.stat_total + 2, p.q2_count[0] + p.q2_count[1]);
            atomicAdd(p.stat_total + 7, p.q2_count[0]);
            atomicAdd(p.stat_total + 8, p.q2_count[1]);
        }
        if (p.counters)                                      // why the high-parts-only decision stage passed them on
            for (int i = 0; i < 4; ++i) atomicAdd(p.stat_total + 3 + i, p.counters[8 + i]);
    }
    if (blockIdx.x == 1 && p.stat_total && p.stripes) {          // ... and the striped copies of the same words
        for (uint32_t sidx = threadIdx.x; sidx < PHK_STRIPES; sidx += blockDim.x) {
            const uint32_t *w = p.stripes + sidx * 32u;
            if (w[1]) atomicAdd(p.stat_total + 1, w[1]);
            for (int i = 0; i < 4; ++i)
                if (w[8 + i]) atomicAdd(p.stat_total + 3 + i, w[8 + i]);
        }
    }
    const FbRecord *rec = static_cast<const FbRecord *>(p.fb_rec);
    for (uint64_t qi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; qi < count;
         qi += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t q = p.fb_list[qi];
        double bd[3] = {INFINITY, INFINITY, INFINITY};
        uint64_t bi[3] = {~0ull, ~0ull, ~0ull};
        double bp = INFINITY, bn = INFINITY;
        const uint32_t nch = fb_group_chunks(count, p.fb_rec_cap);
        for (uint32_t ch = 0; ch < nch; ++ch) {
            const FbRecord r = rec[qi * nch + ch];
            for (int k = 0; k < 3; ++k) {
                if (r.i[k] == 0xFFFFFFFFu) continue;
                double d = r.d[k];
                uint64_t c = r.i[k];
                for (int s = 0; s < 3; ++s)
                    if (fb_less(d, c, bd[s], bi[s])) {
                        const double td = bd[s]; const uint64_t ti = bi[s];
                        bd[s] = d; bi[s] = c; d = td; c = ti;
                    }
            }
            bp = fmin(bp, r.minpos);
            bn = fmin(bn, r.minneg);
        }
        double knn = 0.0, cen = 0.0;
        if (p.method & PHK_METHOD_KNN) {
            int votes = 0;
            for (int k = 0; k < p.kn; ++k) votes += p.labels[bi[k]] ? 1 : 0;
            knn = (2 * votes > p.kn) ? 1.0 : -1.0;
        }
        if (p.method & PHK_METHOD_KMEANS) {
            const double ep = sqrt(bp), en = sqrt(bn);
            cen = tanh((en - ep) / (ep + en));
        }
        p.scores[p.q_base + q] = knn + cen;
    }
    // the last workgroup out zeroes the set's control words: every workgroup's reads of them precede its ticket
    if (p.clean_counters) {
        __shared__ uint32_t s_last;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(p.clean_counters + 15, 1u) == gridDim.x - 1 ? 1u : 0u;
        __syncthreads();
        if (s_last) {
            for (uint32_t i = threadIdx.x; i < 16u; i += blockDim.x) p.clean_counters[i] = 0;
            if (p.clean_stripes)
                for (uint32_t i = threadIdx.x; i < PHK_STRIPES * 32u; i += blockDim.x) p.clean_stripes[i] = 0;
        }
    }
}

// one thread per query: score += tanh((en - ep) / (ep + en)) for the queries whose centroid distances the general-D
// decision kernel left in `pend` (the rest holds the NaN fill); scripts/phamer.py:206-209, 313
__global__ __launch_bounds__(256) void phk_finish_cen_kernel(uint64_t N, const double *__restrict__ pend, double *__restrict__ scores) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N) return;
    const double dp2 = pend[2 * q], dn2 = pend[2 * q + 1];
    if (!(dp2 >= 0.0)) return;
    const double ep = sqrt(dp2), en = sqrt(dn2);
    scores[q] += tanh((en - ep) / (ep + en));
}

// rows list[0 .. n) of a count matrix -> a dense matrix (+ their row sums): the sub-batch of a second pass.  One wave per row.
__global__ __launch_bounds__(256) void phk_gather_rows_kernel(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ rowsum,
                                                              const uint32_t *__restrict__ list, uint64_t n, uint64_t D,
                                                              uint32_t *__restrict__ out, uint32_t *__restrict__ out_sum) {
    const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n) return;
    const uint64_t r = list[w];
    const uint4 *src = reinterpret_cast<const uint4 *>(counts + r * D);
    uint4 *dst = reinterpret_cast<uint4 *>(out + w * D);
    for (uint64_t i = lane; i < D / 4; i += 64) dst[i] = src[i];
    if (lane == 0 && rowsum) out_sum[w] = rowsum[r];
}

// a short hand-over queue goes straight to the brute force: its rows are appended to that queue
__global__ __launch_bounds__(256) void phk_append_queue_kernel(const uint32_t *__restrict__ list, uint32_t n, uint32_t *__restrict__ fb_list,
                                                               uint32_t *__restrict__ fb_count, uint32_t *__restrict__ q_count) {
    __shared__ uint32_t base;
    if (threadIdx.x == 0) {
        base = atomicAdd(fb_count, n);
        *q_count = 0;   // (statistics: these rows are brute-forced, not swept again)
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) fb_list[base + i] = list[i];
}

// ------------------------------------------------------------------------------------
// driver
// ------------------------------------------------------------------------------------
template <int SRC>
static int launch_rerank(phk_ctx *ctx, unsigned blocks, const void *src, const RerankParams &p) {
    // phk_rerank_kernel walks the queries grid-stride (its workgroups keep the training mean in LDS): a few workgroups per CU
    const unsigned cap = (unsigned)ctx->num_cus * 16u;
    const unsigned wblocks = (p.D >= 2048 && blocks > cap) ? cap : blocks;
    switch (p.D) {
        case 256: {
            const char rr = ctx->knobs.rerank;
            if (rr == 'w') {  // one wave per query (the general kernel), for A/B comparison
                PHK_LAUNCH(ctx, "phk_rerank_kernel", phk_rerank_kernel<SRC, 1><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(src, p));
            } else if (rr == 'g') {  // four queries per wave for every query (the decision kernel off)
                PHK_LAUNCH(ctx, "phk_rerank16_kernel",
                           (phk_rerank16_kernel<SRC, 0><<<dim3((unsigned)phk_div_up(p.N, 16)), dim3(256), 0, ctx->stream>>>(src, p)));
            } else {   // one lane per query for what the margin test certifies, then four per wave for the rest
                PHK_LAUNCH(ctx, "phk_decide_kernel",
                           phk_decide_kernel<SRC><<<dim3((unsigned)phk_div_up(p.N, 64)), dim3(64), 0, ctx->stream>>>(src, p));
                PHK_LAUNCH(ctx, "phk_rerank16_kernel",
                           (phk_rerank16_kernel<SRC, 1><<<dim3((unsigned)phk_div_up(p.N, 16)), dim3(256), 0, ctx->stream>>>(src, p)));
            }
            break;
        }
#define PHK_DECIDE_GEN(DS)                                                                                                       \
    do {                                                                                                                         \
        if (SRC == 0 && p.L8 && p.rowsum && ctx->knobs.rerank != 'w') {                                                          \
            /* the lane-per-query decision kernel first; what it hands on, listed, to the wave-per-query kernel */              \
            PHK_LAUNCH(ctx, "phk_decide_gen_kernel", (phk_decide_gen_kernel<DS><<<dim3((unsigned)phk_div_up(p.N, 64)), dim3(64), 0, ctx->stream>>>( \
                                                         static_cast<const uint32_t *>(src), p)));                               \
            RerankParams pl = p;                                                                                                 \
            pl.slow_back = 2;                                                                                                    \
            PHK_LAUNCH(ctx, "phk_rerank_kernel", (phk_rerank_kernel<0, DS, true><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(src, pl))); \
            return PHK_OK;                                                                                                       \
        }                                                                                                                        \
    } while (0)
        case 512:
            PHK_DECIDE_GEN(2);
            if (SRC == 0 && p.L8) { PHK_LAUNCH(ctx, "phk_rerank_kernel", (phk_rerank_kernel<0, 2, true><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(src, p))); }
            else { PHK_LAUNCH(ctx, "phk_rerank_kernel", (phk_rerank_kernel<SRC, 2><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(src, p))); }
            break;
        case 1024:
            PHK_DECIDE_GEN(4);
            if (SRC == 0 && p.L8) { PHK_LAUNCH(ctx, "phk_rerank_kernel", (phk_rerank_kernel<0, 4, true><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(src, p))); }
            else { PHK_LAUNCH(ctx, "phk_rerank_kernel", (phk_rerank_kernel<SRC, 4><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(src, p))); }
            break;
        case 2048:
            PHK_DECIDE_GEN(8);
            if (SRC == 0 && p.L8) { PHK_LAUNCH(ctx, "phk_rerank_kernel", (phk_rerank_kernel<0, 8, true><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(src, p))); }
            else { PHK_LAUNCH(ctx, "phk_rerank_kernel", (phk_rerank_kernel<SRC, 8><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(src, p))); }
            break;
        case 4096:
            PHK_DECIDE_GEN(16);
            if (SRC == 0 && p.L8) { PHK_LAUNCH(ctx, "phk_rerank_kernel", (phk_rerank_kernel<0, 16, true><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(src, p))); }
            else { PHK_LAUNCH(ctx, "phk_rerank_kernel", (phk_rerank_kernel<SRC, 16><<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(src, p))); }
            break;
        default:
            phk_set_error("phk_score: no decision kernel for D = %llu", (unsigned long long)p.D);
            return PHK_ERR_UNSUPPORTED;
    }
#undef PHK_DECIDE_GEN
    return PHK_OK;
}

bool phk_fast_supports_dim(uint64_t D) { return D == 256 || D == 512 || D == 1024 || D == 2048 || D == 4096; }

int phk_score_fast(phk_ctx *ctx, const phk_model *m, const double *d_Q, const uint32_t *d_counts,
                   const uint32_t *d_rowsum, uint64_t N, int method, double *d_scores, uint32_t *d_status) {
    const uint64_t D = m->D;
    // batch: bounds the candidate (200 B/query), fallback (1 KiB/query) and split-query (4 D B/query) workspaces
    // proposal pass: split-f16 MFMA by default; proposal=f32 selects the fp32-input MFMA kernel (k = 4 only)
    const char *prop = ctx->knobs.proposal;
    // count rows: the count-exact kernels (the integer counts are the MFMA operand); proposal=f16 keeps the split-query one
    const bool use_cx = d_counts && !(prop[0] == 'f' && prop[1] == '1') && (D != FAST_D || m->d_Af16h);
    // Second chance (k = 4, count-exact first pass): what the first pass cannot decide -- rows holding a count above
    // 2048, which the fp16 count operand cannot carry (long or low-complexity contigs), and the rare query whose
    // candidate lists fail certification -- is NOT sent to the float64 brute force at once.  Those rows go through the
    // split-query MFMA kernel (any magnitude: the counts are normalised, centred and split on the fly) addressed through
    // the first pass's queue, with their own list set; only what that pass cannot certify either is brute-forced.
    const bool second = use_cx && D == FAST_D;
    // first pass at k = 4: the high-parts-only kernel (1 MFMA per k-step) + its decision stage.  (Rounds 2-4 kept the kernel
    // with both parts in the sweep, phk_knn_f16c_kernel, 2 MFMAs per k-step, behind proposal=cx2; removed in round 5.)
    const bool hi_only = second;
    // general D (k = 5, 6): the same idea exists (HI flavour of the general kernel + phk_rerank_h_kernel) but is opt-in
    // (proposal=hi): on the BASELINE configurations' synthetic reference genomes, which are nearly equidistant from every
    // query, the high-part windows are wide -- 20 % of config 4's queries fell through to the brute force -- and the
    // kernel time saved (111 -> 72 ms) is lost in the tail (profiles/r02/README.md)
    const bool hi_gen = use_cx && D != FAST_D && m->d_lo16 && m->d_betah16 && prop[0] == 'h' && prop[1] == 'i';
    // general D, count rows: the int8 sweep (score_i8.hip; 3 exact-integer MFMAs per 32 dimensions where the f16 count-exact
    // kernel issues 4) unless the model's centroids were replaced or a column mask is set (its records are not updated by
    // the cross-validation service); proposal=cxf keeps the f16 count-exact kernel
    const bool use_i8 = use_cx && D != FAST_D && m->d_A8 && !m->bf_stale && !hi_gen && !(prop[0] == 'c' && prop[1] == 'x' && prop[2] == 'f');
    // its two-part form (H and M digits in the sweep, the L product added by the decision kernel to the window's members) is
    // the default; proposal=i83 keeps all three parts in the sweep
    const bool i8_two = use_i8 && m->d_A8h && m->d_L8 && !(prop[0] == 'i' && prop[1] == '8' && prop[2] == '3');
    // (the split-query workspace is what bounds a batch: 4 D bytes per query for the f16 sweeps, D for the int8 sweep -- whose
    // batches are therefore four times larger: half as many per-batch launches and read-backs at configs[2])
    uint64_t BATCH = 1ull << 20;
    while (BATCH > 4096 && BATCH * D * (use_i8 ? 1 : 4) > (2ull << 30)) BATCH >>= 1;
    if (ctx->knobs.score_batch) BATCH = ctx->knobs.score_batch < 64 ? 64 : ctx->knobs.score_batch;
    const uint64_t nb_max = N < BATCH ? N : BATCH;
    const uint64_t cap2 = second ? (nb_max / 8 > 4096 ? nb_max / 8 : (nb_max < 4096 ? nb_max : 4096)) : 0;
    const uint64_t per_list = nb_max * NSEG * 2, per_list2 = cap2 * NSEG * 2;
    const uint64_t list_bytes = sizeof(float4) + sizeof(uint4) + sizeof(float);
    void *cv, *fb, *rec;
    // the second chance sweeps the reference in PHK_SECOND_SPLITS column parts, each with a list set of its own
    const uint64_t set2_bytes = per_list2 * list_bytes;
    // general D: observed running sums (cand_a); D >= 2048: PHK_GEN_GROUPS list sets (the column groups of the 2-D launch)
    const uint64_t gen_sets = ctx->knobs.gen_groups > 0 ? (uint64_t)(ctx->knobs.gen_groups < 16 ? ctx->knobs.gen_groups : 16)
                                                        : (D >= 2048 ? PHK_GEN_GROUPS : 1);
    const uint64_t set_bytes = per_list * list_bytes;
    const uint64_t ca_bytes = D != FAST_D ? gen_sets * 2 * nb_max * sizeof(float) : 0;
    const uint64_t pend_bytes = (D != FAST_D && (method & PHK_METHOD_KMEANS)) ? nb_max * 2 * sizeof(double) : 0;
    PHK_TRY(phk_ws(ctx, WS_CAND, gen_sets * set_bytes + ca_bytes + PHK_SECOND_SPLITS * set2_bytes + pend_bytes, &cv));
    double *pend = pend_bytes ? (double *)((char *)cv + gen_sets * set_bytes + ca_bytes + PHK_SECOND_SPLITS * set2_bytes) : nullptr;
    uint32_t *ci = (uint32_t *)((char *)cv + per_list * sizeof(float4));
    float *cu = (float *)((char *)ci + per_list * sizeof(uint4));
    float *ca = ca_bytes ? (float *)((char *)cv + gen_sets * set_bytes) : nullptr;
    float *cv2 = (float *)((char *)cv + gen_sets * set_bytes + ca_bytes);
    uint32_t *ci2 = (uint32_t *)((char *)cv2 + per_list2 * sizeof(float4));
    float *cu2 = (float *)((char *)ci2 + per_list2 * sizeof(uint4));
    // WS_SCTL: [32 words: the call's totals, read by phk_score_stats] then TWO sets of {32 counter words, the three query
    // lists, the striped statistics words}, used by alternate batches: batch b's hand-over kernels (second stream, see
    // below) still read set b & 1 while batch b + 1's first pass fills the other.  Nothing here is memset per call or per
    // batch: the last workgroup of a batch's last kernel (phk_fallback_merge_kernel) zeroes the set's counters and stripes
    // after everybody has read them; a memset happens once per allocation and after a call that failed half way.
    // (the words that must read zero sit at FIXED offsets in front -- totals, then each set's counters and stripes -- and the
    // lists, whose size follows the batch, behind them: a call with another batch size finds the same words zeroed)
    const uint64_t ctl_words = 32 + (uint64_t)PHK_STRIPES * 32, list_words = 3 * nb_max + 64 * PHK_SUB_LISTS;
    PHK_TRY(phk_ws(ctx, WS_SCTL, (32 + 2 * ctl_words + 2 * list_words) * sizeof(uint32_t), &fb));
    if (ctx->score_ctl_dirty || ctx->score_ctl_gen != ctx->ws[WS_SCTL].gen) {
        PHK_HIP(hipMemsetAsync(fb, 0, (32 + 2 * ctl_words) * sizeof(uint32_t), ctx->stream));
        ctx->score_ctl_gen = ctx->ws[WS_SCTL].gen;
        ctx->score_totals_zeroed = true;
    }
    ctx->score_ctl_dirty = true;   // (cleared at the end of a call that launched everything)
    PHK_TRY(phk_ws(ctx, WS_QF32, nb_max * FB_CHUNKS * sizeof(FbRecord), &rec));
    // counter words, per batch: [0] first-pass queue length, [1] exact-distance decisions, [2] decide kernel's hand-over
    // count, [3] brute-force queue length after the second chance, [4] its exact-distance decisions, [8..11] why the
    // high-parts-only decision stage passed a query on (window wider than the refined set, window reaching past the
    // lists, refined values too close, centroid leader not certified); [16 ..] totals of the call: brute-forced queries,
    // exact-distance decisions, second-chance queries, the four reasons
    uint32_t *const totals = (uint32_t *)fb;
    uint32_t *fbc = nullptr, *fb_list = nullptr, *slow_list = nullptr, *fb2_list = nullptr, *stripes = nullptr;
    auto use_set = [&](int par) {
        fbc = totals + 32 + (uint64_t)par * ctl_words;
        stripes = fbc + 32;            // striped statistics words (RerankParams::stripes)
        fb_list = totals + 32 + 2 * ctl_words + (uint64_t)par * list_words;
        slow_list = fb_list + nb_max;
        fb2_list = slow_list + nb_max + 64 * PHK_SUB_LISTS;
    };
    use_set(0);
    const uint64_t ncols = m->M + m->n_cpos + m->n_cneg;
    const size_t fb_lds = ((ncols + FB_CHUNKS - 1) / FB_CHUNKS + D) * sizeof(double);
    PHK_REQUIRE(fb_lds <= FB_LDS_MAX, "phk_score: %llu columns exceed the fallback kernel's LDS", (unsigned long long)ncols);  // phk_model_build_fast keeps such models off this path
    // the totals of this call start from zero (phk_count_score_dev had the count planner's kernel zero them)
    if (!ctx->score_totals_zeroed) PHK_HIP(hipMemsetAsync(totals, 0, 32 * sizeof(uint32_t), ctx->stream));
    ctx->score_totals_zeroed = false;
    // General D, count rows (the int8 sweep): routing is PER ROW, never per batch.  The first pass sweeps the whole batch with two
    // digits; its decision kernel hands on what the lists cannot decide -- rows beyond the int8 operand (a bin more than 127
    // from the row's centre: long or compositionally skewed contigs) to one device queue, rows whose two-digit window holds
    // more columns than the lists (references with clusters of near-duplicate genomes) or whose candidates' exact distances
    // do not certify to another.  One 8-byte read-back per batch tells the host the two lengths; each queue's rows are
    // gathered into a dense sub-batch and swept ALONE -- the first by the f16 count-exact kernel, whose operand reaches
    // +-2048, the second by the three-digit int8 sweep, whose windows are 2^-24 wide -- and decided from those lists; only
    // what these passes cannot certify either is brute-forced.  (Round 3 declined a whole batch to the f16 kernel when more
    // than max(16, n / 256) of its rows were beyond the operand, and re-swept a whole batch -- and the rest of the call --
    // with three digits when its brute-force queue grew past max(64, n / 256).)
    uint32_t *q2c = nullptr, *q2_wide = nullptr, *q2_big = nullptr;
    if (use_i8) {
        void *q2;
        PHK_TRY(phk_ws(ctx, WS_QUEUE, 2 * nb_max * sizeof(uint32_t), &q2));
        q2_wide = (uint32_t *)q2;
        q2_big = q2_wide + nb_max;
    }
    // Multi-batch calls at k = 4 (configs[3]: 12 batches per rank): a batch's hand-over kernels -- second-chance sweep, merge,
    // its decision, brute force: 0.17 ms at a few % of the chip -- run on the context's second stream beside the NEXT batch's
    // sweep.  Forked / joined with events inside the call; main-stream work on set b & 1 waits for the tail of batch b - 2.
    const bool tail_aside = hi_only && second && N > BATCH && ctx->knobs.tail_aside;
    hipStream_t main_stream = ctx->stream;
    if (tail_aside && !ctx->aux) {
        PHK_HIP(hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
        for (auto &e : ctx->ev_fork) PHK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : ctx->ev_tail) PHK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    bool tail_used[2] = {false, false};
    auto join_tails = [&]() -> int {
        for (int par = 0; par < 2; ++par)
            if (tail_used[par]) {
                PHK_HIP(hipStreamWaitEvent(main_stream, ctx->ev_tail[par], 0));
                tail_used[par] = false;
            }
        return PHK_OK;
    };
    int rc_loop = PHK_OK;
    for (uint64_t s = 0; s < N && rc_loop == PHK_OK; s += BATCH) {
      const int par = (int)((s / BATCH) & 1);
      auto one_batch = [&]() -> int {
        const uint64_t nb = N - s < BATCH ? N - s : BATCH;
        use_set(par);
        q2c = fbc + 5;
        if (tail_used[par]) {   // the tail of batch b - 2 read this set
            PHK_HIP(hipStreamWaitEvent(main_stream, ctx->ev_tail[par], 0));
            tail_used[par] = false;
        }
        const void *src = d_counts ? (const void *)(d_counts + s * D) : (const void *)(d_Q + s * D);
        const uint32_t *rsum = d_rowsum ? d_rowsum + s : nullptr;
        if (use_i8 && !rsum) {   // the int8 path's kernels (fragments, sweep, lane-per-query decision) take the row sums as input
            void *rs;
            PHK_TRY(phk_ws(ctx, WS_NWIN, nb * sizeof(uint32_t), &rs));
            PHK_LAUNCH(ctx, "phk_rowsum_kernel",
                       phk_rowsum_kernel<<<dim3((unsigned)phk_div_up(nb, 4)), dim3(256), 0, ctx->stream>>>((const uint32_t *)src, nb, D, (uint32_t *)rs));
            rsum = (const uint32_t *)rs;
        }
        // segments the method does not need are skipped by giving them zero blocks
        const uint32_t nref = (method & PHK_METHOD_KNN) ? m->n_rblk_ref : 0;
        const uint32_t npos = (method & PHK_METHOD_KMEANS) ? m->n_rblk_pos : 0;
        const uint32_t nneg = (method & PHK_METHOD_KMEANS) ? m->n_rblk_neg : 0;
        RerankParams p;
        p.N = nb; p.M = m->M; p.n_cpos = m->n_cpos; p.n_cneg = m->n_cneg; p.D = D;
        p.kn = m->kn; p.method = method; p.rmax = m->max_colnorm; p.mu_norm = m->mu_norm;
        p.R64 = m->d_R64; p.C64 = m->d_C64; p.mu64 = m->d_mu64; p.colnorm = m->d_colnorm; p.labels = m->d_labels;
        p.cand_v = (const float *)cv; p.cand_i = ci; p.cand_u = cu; p.fb_rec = rec;
        p.scores = d_scores; p.status = d_status;
        p.fb_count = fbc; p.fb_list = fb_list; p.slow_list = slow_list; p.q_base = s;
        p.stat_total = totals;
        p.counters = fbc;
        p.stripes = stripes;
        p.col_mask = m->has_mask ? m->d_col_mask : nullptr;
        p.slow_cap = nb_max;
        p.eb_cQ = 0.0; p.eb_cI = 0.0; p.eb_hsum = 0.0; p.per_row_scale = 0;
        p.cand_a = ca; p.eb_cAmax = 0.0;
        const double rho = m->rho_inf > 0.0 && m->rho_inf < 1.0 ? m->rho_inf : 1.0;   // max_j |r~'_j|_inf / |r'_j|
        // f16 MFMA chains (see ErrBound): n instructions, each charged u (PHK_MFMA_ACC |x| |y| + PHK_MFMA_PROD |x|_inf |y|_inf)
        auto split_f16_bound = [&](RerankParams &r) {
            // n = 3D/16 instructions on (q' S as hi + lo) x (r' S as hi + lo); running sums <= (P + dq) R S^2; plus the
            // input terms (3 * 2^-22 / u = 12, doubled for dq); subnormal quantum sqrt(D) 2^-25 / S
            // D = 256 (phk_knn_f16_kernel): the hi.hi chain (D/16 instructions) and the cross terms (2D/16 instructions on
            // running sums and products 2^-10 of the first chain's: 11 * 32 * 2^-10 < 1, 18 * 32 * 2^-11 < 1) accumulate
            // separately and meet in two float32 additions (+2 on cP, +1 on cR); the general-D kernel keeps one accumulator
            // D > 256: Q = the largest chunk norm of q' and the observed running sums carry the chain (see ErrBound)
            const double n = (D == FAST_D ? 1.0 : 3.0) * (double)D / 16.0, x = D == FAST_D ? 1.0 : 0.0;
            r.vscale = 1.0 / (4096.0 * 4096.0);
            r.per_row_scale = 0; r.eb_hsum = 0.0;
            r.eb_cA = 6.0; r.eb_cI = (PHK_MFMA_PROD * n + x) * rho; r.eb_cIf = PHK_MFMA_PROD * n + x; r.eb_cR = 6.0 + x;
            if (D == FAST_D) {
                r.eb_cQ = 0.0; r.eb_cP = PHK_MFMA_ACC * n + 24.0 + 3.0 * x; r.eb_cAmax = 0.0;
            } else {
                r.eb_cQ = PHK_MFMA_ACC * n; r.eb_cP = 24.0; r.eb_cAmax = PHK_MFMA_ACC * n;
            }
            r.eb_abs = std::sqrt((double)D) * 5.9604644775390625e-08 / 4096.0;
        };
        // the error models of the count-exact lists (see ErrBound)
        auto i8_bound = [&](RerankParams &r, bool two) {
            // values are T v (per row), from exact integer sums
            const double ku = m->kappa8 / 5.9604644775390625e-08;
            r.vscale = 1.0; r.per_row_scale = 1; r.cand_a = nullptr;
            r.eb_cA = 2.0; r.eb_cQ = 4.0; r.eb_cI = 0.0; r.eb_cIf = 0.0; r.eb_cAmax = 0.0;
            r.eb_cP = ku + 2.0; r.eb_cR = ku * (1.0 + m->kappa8) + 3.0; r.eb_abs = 0.0;
            r.eb_hsum = m->hsum8;
            r.L8 = nullptr;
            if (two) {   // a refined value: one more fused multiply-add on |v| (u |v| <= u (P R + R^2 / 2)); the conversion of S_L
                         // (|g S_L| <= 2^-15 |x| |y|) is inside cQ, which the two-part value's single conversion leaves room in
                r.eb_cP += 1.0; r.eb_cR += 1.0;
                // the two-part sweep's lists carry 5 index bits in the value (score_i8.hip): 31 ulp <= 62 u |v|
                r.eb_cP += 62.0; r.eb_cR += 31.0;
                r.L8 = m->d_L8; r.T8 = m->d_T8;
                r.t8_blk[0] = 0; r.t8_blk[1] = m->n_rblk_ref; r.t8_blk[2] = m->n_rblk_ref + m->n_rblk_pos;
                for (int sg = 0; sg < 3; ++sg) r.lam8[sg] = m->lam8[sg];
            }
        };
        auto cx_bound = [&](RerankParams &r) {
            // values are T S v (per row); n = 2D/16 instructions on (c - c0) x (r~' S as hi, lo), + 3 for the bias -> fp32,
            // the final fma and slack; the residue of the centring through hsum
            // (general D keeps its indices in registers: no embedded index bits, 62 / 31 less on cP / cR)
            const double n = 2.0 * (double)D / 16.0;
            r.vscale = 1.0 / 4096.0; r.per_row_scale = 1; r.L8 = nullptr;
            r.cand_a = ca;
            r.eb_cA = 1.0; r.eb_cQ = PHK_MFMA_ACC * n + 3.0; r.eb_cI = PHK_MFMA_PROD * n * rho; r.eb_cIf = PHK_MFMA_PROD * n;
            r.eb_cAmax = D != FAST_D ? PHK_MFMA_ACC * n : 0.0;
            r.eb_cP = D == FAST_D ? 67.0 : 5.0; r.eb_cR = D == FAST_D ? 36.0 : 5.0;
            r.eb_abs = std::sqrt((double)D) * 5.9604644775390625e-08 / 4096.0;
            r.eb_hsum = m->hsum_train > m->hsum_cen ? m->hsum_train : m->hsum_cen;
        };
        // column groups of the int8 sweep: its optimum is 2 at D >= 2048 -- configs[4], two-part kernel: 40.1 / 38.5 / 42.3 /
        // 40.3 / 44.2 ms with 1 / 2 / 3 / 4 / 6 groups; the f16 kernel's is PHK_GEN_GROUPS
        const uint32_t i8_groups = ctx->knobs.gen_groups > 0 ? (uint32_t)gen_sets : (gen_sets > 2 ? 2u : (uint32_t)gen_sets);
        const bool i8_now = use_i8;
        if (i8_now) {
            PHK_TRY(phk_launch_proposal_i8_general(ctx, m, (const uint32_t *)src, rsum, nb, nref, npos, nneg, (float *)cv, ci, cu,
                                                   i8_groups, set_bytes, i8_two));
            i8_bound(p, i8_two);
            p.q2_count = q2c; p.q2_big = q2_big; p.q2_wide = i8_two ? q2_wide : nullptr;
            p.rowsum = rsum;
        } else if (use_cx) {
            cx_bound(p);
        } else {
            split_f16_bound(p);
        }
        if (i8_now) {
            // (launched above)
        } else if (D != FAST_D) {
            PHK_TRY(phk_launch_proposal_f16_general(ctx, m, src, d_counts != nullptr, use_cx, rsum, nb, nref, npos, nneg,
                                                    (float *)cv, ci, cu, ca, hi_gen, (uint32_t)gen_sets, set_bytes));
        } else if (hi_only) {
            PHK_TRY(phk_launch_proposal_f16h(ctx, m, (const uint32_t *)src, rsum, nb, nref, npos, nneg, (float *)cv, ci, cu));
        } else {
            PHK_TRY(phk_launch_proposal_f16(ctx, m, src, d_counts != nullptr, rsum, nb, nref, npos, nneg, (float *)cv,
                                            ci, cu));
        }
        const unsigned rblocks = (unsigned)phk_div_up(nb, 4);
        if (hi_only) {
            HiParams hp;
            hp.lo16 = m->d_lo16;
            for (int sg = 0; sg < 3; ++sg) {
                for (int i = 0; i <= 64; ++i) hp.lam_tab[sg][i] = m->lam_tab[sg][i];
                hp.lam_r0[sg] = m->lam_r0[sg];
                hp.lam_inv_step[sg] = 1.0 / m->lam_step[sg];
            }
            const dim3 dg((unsigned)phk_div_up(nb, 64)), db(64);
            const bool d_knn = (p.method & PHK_METHOD_KNN) != 0, d_cen = (p.method & PHK_METHOD_KMEANS) != 0;
            // the high-parts-only kernel issues D/16 MFMAs per value, not the count-exact kernel's 2D/16
            // (the low product the decision stage adds has its own term, see phk_decide_h_kernel)
            RerankParams pd = p;
            pd.sub_lists = PHK_SUB_LISTS;
            pd.sub_cap = 64 * phk_div_up(phk_div_up(nb, 64), PHK_SUB_LISTS);
            pd.eb_cQ = PHK_MFMA_ACC * ((double)D / 16.0) + 3.0;
            pd.eb_cI = PHK_MFMA_PROD * ((double)D / 16.0) * rho;
            pd.eb_cIf = PHK_MFMA_PROD * ((double)D / 16.0);
            // Round 5: the bias is the sweep's 17th MFMA step, - T b~_j as nine products of float16 pieces.  That instruction
            // runs on |running sum| <= |counts' sum| + T |b~| and its largest nominal product is <= T |b~| (1 + 2^-11)^2:
            // u (11 A + 18 p) adds 11 u on the counts' sum (the Q R term), and (11 + 18 (1 + 2^-10)) u on T |b~|, in v units
            // |b~| / S per column as below.  The pieces carry the bias rounded to the 2^-14 2^-e grid: an absolute
            // 2^-15 2^-e / S per value.
            // (charged per column: |b_j| / S = |(mu - 1/D) . r~'_j + |r~'_j|^2 / 2| <= |mu - 1/D| R + R^2 / 2 with R >= |r'_j| (1 + 2^-21);
            // by the model's largest bias, an absolute term, it cost the exact-distance kernel 26 % more queries -- outlying
            // columns far from every query set it -- and by |mu| instead of |mu - 1/D| 16 %)
            pd.eb_cQ += PHK_MFMA_ACC;
            const double cb = (PHK_MFMA_ACC + PHK_MFMA_PROD * (1.0 + 1.0 / 1024.0)) * (1.0 + 1.0 / 512.0);   // (|hi_j| <= S |r'_j| (1 + 2^-11); the pieces' own rounding)
            pd.eb_cM = cb;
            pd.eb_M = m->mu_tilde_norm;
            pd.eb_cR += 0.5 * cb;
            pd.eb_babs = std::ldexp(1.0, -15 - m->bias_e) / 4096.0;
            if (d_knn && d_cen) {
                PHK_LAUNCH(ctx, "phk_decide_h_kernel", (phk_decide_h_kernel<true, true><<<dg, db, 0, ctx->stream>>>((const uint32_t *)src, pd, hp)));
            } else if (d_knn) {
                PHK_LAUNCH(ctx, "phk_decide_h_kernel", (phk_decide_h_kernel<true, false><<<dg, db, 0, ctx->stream>>>((const uint32_t *)src, pd, hp)));
            } else {
                PHK_LAUNCH(ctx, "phk_decide_h_kernel", (phk_decide_h_kernel<false, true><<<dg, db, 0, ctx->stream>>>((const uint32_t *)src, pd, hp)));
            }
            // what it passes on is decided from the same lists by exact candidate distances where possible.  For that
            // kernel the lists' error model is the count-exact one plus the missing low product, |q'| |lo_j| / S with
            // |lo_j| <= 2^-11 (1 + 2^-11) S |r'_j| + sqrt(D) 2^-25 (half an ulp of the high part per element; the
            // second term covers fp16 subnormals): 2^-11 / u = 8192 more on cP, the absolute term doubled
            RerankParams ph = pd;
            ph.eb_cP += 8192.0 * (1.0 + 1.0 / 2048.0) + 1.0;
            ph.eb_abs *= 2.0;
            ph.slow_back = 3;   // front and back list in one launch
            // (waves: sub_lists x (sub_cap / 4 + 2) local ones, four per workgroup)
            PHK_LAUNCH(ctx, "phk_rerank16_kernel",
                       (phk_rerank16_kernel<0, 1><<<dim3((unsigned)phk_div_up((uint64_t)PHK_SUB_LISTS * (pd.sub_cap / 4 + 2), 4)), dim3(256), 0, ctx->stream>>>(src, ph)));
        } else if (hi_gen) {
            HiParams hp;
            hp.lo16 = m->d_lo16;
            for (int sg = 0; sg < 3; ++sg) {
                for (int i = 0; i <= 64; ++i) hp.lam_tab[sg][i] = m->lam_tab[sg][i];
                hp.lam_r0[sg] = m->lam_r0[sg];
                hp.lam_inv_step[sg] = 1.0 / m->lam_step[sg];
            }
#define PHK_RH(DS) PHK_LAUNCH(ctx, "phk_rerank_h_kernel", (phk_rerank_h_kernel<DS><<<dim3(rblocks), dim3(256), 0, ctx->stream>>>((const uint32_t *)src, p, hp)))
            if (D == 512) { PHK_RH(2); } else if (D == 1024) { PHK_RH(4); } else if (D == 2048) { PHK_RH(8); } else { PHK_RH(16); }
#undef PHK_RH
            // what it passes on: the one-wave-per-query kernel on the listed queries, the lists under the high-part error
            // model (see the k = 4 path above)
            RerankParams ph = p;
            ph.eb_cP += 8192.0 * (1.0 + 1.0 / 2048.0) + 1.0;
            ph.eb_abs *= 2.0;
            ph.slow_back = 2;
            PHK_TRY(launch_rerank<0>(ctx, rblocks, src, ph));
        } else {
            RerankParams pr = p;
            if (pend && ctx->knobs.rerank != 'w') {   // general D: the proximity metric is finished by a lane-per-query kernel
                PHK_HIP(hipMemsetAsync(pend, 0xFF, nb * 2 * sizeof(double), ctx->stream));   // NaN: not decided here
                pr.pend = pend;
            }
            if (d_counts) PHK_TRY(launch_rerank<0>(ctx, rblocks, src, pr));
            else PHK_TRY(launch_rerank<1>(ctx, rblocks, src, pr));
            if (i8_now) {
                // ---- the second passes: each hand-over queue as a dense sub-batch, swept alone ----
                uint32_t q2n[2] = {0, 0};
                PHK_HIP(hipMemcpyAsync(q2n, q2c, sizeof(q2n), hipMemcpyDeviceToHost, ctx->stream));
                PHK_HIP(hipStreamSynchronize(ctx->stream));
                for (int pass = 0; pass < 2; ++pass) {   // 0: three digits for the wide windows; 1: the f16 kernel for the long rows
                    const uint64_t nq = q2n[pass];
                    if (!nq) continue;
                    const uint32_t *list = pass == 0 ? q2_wide : q2_big;
                    // A sweep of the whole reference for a handful of rows is one workgroup walking every column block
                    // (0.5 ms at configs[2], where a batch queues ~7 rows): below PHK_SUBPASS_MIN rows the float64 brute
                    // force, which takes eight queued rows per workgroup and cuts the reference into chunks, is cheaper.
                    if (nq < PHK_SUBPASS_MIN) {
                        PHK_LAUNCH(ctx, "phk_append_queue_kernel",
                                   phk_append_queue_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(list, (uint32_t)nq, fb_list, fbc, q2c + pass));
                        continue;
                    }
                    void *sub;
                    PHK_TRY(phk_ws(ctx, WS_SUB, nq * (D + 1) * sizeof(uint32_t), &sub));
                    uint32_t *sub_counts = (uint32_t *)sub, *sub_sum = sub_counts + nq * D;
                    PHK_LAUNCH(ctx, "phk_gather_rows_kernel",
                               phk_gather_rows_kernel<<<dim3((unsigned)phk_div_up(nq, 4)), dim3(256), 0, ctx->stream>>>(
                                   (const uint32_t *)src, rsum, list, nq, D, sub_counts, sub_sum));
                    const uint32_t *sub_rs = rsum ? sub_sum : nullptr;
                    RerankParams p2 = pr;
                    p2.N = nq; p2.out_map = list; p2.status = nullptr; p2.rowsum = sub_rs;
                    p2.q2_count = nullptr; p2.q2_wide = p2.q2_big = nullptr;
                    if (pass == 0) {
                        PHK_TRY(phk_launch_proposal_i8_general(ctx, m, sub_counts, sub_rs, nq, nref, npos, nneg, (float *)cv, ci, cu,
                                                               i8_groups, set_bytes, false));
                        i8_bound(p2, false);
                    } else {
                        PHK_TRY(phk_launch_proposal_f16_general(ctx, m, sub_counts, true, true, sub_rs, nq, nref, npos, nneg,
                                                                (float *)cv, ci, cu, ca, false, (uint32_t)gen_sets, set_bytes));
                        cx_bound(p2);
                    }
                    PHK_TRY(launch_rerank<0>(ctx, (unsigned)phk_div_up(nq, 4), sub_counts, p2));
                }
            }
            if (pr.pend)
                PHK_LAUNCH(ctx, "phk_finish_cen_kernel",
                           phk_finish_cen_kernel<<<dim3((unsigned)phk_div_up(nb, 256)), dim3(256), 0, ctx->stream>>>(nb, pend, d_scores + s));
        }
        RerankParams pf = p;   // what the brute force works from
        pf.status = d_status;
        pf.fb_rec_cap = nb_max * FB_CHUNKS;
        if (tail_aside) {   // ---- from here on: the batch's tail, on the second stream ----
            PHK_HIP(hipEventRecord(ctx->ev_fork[par], main_stream));
            PHK_HIP(hipStreamWaitEvent(ctx->aux, ctx->ev_fork[par], 0));
            ctx->stream = ctx->aux;
        }
        if (second) {
            const uint64_t cap = nb < cap2 ? nb : cap2;
            PHK_TRY(phk_launch_proposal_f16(ctx, m, src, true, rsum, cap, nref, npos, nneg, cv2, ci2, cu2, fb_list, fbc,
                                            PHK_SECOND_SPLITS, set2_bytes));
            RerankParams p2 = p;
            split_f16_bound(p2);
            p2.N = cap;
            p2.cand_v = cv2; p2.cand_i = ci2; p2.cand_u = cu2; p2.cand_a = nullptr;
            p2.map = fb_list; p2.map_count = fbc;
            p2.fb_count = fbc + 3; p2.fb_list = fb2_list;
            p2.exact_extra = fbc + 1;
            PHK_LAUNCH(ctx, "phk_rerank16_kernel",
                       (phk_rerank16_kernel<0, 2><<<dim3((unsigned)phk_div_up(cap, 16)), dim3(256), 0, ctx->stream>>>(src, p2)));
            pf = p2;
            pf.N = nb;
            pf.fb_rec_cap = nb_max * FB_CHUNKS;
        }
        if (D == FAST_D) {
            if (d_counts) {
                PHK_LAUNCH(ctx, "phk_fallback_partial_kernel",
                           phk_fallback_partial_kernel<0><<<dim3((unsigned)ctx->num_cus * 2), dim3(256), fb_lds, ctx->stream>>>(src, pf));
            } else {
                PHK_LAUNCH(ctx, "phk_fallback_partial_kernel",
                           phk_fallback_partial_kernel<1><<<dim3((unsigned)ctx->num_cus * 2), dim3(256), fb_lds, ctx->stream>>>(src, pf));
            }
        } else {
            // general D: the queued queries eight at a time against a chunk of the reference (see the kernel)
            pf.fb_rec_cap = nb_max * FB_CHUNKS;
            const dim3 fg((unsigned)ctx->num_cus * 2), fbk(512);
            const size_t fgl = 2 * D * sizeof(double);   // <= 64 KiB (D <= 4096)
#define PHK_FBG(DS)                                                                                                        \
    do {                                                                                                                   \
        if (d_counts) { PHK_LAUNCH(ctx, "phk_fallback_group_kernel", (phk_fallback_group_kernel<0, DS><<<fg, fbk, fgl, ctx->stream>>>(src, pf))); } \
        else { PHK_LAUNCH(ctx, "phk_fallback_group_kernel", (phk_fallback_group_kernel<1, DS><<<fg, fbk, fgl, ctx->stream>>>(src, pf))); }          \
    } while (0)
            if (D == 512) PHK_FBG(2); else if (D == 1024) PHK_FBG(4); else if (D == 2048) PHK_FBG(8); else PHK_FBG(16);
#undef PHK_FBG
        }
        pf.clean_counters = fbc;   // the set's last kernel leaves its counters and stripes zeroed for batch b + 2 / the next call
        pf.clean_stripes = stripes;
        PHK_LAUNCH(ctx, "phk_fallback_merge_kernel",
                   phk_fallback_merge_kernel<<<dim3(64), dim3(256), 0, ctx->stream>>>(pf));
        if (tail_aside) {
            PHK_HIP(hipEventRecord(ctx->ev_tail[par], ctx->aux));
            tail_used[par] = true;
        }
        return PHK_OK;
      };
      rc_loop = one_batch();
      ctx->stream = main_stream;   // (whatever the batch's tail did with it)
    }
    {   // everything enqueued on the second stream is done before anything the caller enqueues next
        const int rcj = join_tails();
        if (rc_loop == PHK_OK) rc_loop = rcj;
    }
    if (rc_loop == PHK_OK) ctx->score_ctl_dirty = false;
    return rc_loop;
}

// per-device kernel attributes, called from phk_create
int phk_score_mfma_init_device(phk_ctx *ctx) {
    (void)ctx;
    PHK_HIP(hipFuncSetAttribute((const void *)phk_fallback_partial_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB_LDS_MAX));
    PHK_HIP(hipFuncSetAttribute((const void *)phk_fallback_partial_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB_LDS_MAX));
    return PHK_OK;
}

// ---- cross-validation service: one resident model, a fold = a column mask + that fold's centroids ----
extern "C" int phk_model_set_centroids(phk_ctx *ctx, phk_model *m, const double *cpos, uint64_t n_cpos, const double *cneg,
                                       uint64_t n_cneg) {
    PHK_ENTER(ctx, "phk_model_set_centroids");
    PHK_REQUIRE(m && cpos && cneg, "phk_model_set_centroids: NULL");
    PHK_REQUIRE(n_cpos == m->n_cpos && n_cneg == m->n_cneg && n_cpos > 0,
                "phk_model_set_centroids: the model was created with %llu + %llu centroids (got %llu + %llu)",
                (unsigned long long)m->n_cpos, (unsigned long long)m->n_cneg, (unsigned long long)n_cpos, (unsigned long long)n_cneg);
    const uint64_t D = m->D;
    PHK_HIP(hipStreamSynchronize(ctx->stream));   // kernels still reading the old centroids
    PHK_HIP(hipMemcpy(m->d_C64, cpos, n_cpos * D * sizeof(double), hipMemcpyHostToDevice));
    PHK_HIP(hipMemcpy(m->d_C64 + n_cpos * D, cneg, n_cneg * D * sizeof(double), hipMemcpyHostToDevice));
    if (!m->fast) return PHK_OK;
    std::vector<double> cnorm(n_cpos + n_cneg);
    double mx = m->max_colnorm_train;
    for (uint64_t r = 0; r < n_cpos + n_cneg; ++r) {
        const double *row = r < n_cpos ? cpos + r * D : cneg + (r - n_cpos) * D;
        double s2 = 0.0;
        for (uint64_t d = 0; d < D; ++d) {
            const double v = (double)(float)(row[d] - m->h_mu[d]);
            s2 += v * v;
        }
        cnorm[r] = std::sqrt(s2);
        PHK_REQUIRE(cnorm[r] == cnorm[r] && !std::isinf(cnorm[r]), "phk_model_set_centroids: centroid %llu is not finite", (unsigned long long)r);
        mx = cnorm[r] > mx ? cnorm[r] : mx;
    }
    PHK_HIP(hipMemcpy(m->d_colnorm + m->M, cnorm.data(), cnorm.size() * sizeof(double), hipMemcpyHostToDevice));
    m->max_colnorm = mx;
    m->cen_replaced = true;
    m->bf_stale = true;
    return phk_model_update_centroids_f16(m, cpos, cneg, cnorm.data());
}

extern "C" int phk_model_set_column_mask(phk_ctx *ctx, phk_model *m, const uint8_t *mask) {
    PHK_ENTER(ctx, "phk_model_set_column_mask");
    PHK_REQUIRE(m, "phk_model_set_column_mask: NULL model");
    if (!mask && !m->has_mask) return PHK_OK;
    PHK_HIP(hipStreamSynchronize(ctx->stream));
    if (mask) {
        uint64_t kept = 0;
        for (uint64_t c = 0; c < m->M; ++c) kept += mask[c] ? 0 : 1;
        PHK_REQUIRE(kept >= (uint64_t)m->kn, "phk_model_set_column_mask: %llu unmasked train rows, k_neighbors = %d",
                    (unsigned long long)kept, m->kn);
        if (!m->d_col_mask) PHK_HIP(hipMalloc((void **)&m->d_col_mask, m->M + 16));
        PHK_HIP(hipMemcpy(m->d_col_mask, mask, m->M, hipMemcpyHostToDevice));
    }
    m->has_mask = mask != nullptr;
    if (m->fast) {
        // (the fp32 / int8 operands are never masked -- those sweeps stand down while a mask is set -- so clearing the mask
        // makes them valid again unless the centroids were replaced meanwhile)
        m->bf_stale = m->has_mask || m->cen_replaced;
        PHK_TRY(phk_model_apply_mask_f16(ctx, m));
        PHK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return PHK_OK;
}
