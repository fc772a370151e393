// score_i8.hip -- int8-MFMA proposal pass for uint32 count rows at general D (k = 5, 6: D = 1024, 4096; gfx950).
//
// Replaces, for count inputs, the two-MFMA count-exact f16 sweep of score_f16.hip (phk_knn_f16_general_kernel<CX>) in the
// candidate search of learning.knn / the nearest-centroid loop (scripts/learning.py:118-128, scripts/phamer.py:250-256).
// v_mfma_i32_32x32x32_i8 runs at twice the f16 rate per instruction and accumulates in int32 EXACTLY, so
//   * the query operand is the count row minus its centre c0 (phk_row_center) as int8 -- exact for |c - c0| <= 127, which
//     at D >= 1024 covers every contig whose bins hold a few dozen windows (rows beyond it get empty lists and take the
//     brute-force queue, as rows above 2048 do in the f16 kernels);
//   * a reference column x_j = r_j - mu is held in 24-bit fixed point with a per-column quantum g_j, as THREE int8 parts
//     n = 65536 H + 256 M + L (balanced digits in [-128, 127]) -- the precision of the f16 kernel's hi + lo pair
//     (2^-22 |x|) -- so a value costs 3 MFMAs per 32 dimensions where the f16 kernel issues 4 (measured bare-loop rates
//     on this chip: 19.0 ns per i8 MFMA, 20.2 ns per f16 MFMA per SIMD; tools/micro/mfma_i8_vs_f16.hip) -- and by default
//     only TWO: the sweep multiplies H and M, and the decision kernel adds g_j S_L, an exact integer product with the
//     column's L digits, to the candidates inside the window the missing part opens (see below and phk_rerank_kernel);
//   * the three part sums are exact integers; the value  T v_j = g_j (65536 S_H + 256 S_M + S_L) - T b_j  is formed once
//     per (query, column) in float32 in the tile epilogue.  No rounding model of the matrix pipe enters the error bound:
//     what is left is the quantisation of the column (kappa = max_j |x_j - x~_j| / |x_j|, computed at build), three
//     int -> float conversions and three fused multiply-adds (phk_score_fast: i8 coefficients).
// The lane map of the instruction (row / column = lane & 31, k = 16 (lane >> 5) + byte; C/D as the f32 forms) is checked
// with exact integer data by tools/micro/mfma_i8_vs_f16.hip and, through the whole path, by the parity tests.
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

#include "phk_common.h"
#include "score_lists.h"
#include "score_model.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


#define I8_NMAX 8300000.0      // |n| <= this: H = n / 65536 (balanced) stays within [-127, 127]
#define I8_KS 2                // k-steps (32 dimensions each) per chunk of the sweep: chunks of 64 dimensions
#define I8_L1_MAX PHK_I8_L1_MAX   // (phk_common.h)
#define I8_SENT 0x03FFFFFFu    // id of an empty list slot (two-part sweep)
#define I8_INS2_BLOCKS 320     // sweeps of at most this many column blocks (10 240 columns) insert without tests (measured: see the epilogue)
#define I8_CT_MAX 6            // most column blocks a tile of any variant holds (padding of the record / term arrays)
// Two variants of the sweep, <parts NP, column blocks per tile CT> -- both keep 192 int32 accumulator registers per wave and
// 24 MFMAs + 40 LDS-DMA pieces per 64-dimension step:
//   <3, 4>  all three parts: the lists carry full-precision values (proposal=i83);
//   <2, 6>  H and M only (the default): a value is short of the full one by exactly g_j S_L, S_L = sum_i (c_i - c0) L_ji,
//           |g_j S_L| <= |c - c0| g_j |L_j| (Cauchy-Schwarz) -- 2^-16 of |c - c0| |x_j| where the 8-bit H alone would leave
//           2^-8.  The decision stage (phk_rerank_kernel<.., I8H>) adds g_j S_L, an exact integer dot product with the
//           column's L digits (row-major d_L8), to the candidates inside the window that bound opens around the need-th
//           value; they then carry the three-part value and everything downstream is the three-part path's.  A third fewer
//           MFMAs per value and six column blocks per query-fragment read instead of four.
#ifndef I8_NW
#define I8_NW 8                // waves per workgroup (they share the step sets streamed into LDS)
#endif

// ------------------------------------------------------------------------------------
// host: 24-bit fixed-point columns in MFMA fragment order.  Block record of the <NP, .> sweep: D / 64 chunks of 2 NP pieces
// [part < NP][k-step s < 2][lane (column i, half h)][16 bytes = dimensions 64 c + 32 s + 16 h + 0..15] (rec3: H, M, L;
// rec2: H, M); the L digits also row-major, [column][D], for the decision stage of the two-part sweep; beside the records,
// 64 floats per block: the 32 quanta g_j and the 32 bias terms b_j = (mu - 1/D).x~_j + |x~_j|^2 / 2 (x~ = the column as
// quantised in all three parts; see score_lists.h on the centred counts).  lam[col] = g_j |L_j|_2: what a two-part value
// can be short of the full one, per unit of |c - c0| / T.
// ------------------------------------------------------------------------------------
static void pack_segment_i8(const double *rows, uint64_t n, uint64_t D, const double *mu, const double *colnorm,
                            std::vector<uint8_t> &rec3, std::vector<uint8_t> &rec2, std::vector<int8_t> &low,
                            std::vector<float> &term, uint64_t cb0, std::vector<double> &kappa, std::vector<double> &hsum,
                            std::vector<double> &lam, uint64_t col0) {
    const uint64_t nblk = phk_div_up(n, 32);
    const int nchunk = (int)(D / (32 * I8_KS));
    const uint64_t rec3_bytes = (uint64_t)nchunk * 3 * I8_KS * 1024, rec2_bytes = (uint64_t)nchunk * 2 * I8_KS * 1024;
    const double shift = 1.0 / (double)D;
    phk_parallel_for(nblk, [&, nchunk](uint64_t b) {
        uint8_t *blk3 = rec3.data() + (cb0 + b) * rec3_bytes, *blk2 = rec2.data() + (cb0 + b) * rec2_bytes;
        float *terms = term.data() + (cb0 + b) * 64;
        for (int i = 0; i < 32; ++i) {
            const uint64_t r = b * 32 + i;
            if (r >= n) {   // padding column: zero operand, never selectable
                terms[i] = 0.0f;
                terms[32 + i] = 1.0e30f;
                continue;
            }
            double xmax = 0.0;
            for (uint64_t d = 0; d < D; ++d) {
                const double x = std::fabs(rows[r * D + d] - mu[d]);
                xmax = x > xmax ? x : xmax;
            }
            const float gf = xmax > 0.0 ? (float)(xmax / I8_NMAX) : 1.0e-30f;
            const double g = (double)gf;   // the kernel multiplies by the float: quantise against exactly that value
            double mudot = 0.0, nrm2 = 0.0, xsum = 0.0, d2 = 0.0, l2 = 0.0;
            int8_t *lrow = low.data() + (col0 + r) * D;
            for (int c = 0; c < nchunk; ++c)
                for (int s = 0; s < I8_KS; ++s)
                    for (int h = 0; h < 2; ++h)
                        for (int e = 0; e < 16; ++e) {
                            const uint64_t d = 32 * I8_KS * c + 32 * s + 16 * h + e;
                            const double x = rows[r * D + d] - mu[d];
                            long long q = std::llround(x / g);
                            q = q > (long long)I8_NMAX ? (long long)I8_NMAX : (q < -(long long)I8_NMAX ? -(long long)I8_NMAX : q);
                            const long long L = ((q + 128) % 256 + 256) % 256 - 128;
                            const long long q1 = (q - L) / 256;
                            const long long M = ((q1 + 128) % 256 + 256) % 256 - 128;
                            const long long H = (q1 - M) / 256;
                            const int lane = h * 32 + i;
                            const long long part[3] = {H, M, L};
                            for (int p = 0; p < 3; ++p)
                                reinterpret_cast<int8_t *>(blk3 + ((uint64_t)c * 3 * I8_KS + p * I8_KS + s) * 1024 + lane * 16)[e] = (int8_t)part[p];
                            for (int p = 0; p < 2; ++p)
                                reinterpret_cast<int8_t *>(blk2 + ((uint64_t)c * 2 * I8_KS + p * I8_KS + s) * 1024 + lane * 16)[e] = (int8_t)part[p];
                            lrow[d] = (int8_t)L;
                            l2 += (double)(L * L);
                            const double xt = g * (double)q;   // the column as the kernel sees it
                            mudot += (mu[d] - shift) * xt;
                            nrm2 += xt * xt;
                            xsum += xt;
                            d2 += (x - xt) * (x - xt);
                        }
            terms[i] = gf;
            terms[32 + i] = (float)(mudot + 0.5 * nrm2);
            kappa[col0 + r] = colnorm[col0 + r] > 0.0 ? std::sqrt(d2) / colnorm[col0 + r] : (d2 > 0.0 ? 1.0 : 0.0);
            hsum[col0 + r] = std::fabs(xsum);
            lam[col0 + r] = g * std::sqrt(l2);
        }
    });
}

int phk_model_build_i8(phk_model *m, const double *pos, const double *neg, const double *cpos, const double *cneg,
                       const double *mu, const double *colnorm) {
    const uint64_t D = m->D;
    if (D == FAST_D || D % 256 != 0) return PHK_OK;
    const uint64_t nblk = (uint64_t)m->n_rblk_ref + m->n_rblk_pos + m->n_rblk_neg;
    const uint64_t nchunk = D / (32 * I8_KS);
    const uint64_t rec3_bytes = nchunk * 3 * I8_KS * 1024, rec2_bytes = nchunk * 2 * I8_KS * 1024;
    const uint64_t ncols = m->M + m->n_cpos + m->n_cneg;
    // padding blocks: the sweep's prefetch runs past the end unchecked; a tile's terms are fetched as whole 1 KiB pieces
    std::vector<uint8_t> rec3((nblk + 2 * I8_CT_MAX) * rec3_bytes, 0), rec2((nblk + 2 * I8_CT_MAX) * rec2_bytes, 0);
    std::vector<int8_t> low(ncols * D, 0);
    std::vector<float> term((nblk + 2 * I8_CT_MAX + 4) * 64, 0.0f);
    for (uint64_t b = nblk; b < nblk + 2 * I8_CT_MAX + 4; ++b)
        for (int i = 0; i < 32; ++i) term[b * 64 + 32 + i] = 1.0e30f;
    std::vector<double> kappa(ncols, 0.0), hsum(ncols, 0.0), lam(ncols, 0.0);
    {
        std::vector<double> train(m->M * D);
        std::copy(pos, pos + m->n_pos * D, train.begin());
        std::copy(neg, neg + m->n_neg * D, train.begin() + m->n_pos * D);
        pack_segment_i8(train.data(), m->M, D, mu, colnorm, rec3, rec2, low, term, 0, kappa, hsum, lam, 0);
    }
    if (m->n_cpos) pack_segment_i8(cpos, m->n_cpos, D, mu, colnorm, rec3, rec2, low, term, m->n_rblk_ref, kappa, hsum, lam, m->M);
    if (m->n_cneg)
        pack_segment_i8(cneg, m->n_cneg, D, mu, colnorm, rec3, rec2, low, term, (uint64_t)m->n_rblk_ref + m->n_rblk_pos, kappa,
                        hsum, lam, m->M + m->n_cpos);
    m->kappa8 = m->hsum8 = 0.0;
    m->lam8[0] = m->lam8[1] = m->lam8[2] = 0.0;
    for (uint64_t c = 0; c < ncols; ++c) {
        m->kappa8 = kappa[c] > m->kappa8 ? kappa[c] : m->kappa8;
        m->hsum8 = hsum[c] > m->hsum8 ? hsum[c] : m->hsum8;
        const int sg = c < m->M ? 0 : (c < m->M + m->n_cpos ? 1 : 2);
        m->lam8[sg] = lam[c] > m->lam8[sg] ? lam[c] : m->lam8[sg];
    }
    if (hipMalloc(&m->d_A8, rec3.size()) != hipSuccess) return PHK_ERR_NOMEM;
    if (hipMemcpy(m->d_A8, rec3.data(), rec3.size(), hipMemcpyHostToDevice) != hipSuccess) return PHK_ERR_HIP;
    if (hipMalloc(&m->d_A8h, rec2.size()) != hipSuccess) return PHK_ERR_NOMEM;
    if (hipMemcpy(m->d_A8h, rec2.data(), rec2.size(), hipMemcpyHostToDevice) != hipSuccess) return PHK_ERR_HIP;
    if (hipMalloc((void **)&m->d_L8, low.size() + 64) != hipSuccess) return PHK_ERR_NOMEM;
    if (hipMemcpy(m->d_L8, low.data(), low.size(), hipMemcpyHostToDevice) != hipSuccess) return PHK_ERR_HIP;
    if (hipMalloc(&m->d_T8, term.size() * sizeof(float)) != hipSuccess) return PHK_ERR_NOMEM;
    if (hipMemcpy(m->d_T8, term.data(), term.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return PHK_ERR_HIP;
    for (uint64_t b = 0; b < term.size() / 64; ++b)
        for (int i = 0; i < 32; ++i) term[b * 64 + i] *= 256.0f;   // (a power of two: exact)
    if (hipMalloc(&m->d_T8h, term.size() * sizeof(float)) != hipSuccess) return PHK_ERR_NOMEM;
    if (hipMemcpy(m->d_T8h, term.data(), term.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return PHK_ERR_HIP;
    m->rec8_bytes = rec3_bytes;
    m->rec8h_bytes = rec2_bytes;
    return PHK_OK;
}

// ------------------------------------------------------------------------------------
// counts -> int8 query fragments: one wave per (query block of 32, 256 dimensions); Bq8[(qb D/32 + s) 64 + lane] holds the 16
// centred counts of query j = lane & 31, dimensions 32 s + 16 (lane >> 5) + 0..15.  Rows with |c - c0| > 127 are
// flagged in `big` (N + 1 words: the last one counts them).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void phk_split_queries_i8_kernel(const uint32_t *__restrict__ counts,
                                                                   const uint32_t *__restrict__ rowsum, uint64_t N, uint64_t D,
                                                                   uint4 *__restrict__ Bq, uint32_t *__restrict__ big,
                                                                   const uint32_t *__restrict__ mode_word) {
    // mode_word: the operand may have been prepared by the count kernel (PhkPrep8): 0 = all of it (nothing to do), 1 = all
    // but the rows flagged PHK_PREP8_MISSING, 2 (or no word) = none of it
    const uint32_t mode = mode_word ? phk_uniform_load(mode_word) : 2u;
    if (mode == 0u) return;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const uint64_t nchunk = D / 256;
    const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nqb = (N + 31) / 32;
    if (w >= nqb * nchunk) return;
    const uint64_t qb = w / nchunk, c = w % nchunk;
    const uint64_t qrow = (qb * 32 + j < N) ? qb * 32 + j : N - 1;
    const bool mine = mode == 2u || (big[qrow] & PHK_PREP8_MISSING) != 0u;   // (patch mode: this lane's row still lacks its fragments)
    if (mode == 1u && !__any(mine)) return;
    const int cen = (int)phk_row_center(rowsum[qrow], (uint32_t)D);
    uint4 *out = Bq + (w * 8) * 64 + lane;
    uint32_t mx = 0, l1 = 0;
#pragma unroll 2
    for (int s = 0; s < 8; ++s) {
        // 16 consecutive counts = one 64-byte piece of the row
        const uint4 *row = reinterpret_cast<const uint4 *>(counts + qrow * D + 256 * c + 32 * s + 16 * h);
        uint4 v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = row[e];
        uint32_t pk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t cc[4] = {v[e].x, v[e].y, v[e].z, v[e].w};
            uint32_t word = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int d = (int)cc[b] - cen;
                const uint32_t ad = (uint32_t)(d < 0 ? -d : d);
                mx = max(mx, max(ad, cc[b] >> 31 ? 0xFFFFFFFFu : 0u));
                l1 += min(ad, 127u);
                const int q = d < -127 ? -127 : (d > 127 ? 127 : d);
                word |= ((uint32_t)q & 0xFFu) << (8 * b);
            }
            pk[e] = word;
        }
        if (mine) out[s * 64] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
    // bit 31 of a row's word: some bin does not fit the operand
    if (mx > 127u && mine) atomicOr(big + qrow, 0x80000000u);
    // the low bits: |c - c0|_1 of the row, which the sweep's epilogue needs below 65 000 (it folds two exact part sums in one
    // 32-bit integer).  |c - c0|_1 <= T + D c0, so only rows with a large sum have to be measured: none of a 10 kb batch
    if (2ull * rowsum[qrow] + D > I8_L1_MAX) {
        l1 += __shfl_xor(l1, 32);
        if (h == 0 && qb * 32 + j < N && mine) atomicAdd(big + qrow, l1);
    }
}

// ------------------------------------------------------------------------------------
// the sweep: a wave keeps 32 queries x CT column blocks x NP parts in int32 accumulators (192 registers) and walks the
// dimensions in chunks of 64.  Everything a workgroup needs for one (tile, chunk) step -- the NP x 2 fragments of each of
// the tile's CT column blocks AND the 2 query fragments of each of its 8 waves, 40 pieces of 1 KiB -- is one "step
// set" in LDS, filled by LDS-DMA (5 pieces per wave) and kept in a ring of I8_NBUF sets: the set of step i + 2 is
// requested while step i computes (24 MFMAs per wave and step, ~1 us: two steps cover an L2 / fabric round trip), and
// a wave waits with s_waitcnt vmcnt(5) for ITS pieces of the current set only, then the barrier.  No vector-memory
// instruction other than those DMAs is issued inside a tile, so the count is exact; the query fragments go through LDS
// for that reason and because 192 accumulators leave no room for a chunk's fragments in registers.
//
// What the shape is tuned against (profiles/r03/README.md, int8 section; timers: -DI8_TIMERS): a wave issues about one
// instruction per 5 cycles, an MFMA of this shape occupies the pipe for 32, and two waves share a SIMD -- so the step is
// issue-bound as soon as a wave spends more than ~12 instructions per MFMA.  Hence: DMA addresses are running scalar
// pointers (2 scalar adds per piece; the first version's 64-bit address arithmetic per piece made the step 285
// instructions long, now 138), the requests sit between the MFMAs, not between the barrier and the first one, and the
// fragment reads are written out as ds_read / s_waitcnt lgkmcnt pairs three MFMAs ahead of their use.  LDS bandwidth is
// not the limit (tools/micro/lds_per_mfma.hip: 13 fragment reads per 12 MFMAs cost 12 % of the bare MFMA rate), nor
// is the ring depth (4 sets: same time) or the column-group count (1 / 2 / 4 groups within 4 %).  Two 4-wave workgroups
// per CU (-DI8_NW=4 -DI8_NBUF=2) are 25 % slower: twice the column bytes per MFMA through L2 and a one-step prefetch.
// Segment handling, candidate lists, column groups (2-D launch) as phk_knn_f16_general_kernel.
// ------------------------------------------------------------------------------------
#ifndef I8_NBUF
#define I8_NBUF 3
#endif
template <int NP, int CT>
struct I8Shape {
    static constexpr int APIECES = NP * I8_KS;                     // 1 KiB pieces per (block, chunk): NP parts x I8_KS k-steps
    static constexpr int SET_PIECES = CT * APIECES + I8_NW * I8_KS;   // 40
    static constexpr int SET_BYTES = SET_PIECES * 1024;
    static constexpr int PER_WAVE = SET_PIECES / I8_NW;            // 5 (8 waves)
    static constexpr int NX = NP * I8_KS * CT;                     // 24 MFMAs per step
    static constexpr int REQ_STRIDE = NX / PER_WAVE;               // one request every so many MFMAs
    static constexpr int TERM_PIECES = (CT * 256 + 1023) / 1024;   // the tile's column terms: CT blocks x 64 floats
    static constexpr int LDS_BYTES = I8_NBUF * SET_BYTES + TERM_PIECES * 1024;
    static_assert(SET_PIECES % I8_NW == 0 && (CT * APIECES) % I8_NW == 0 && I8_NBUF >= 2 && NX % 4 == 0 && CT <= I8_CT_MAX &&
                      TERM_PIECES <= I8_NW, "pieces per wave");
};

// a wave-uniform pointer the compiler may have computed in the vector ALU (64-bit multiplies), pinned into scalar registers:
// the DMA instruction's base operand is an "s" constraint of inline asm, which is not legalised for it
template <typename T>
__device__ __forceinline__ const T *i8_uniform_ptr(const T *q) {
    const uint64_t a = (uint64_t)(uintptr_t)q;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return reinterpret_cast<const T *>((uintptr_t)(((uint64_t)hi << 32) | lo));
}

template <int NP, int CT, int INS = 0>
__global__ __launch_bounds__(64 * I8_NW, 8 / I8_NW) void phk_knn_i8_general_kernel(
    const uint4 *__restrict__ Bq, uint64_t N, uint32_t nchunk, const uint4 *__restrict__ A8, uint64_t rec_u4,
    const uint4 *__restrict__ T8, const uint32_t *__restrict__ rowsum, const uint32_t *__restrict__ big, uint32_t blk0, uint32_t nblk_ref, uint32_t nblk_pos,
    uint32_t nblk_neg, float *__restrict__ cand_v, uint32_t *__restrict__ cand_i, float *__restrict__ cand_u, uint32_t ngroups,
    uint64_t set_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // I8_NBUF step sets + the tile's column terms
    typedef I8Shape<NP, CT> SH;
    constexpr int I8_CT = CT, I8_APIECES = SH::APIECES, I8_SET_BYTES = SH::SET_BYTES, I8_PER_WAVE = SH::PER_WAVE,
                  I8_REQ_STRIDE = SH::REQ_STRIDE;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const uint64_t nqb = (N + 31) / 32;
    uint64_t wg = blockIdx.x;
    uint32_t col0 = 0;
    if (ngroups > 1) {   // 2-D launch: see phk_knn_f16_general_kernel
        uint32_t g;
        if (ngroups >> 31) {   // one launch per column group (bits 16 .. 30: which): the groups follow one another in time, so
                               // that a group's column records stay in the XCDs' L2 caches while every workgroup streams them
            g = (ngroups >> 16) & 0x7FFFu;
            ngroups &= 0xFFFFu;
        } else {
            const uint32_t li = blockIdx.x >> 3;
            g = li % ngroups;
            wg = (uint64_t)(li / ngroups) * 8 + (blockIdx.x & 7u);
        }
        const uint32_t b0 = (uint32_t)((uint64_t)nblk_ref * g / ngroups), b1 = (uint32_t)((uint64_t)nblk_ref * (g + 1) / ngroups);
        blk0 += b0;
        col0 = 32u * b0;
        nblk_ref = b1 - b0;
        if (g + 1 != ngroups) nblk_pos = nblk_neg = 0;
        cand_v = reinterpret_cast<float *>(reinterpret_cast<char *>(cand_v) + (uint64_t)g * set_bytes);
        cand_i = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(cand_i) + (uint64_t)g * set_bytes);
        cand_u = reinterpret_cast<float *>(reinterpret_cast<char *>(cand_u) + (uint64_t)g * set_bytes);
    }
    if (wg * I8_NW >= nqb) return;   // padding workgroup of the 2-D numbering (uniform: before any barrier)
    const uint64_t qb = wg * I8_NW + wave;
    const uint64_t q0 = qb * 32;
    const uint32_t total = nblk_ref + nblk_pos + nblk_neg;
    const uint32_t seg_end0 = nblk_ref, seg_end1 = nblk_ref + nblk_pos;
    const uint32_t ntile = (total + I8_CT - 1) / I8_CT;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem;

    // The prefetch head.  Piece i of this wave (piece p = wave + I8_NW i of the set) has a running scalar pointer: a column
    // fragment (block cb = p / 6 of the tile, piece r = p % 6 of its chunk) advances by one chunk of its block's record per
    // step and, after the last chunk, to chunk 0 of the block I8_CT further on (records are contiguous: (CT - 1) nchunk + 1 chunks
    // ahead); a query fragment (wave w's block, k-step s) advances by one chunk and wraps.  No bounds are applied: a
    // partial last tile and the I8_NBUF - 1 sets requested past the end read the records' padding blocks
    // (phk_model_build_i8) or the next segment's columns, and those results are never used.
    constexpr int NA = (I8_CT * I8_APIECES) / I8_NW;   // column fragments per wave and set
    const char *pp[I8_PER_WAVE];
    uint32_t pl[I8_PER_WAVE];                          // LDS offset of the piece within its set
#pragma unroll
    for (int i = 0; i < I8_PER_WAVE; ++i) {
        const uint32_t p = (uint32_t)wave + I8_NW * i;
        if (i < NA) {
            const uint32_t cb = p / I8_APIECES, r = p % I8_APIECES;
            pp[i] = reinterpret_cast<const char *>(A8) + ((uint64_t)(blk0 + cb) * nchunk * I8_APIECES + r) * 1024;
        } else {
            const uint32_t w = (p - I8_CT * I8_APIECES) / I8_KS, s = (p - I8_CT * I8_APIECES) % I8_KS;
            uint64_t qw = wg * I8_NW + w;
            qw = qw < nqb ? qw : nqb - 1;              // padding waves re-read the last block; nothing is written
            pp[i] = reinterpret_cast<const char *>(Bq) + (qw * nchunk * I8_KS + s) * 1024;
        }
        pp[i] = i8_uniform_ptr(pp[i]);
        pl[i] = lds_base + p * 1024u;
    }
    const uint32_t lane16 = (uint32_t)lane * 16u;
    uint32_t pc = 0;                                   // chunk of the set requested next
    const int64_t a_step = I8_APIECES * 1024, a_wrap = (int64_t)((I8_CT - 1) * nchunk + 1) * I8_APIECES * 1024;
    const int64_t b_step = I8_KS * 1024, b_wrap = -(int64_t)(nchunk - 1) * I8_KS * 1024;
    auto request_piece = [&](uint32_t buf, int i) {
        const uint32_t lp = pl[i] + buf * (uint32_t)I8_SET_BYTES;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(pp[i]), "s"(lp) : "memory");
        const bool last = pc + 1 == nchunk;
        pp[i] += i < NA ? (last ? a_wrap : a_step) : (last ? b_wrap : b_step);
        if (i + 1 == I8_PER_WAVE) pc = last ? 0 : pc + 1;
    };
    auto request_set = [&](uint32_t buf) {
#pragma unroll
        for (int i = 0; i < I8_PER_WAVE; ++i) request_piece(buf, i);
    };

    float lv[CAND];
    uint32_t li[CAND];
    float ldrop = -3.0e38f;
#pragma unroll
    for (int c = 0; c < CAND; ++c) {
        lv[c] = -3.0e38f;
        li[c] = 0xFFFFFFFFu;
    }
    // Two-part sweep (NP == 2): the list of the k = 4 count-exact kernels (score_f16.hip) instead -- the low 5 mantissa bits
    // of a value carry (r << 1) | fresh, r = which of the lane's 16 rows of the column block, fresh = inserted during this
    // block; the sorted 5-deep value list (4 candidates + the best dropped value) moves with 5 v_med3 per value and no
    // index registers, and after a block the block number is shifted into the id list at the fresh positions (settle).
    // The classic list above costs ~19 instructions per value on a 4 510-column sweep -- two ballots, a parked candidate
    // and a 20-instruction sorted insert whenever a lane would park a second one, and on a short sweep nearly every pair of
    // values holds a candidate in some lane -- which made the tile epilogue a third of the D = 1024 sweep; this one ~11.
    // The 31-ulp perturbation is part of the two-part lists' error bound (phk_score_fast: +62 on cP, +31 on cR), small
    // beside the window the missing third part opens anyway.  The three-part sweep keeps exact values.
    const float vempty = __uint_as_float(__float_as_uint(-3.0e38f) & ~31u);   // empty slot: fresh bit clear
    const float fbig = 3.3e38f;
    float l5[5] = {vempty, vempty, vempty, vempty, vempty};
    uint32_t lb[4] = {I8_SENT, I8_SENT, I8_SENT, I8_SENT};
    auto insert5 = [&](float w, int r) {
        const float x = __uint_as_float((__float_as_uint(w) & ~31u) | (uint32_t)(2 * r + 1));
        // in place, last slot first (slot c takes med3(slot c-1, slot c, x), both still the old values): written as builtins
        // the five results are temporaries, and the join behind the wave-uniform skip moves them into the list's registers --
        // 4 v_mov per insertion, 384 per tile epilogue (as in phk_knn_f16h_kernel, score_f16.hip)
        asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(l5[4]) : "v"(l5[3]), "v"(x));
        asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(l5[3]) : "v"(l5[2]), "v"(x));
        asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(l5[2]) : "v"(l5[1]), "v"(x));
        asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(l5[1]) : "v"(l5[0]), "v"(x));
        asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(l5[0]) : "v"(x), "v"(fbig));
    };
    auto settle = [&](uint32_t cur) {
        uint32_t m[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(m[c]) : "v"(l5[c]));   // 0 / ~0 from the fresh bit
        lb[3] = phk_bfi_hw(m[0], lb[2], lb[3]);
        lb[2] = phk_bfi_hw(m[0], lb[1], lb[2]);
        lb[1] = phk_bfi_hw(m[0], lb[0], lb[1]);
        lb[0] = phk_bfi_hw(m[0], cur, lb[0]);
        lb[3] = phk_bfi_hw(m[1], lb[2], lb[3]);
        lb[2] = phk_bfi_hw(m[1], lb[1], lb[2]);
        lb[1] = phk_bfi_hw(m[1], cur, lb[1]);
        lb[3] = phk_bfi_hw(m[2], lb[2], lb[3]);
        lb[2] = phk_bfi_hw(m[2], cur, lb[2]);
        lb[3] = phk_bfi_hw(m[3], cur, lb[3]);
#pragma unroll
        for (int c = 0; c < 5; ++c) l5[c] = __uint_as_float(__float_as_uint(l5[c]) & ~1u);
    };
    const uint64_t qr = (qb < nqb && q0 + j < N) ? q0 + j : N - 1;
    const float negT = -(float)rowsum[qr];
    const f32x2 negT2 = {negT, negT};
    [[maybe_unused]] const f32x2 c256 = {256.0f, 256.0f};
    float pend_v = -3.0e38f;     // the lane's parked candidate (see the epilogue): value, position in the tile
    uint32_t pend_i = 0;
    const uint32_t bigw = big[qr];
    const bool isbig = (bigw >> 31) != 0 || (bigw & 0x3FFFFFFFu) > I8_L1_MAX;   // (see phk_split_queries_i8_kernel; bit 30: PHK_PREP8_MISSING)
    int seg = 0;
    uint32_t seg_first = 0;
    while (seg < NSEG && (seg == 0 ? seg_end0 : seg == 1 ? seg_end1 : total) == 0) {   // leading segments without columns
        if (qb < nqb && q0 + j < N) cand_store_empty(cand_v, cand_i, cand_u, seg, h, q0 + j, N);
        ++seg;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the loads above: from here on the DMAs are counted
    if (ntile) {
#pragma unroll
        for (int b = 0; b + 1 < I8_NBUF; ++b) request_set(b);
    }
    uint32_t cur = 0, nxt = I8_NBUF - 1;
#ifdef I8_TIMERS   // diagnostic build: cycles per phase, printed by two workgroups (tools/diag/sweep_i8_build.sh "-DI8_TIMERS")
    uint64_t tm_wait = 0, tm_bar = 0, tm_body = 0, tm_epi = 0, tm_mark = 0;
    const uint64_t tm_start = __builtin_readcyclecounter();
#define I8_TM(acc) do { const uint64_t now_ = __builtin_readcyclecounter(); acc += now_ - tm_mark; tm_mark = now_; } while (0)
    tm_mark = tm_start;
#else
#define I8_TM(acc) do { } while (0)
#endif   // ring positions of the set being read / requested
    // The step loop is rotated by three MFMAs: the last three of a step (their fragments already in registers) are issued
    // AFTER the next step's barrier, between that step's first fragment reads -- the matrix pipe has work while the first
    // LDS reads of a step are in flight, the one stretch it would otherwise idle through in every step (both waves of a
    // SIMD stand at the same barrier).  At the first step of a tile those three run on stale fragments into accumulators
    // that are cleared right after; after the last step they are issued before the epilogue.
    constexpr int NX = SH::NX;   // 24 MFMAs per step: x = (k-step, column block, part)
    i32x4 fa[4], fb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = i32x4{0, 0, 0, 0};
    fb[0] = fb[1] = i32x4{0, 0, 0, 0};
#define I8_A_OFF(x) ((((x) / NP % I8_CT) * I8_APIECES + ((x) % NP) * I8_KS + (x) / (NP * I8_CT)) * 1024)
#define I8_DS_READ(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(la), "n"(off))
#define I8_MFMA(x) acc[(x) / NP % I8_CT][(x) % NP] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[(x) & 3], fb[(x) / (NP * I8_CT)], acc[(x) / NP % I8_CT][(x) % NP], 0, 0, 0)
#define I8_MFMA0(x) acc[(x) / NP % I8_CT][(x) % NP] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[(x) & 3], fb[(x) / (NP * I8_CT)], zero16, 0, 0, 0)
    i32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0;
    for (uint32_t t = 0; t < ntile; ++t) {
        i32x16 acc[I8_CT][NP];
        // One (tile, chunk) step.  FIRST (chunk 0 of a tile): the accumulators are not cleared -- 192 v_mov per tile and wave,
        // 6 % of a D = 1024 tile's issue slots -- but written by the first k-step's MFMAs from a zero C operand, and the three
        // rotated MFMAs at the head (the previous tile's were issued before its epilogue) are left out.
        auto step = [&](auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            // this wave's pieces of the current set have landed once at most those of the I8_NBUF - 2 later sets are outstanding
#if defined(I8_TIMERS) && I8_TIMERS == 2   // fine timers: four counter reads per step
            I8_TM(tm_epi);
#endif
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(I8_PER_WAVE * (I8_NBUF - 2)) : "memory");
#if defined(I8_TIMERS) && I8_TIMERS == 2
            I8_TM(tm_wait);
#endif
            __syncthreads();   // ... and everybody's; the set read one step ago is free
#if defined(I8_TIMERS) && I8_TIMERS == 2
            I8_TM(tm_bar);
#endif
            if (FIRST && wave < SH::TERM_PIECES) {   // the tile's column terms (I8_CT blocks x 64 floats, whole 1 KiB pieces): older than
                                         // the pieces requested below, so the next step's vmcnt wait covers them; read in the epilogue, many barriers on
                const uint4 *g = i8_uniform_ptr(T8 + ((uint64_t)blk0 + (uint64_t)t * I8_CT) * 16 + (uint32_t)wave * 64u);
                const uint32_t lp = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(I8_NBUF * I8_SET_BYTES) + (uint32_t)wave * 1024u);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(g), "s"(lp) : "memory");
            }
            // 24 column fragments in sequence, read from LDS four MFMAs ahead of their use through a ring of four fragment
            // registers.  Written with explicit ds_read / s_waitcnt: left to the compiler, each fragment is read right before
            // its MFMA (it minimises live registers here) and the LDS latency shows.
            const uint32_t la = lds_base + cur * (uint32_t)I8_SET_BYTES + (uint32_t)lane * 16u;
            const uint32_t lb = la + (uint32_t)(I8_CT * I8_APIECES + I8_KS * wave) * 1024u;
            asm volatile("ds_read_b128 %0, %1" : "=v"(fb[0]) : "v"(lb));
            I8_DS_READ(fa[0], I8_A_OFF(0));
            if (!FIRST) I8_MFMA(NX - 3);   // the previous step's last three, on fa[1..3] and the old fb[1]
            I8_DS_READ(fa[1], I8_A_OFF(1));
            if (!FIRST) I8_MFMA(NX - 2);
            I8_DS_READ(fa[2], I8_A_OFF(2));
            if (!FIRST) I8_MFMA(NX - 1);
            I8_DS_READ(fa[3], I8_A_OFF(3));
            asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(fb[1]) : "v"(lb));
#pragma unroll
            for (int x = 0; x < NX - 3; ++x) {
                // reads in flight behind fragment x: x + 1 .. x + 3 (and the second query fragment while x < 4)
                if (x < 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[x & 3]), "+v"(fb[0]));
                else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[x & 3]), "+v"(fb[1]));
                if (FIRST && x < NP * I8_CT) I8_MFMA0(x);   // (k-step 0: the first touch of each accumulator)
                else I8_MFMA(x);
                if (x + 4 < NX) I8_DS_READ(fa[x & 3], I8_A_OFF(x + 4));
                // the set two steps ahead is requested here, one piece every fourth MFMA: the address arithmetic and the
                // DMA issue run under the matrix pipe's shadow instead of between the barrier and the first MFMA
                if (x % I8_REQ_STRIDE == 1 && x / I8_REQ_STRIDE < I8_PER_WAVE) request_piece(nxt, x / I8_REQ_STRIDE);
            }
            // fragments NX - 3 .. NX - 1 stay in fa[1..3] for the next step; their reads are done before this wave reaches the
            // barrier behind which the set may be overwritten
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]));
#if defined(I8_TIMERS) && I8_TIMERS == 2
            I8_TM(tm_body);
#endif
            cur = cur + 1 == I8_NBUF ? 0 : cur + 1;
            nxt = nxt + 1 == I8_NBUF ? 0 : nxt + 1;
        };
        step(std::true_type{});
        for (uint32_t c = 1; c < nchunk; ++c) step(std::false_type{});
        I8_MFMA(NX - 3);
        I8_MFMA(NX - 2);
        I8_MFMA(NX - 1);
        I8_TM(tm_body);   // (coarse timers: the whole step loop of the tile)
        // epilogue of the tile: T v = g (256 (256 S_H + S_M) + S_L) - T b, insertion, segment flushes
#pragma unroll
        for (int cb = 0; cb < I8_CT; ++cb) {
            const uint32_t blk = t * I8_CT + cb;
            if (blk < total) {
                const float4 *gp = reinterpret_cast<const float4 *>(smem + I8_NBUF * I8_SET_BYTES) + cb * 16 + h, *bp = gp + 8;
                // column of the tile's first row of this lane, relative to the segment's first column (wraps below zero for the
                // blocks of a segment that ended inside this tile: their values were inserted when it ended)
                const uint32_t tbase = 32u * (t * I8_CT - seg_first) + 4u * (uint32_t)h;
                // Eight values at a time, without a branch between them: the LDS reads of their column terms and the conversion /
                // multiply chains of the four pairs overlap (with a test after every pair each pair waited for its own term
                // reads and its own dependent chain: by the phase timers the epilogue was 37 % of the D = 1024 sweep, ~18
                // cycles per instruction).  Then one test for the eight -- late in a long sweep most hold no candidate in any
                // lane -- and the tests per pair.
                bool touched = false;   // (wave-uniform) a value of this block entered some lane's list
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {   // (half a block at a time: the terms of all 16 values at once cost spilled registers)
                    float4 g4[2], b4[2];
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        g4[m] = gp[2 * (2 * hb + m)];
                        b4[m] = bp[2 * (2 * hb + m)];
                    }
                    f32x2 val[4];
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const f32x2 gg[2] = {{g4[m].x, g4[m].y}, {g4[m].z, g4[m].w}}, bb[2] = {{b4[m].x, b4[m].y}, {b4[m].z, b4[m].w}};
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {   // two values per packed float32 instruction
                            const int r = 4 * (2 * hb + m) + 2 * e2;
                            // 256 S_H + S_M as one exact 32-bit integer (|c - c0|_1 <= I8_L1_MAX), then two conversions per value
                            // (two parts: one, and the value is 256 g (256 S_H + S_M) - T b: the full one less g S_L)
                            const f32x2 fhm = {(float)(acc[cb][0][r] * 256 + acc[cb][1][r]), (float)(acc[cb][0][r + 1] * 256 + acc[cb][1][r + 1])};
                            f32x2 sf;
                            if (NP == 3) {
                                const f32x2 fl = {(float)acc[cb][NP - 1][r], (float)acc[cb][NP - 1][r + 1]};
                                sf = __builtin_elementwise_fma(c256, fhm, fl);
                            } else {
                                sf = fhm;   // (the factor 256 is in the quantum: this variant reads the terms of d_T8h)
                            }
                            val[2 * m + e2] = __builtin_elementwise_fma(sf, gg[e2], negT2 * bb[e2]);
                        }
                    }
                    if (NP == 2 && INS == 2) {
                        // a short sweep (a few thousand columns): most values still enter some lane's list, and a wave-uniform
                        // test costs a compare, a branch and the bubble behind it -- more than the six instructions it guards
                        touched = true;
#pragma unroll
                        for (int pr = 0; pr < 4; ++pr)
#pragma unroll
                            for (int e = 0; e < 2; ++e) insert5(val[pr][e], 8 * hb + 2 * pr + e);
                        continue;
                    }
                    const float vmax = fmaxf(fmaxf(fmaxf(val[0][0], val[0][1]), fmaxf(val[1][0], val[1][1])),
                                             fmaxf(fmaxf(val[2][0], val[2][1]), fmaxf(val[3][0], val[3][1])));
                    if (NP == 2) {
                        // one test for the eight (late in a long sweep most hold no candidate in any lane), then per value: the
                        // five v_med3 only when some lane can place it (INS == 1: all eight without further tests)
                        if (__builtin_amdgcn_ballot_w64(vmax > l5[4]) != 0) {
                            touched = true;
#pragma unroll
                            for (int pr = 0; pr < 4; ++pr)
#pragma unroll
                                for (int e = 0; e < 2; ++e)
                                    if (INS == 1 || __builtin_amdgcn_ballot_w64(val[pr][e] > l5[4]) != 0) insert5(val[pr][e], 8 * hb + 2 * pr + e);
                        }
                    } else if (__builtin_amdgcn_ballot_w64(vmax > ldrop) != 0) {
#pragma unroll
                        for (int pr = 0; pr < 4; ++pr) {
                            // A value that can enter the list (> everything the list dropped) is parked in the lane's pending slot
                            // with its position in the tile (a compile-time constant); the 4-deep sorted insert -- 20 instructions
                            // for the whole wave -- runs when some lane would park a second one, an order of magnitude less often
                            // than "some lane of the 64 has a candidate", and once at the end of the tile / segment.  The pair is
                            // looked at only if its larger value is a candidate in some lane.
                            if (__builtin_amdgcn_ballot_w64(fmaxf(val[pr][0], val[pr][1]) > ldrop) != 0) {
#pragma unroll
                                for (int e = 0; e < 2; ++e) {
                                    const int r = 8 * hb + 2 * pr + e;
                                    const bool needs = val[pr][e] > ldrop;
                                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(needs && pend_v > -3.0e38f) != 0, 0)) {
                                        list_insert(lv, li, ldrop, pend_v, tbase + pend_i);   // (lanes without a parked value insert -3e38: no change)
                                        pend_v = -3.0e38f;
                                    }
                                    pend_v = needs ? val[pr][e] : pend_v;
                                    pend_i = needs ? (uint32_t)(32 * cb + (r & 3) + 8 * (r >> 2)) : pend_i;
                                }
                            }
                        }
                    }
                }
                if (NP == 2) {
                    if (touched) settle(blk);
                } else if (cb + 1 == I8_CT || (seg < NSEG && blk + 1 == (seg == 0 ? seg_end0 : seg == 1 ? seg_end1 : total))) {
                    if (__builtin_amdgcn_ballot_w64(pend_v > -3.0e38f) != 0) list_insert(lv, li, ldrop, pend_v, tbase + pend_i);
                    pend_v = -3.0e38f;
                }
                while (seg < NSEG && blk + 1 == (seg == 0 ? seg_end0 : seg == 1 ? seg_end1 : total)) {
                    if (qb < nqb && q0 + j < N) {
                        if (isbig) {
                            cand_store(cand_v, cand_i, cand_u, seg, h, q0 + j, N, -3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f,
                                       0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 3.0e38f);
                        } else if (NP == 2) {
                            // column of a slot: its block (id list, relative to the segment's first block) and its row (value bits)
                            const uint32_t o = (seg == 0 ? col0 : 0u) + 4u * (uint32_t)h - 32u * seg_first;
                            uint32_t ix[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const uint32_t r = (__float_as_uint(l5[c]) >> 1) & 15u;
                                ix[c] = lb[c] == I8_SENT ? 0xFFFFFFFFu : lb[c] * 32u + (r & 3u) + 8u * (r >> 2) + o;
                            }
                            cand_store(cand_v, cand_i, cand_u, seg, h, q0 + j, N, l5[0], l5[1], l5[2], l5[3], ix[0], ix[1], ix[2], ix[3],
                                       l5[4]);
                        } else {
                            const uint32_t o = seg == 0 ? col0 : 0u;   // (a column group: indices relative to the whole train segment)
                            cand_store(cand_v, cand_i, cand_u, seg, h, q0 + j, N, lv[0], lv[1], lv[2], lv[3],
                                       li[0] == 0xFFFFFFFFu ? li[0] : li[0] + o, li[1] == 0xFFFFFFFFu ? li[1] : li[1] + o,
                                       li[2] == 0xFFFFFFFFu ? li[2] : li[2] + o, li[3] == 0xFFFFFFFFu ? li[3] : li[3] + o, ldrop);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < CAND; ++c) {
                        lv[c] = -3.0e38f;
                        li[c] = 0xFFFFFFFFu;
                    }
                    ldrop = -3.0e38f;
#pragma unroll
                    for (int c = 0; c < 5; ++c) l5[c] = vempty;
#pragma unroll
                    for (int c = 0; c < 4; ++c) lb[c] = I8_SENT;
                    seg = __builtin_amdgcn_readfirstlane(seg + 1);   // (wave-uniform by construction: keeps the tests on it scalar)
                    seg_first = blk + 1;
                }
            }
        }
        I8_TM(tm_epi);
    }
#undef I8_MFMA
#undef I8_MFMA0
#undef I8_DS_READ
#undef I8_A_OFF
#ifdef I8_TIMERS
    if (lane == 0 && (blockIdx.x == 8 || blockIdx.x == 1001))
        printf("i8 timers wg %u wave %d: total %llu wait %llu barrier %llu body %llu epilogue+loop %llu (shader clock ticks)\n", blockIdx.x, wave,
               (unsigned long long)(tm_mark - tm_start), (unsigned long long)tm_wait, (unsigned long long)tm_bar,
               (unsigned long long)tm_body, (unsigned long long)tm_epi);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the sets requested past the end
    for (; seg < NSEG; ++seg) {
        if (qb < nqb && q0 + j < N) cand_store_empty(cand_v, cand_i, cand_u, seg, h, q0 + j, N);
    }
}

// ------------------------------------------------------------------------------------
// Measured and NOT kept (round 4; profiles/r04/README.md, "epilogue under the matrix pipe"): the tile epilogue of the two-part
// sweep inside the step loop.  All eight waves of a workgroup reach a tile's epilogue together -- 29 % of the D = 1024 sweep
// by the phase timers (tools/diag/i8_timers.sh) with no MFMA in flight -- and two sets of 192 accumulator registers do not
// exist.  Three kernels that rotate the 192 registers through accumulate / epilogue phases instead, all parity-green: three
// streams of two blocks in slots of 8 steps (91 ms at configs[2] against 85); six one-block streams three steps apart that
// pause two steps for their values (90); six streams that never pause, a finished block's 16 values met with the lists
// straight out of the accumulators behind the first 16 MFMAs of the step in which the stream starts its next block, its own
// four MFMAs last (24 MFMAs in every step, the 40-piece sets of this kernel; 90).  In each the step loop WITHOUT a single
// epilogue instruction already took 76 - 78 ms against this kernel's 60, and the interleaved values added 12 - 13 ms where
// they cost 25 outside the loop.  A wave issues about one instruction of any kind per 5 cycles (tools/micro/mfma_valu_overlap:
// six vector instructions per MFMA and wave hide under the pipe, each further one costs 1.25 ns per MFMA and SIMD), and the
// bookkeeping of per-stream phases, block pointers and term slots put 160 - 180 instructions into a step that has 138 here.
// ------------------------------------------------------------------------------------
__global__ void phk_rowsum_kernel(const uint32_t *__restrict__ counts, uint64_t N, uint64_t D, uint32_t *__restrict__ out);
__global__ void phk_merge_list_sets_kernel(float *__restrict__ cv, uint32_t *__restrict__ ci, float *__restrict__ cu, uint64_t Nlist,
                                           uint64_t set_bytes, int S, const uint32_t *__restrict__ qcount, float *__restrict__ ca,
                                           uint64_t ca_set);

// proposal pass for uint32 counts at D = 512 .. 4096: row sums (if needed) -> int8 query fragments -> sweep (-> merge)
int phk_launch_proposal_i8_general(phk_ctx *ctx, const phk_model *m, const uint32_t *d_counts, const uint32_t *d_rowsum, uint64_t nb,
                                   uint32_t nref, uint32_t npos, uint32_t nneg, float *cv, uint32_t *ci, float *cu, uint32_t groups,
                                   uint64_t set_bytes, bool two_parts) {
    const uint64_t D = m->D, nchunk = D / (32 * I8_KS), nchunk256 = D / 256;
    const uint64_t nqb = phk_div_up(nb, 32);
    void *bq, *rs = nullptr, *bg;
    // the operand the count kernel prepared (phk_count_score_dev), when this batch is a whole-block range of its matrix
    const PhkPrep8 &pp = ctx->prep8;
    const uint64_t r0 = pp.armed && d_counts >= pp.counts ? (uint64_t)(d_counts - pp.counts) / D : 0;
    const bool prepared = pp.armed && pp.D == D && d_counts >= pp.counts && (uint64_t)(d_counts - pp.counts) % D == 0 && r0 % 32 == 0 &&
                          r0 + nb <= pp.n && (nb % 32 == 0 || r0 + nb == pp.n);
    if (prepared) {
        bq = (char *)pp.frag + (r0 / 32) * (D / 32) * 1024;
        bg = pp.big + r0;
    } else {
    PHK_TRY(phk_ws(ctx, WS_Q64, nqb * (D / 32) * 1024, &bq));
    }
    if (!d_rowsum) {
        PHK_TRY(phk_ws(ctx, WS_NWIN, nb * sizeof(uint32_t), &rs));
        PHK_LAUNCH(ctx, "phk_rowsum_kernel",
                   phk_rowsum_kernel<<<dim3((unsigned)phk_div_up(nb, 4)), dim3(256), 0, ctx->stream>>>(d_counts, nb, D, (uint32_t *)rs));
        d_rowsum = (const uint32_t *)rs;
    }
    if (!prepared) {
        PHK_TRY(phk_ws(ctx, WS_LONG, (nb + 1) * sizeof(uint32_t), &bg));
        PHK_HIP(hipMemsetAsync(bg, 0, (nb + 1) * sizeof(uint32_t), ctx->stream));
    }
    // (prepared: the kernel completes what the count kernel left -- usually nothing: its waves read the mode word and leave)
    PHK_LAUNCH(ctx, "phk_split_queries_i8_kernel",
               phk_split_queries_i8_kernel<<<dim3((unsigned)phk_div_up(nqb * nchunk256, 4)), dim3(256), 0, ctx->stream>>>(
                   d_counts, d_rowsum, nb, D, (uint4 *)bq, (uint32_t *)bg, prepared ? pp.big + pp.n : nullptr));
    // Rows that do not fit the int8 operand (a bin more than 127 away from the row's centre: long or compositionally
    // skewed contigs) are flagged in `big`: the sweep stores sentinel lists for them and the decision kernel hands them to
    // the f16 count-exact sweep, row by row (phk_score_fast).
    const uint32_t blk0 = nref ? 0 : m->n_rblk_ref;
    const bool seq = ctx->knobs.gen_seq && groups > 1;   // one launch per column group instead of a 2-D launch
    const uint32_t ng = (groups > 1 && nref >= 8u * groups && nqb >= 64) ? groups : 1u;
    const uint64_t nqg = phk_div_up(nqb, I8_NW);
    const unsigned gblocks = ng > 1 && !seq ? (unsigned)(phk_div_up(nqg, 8) * 8 * ng) : (unsigned)nqg;
    if (two_parts && seq && ng > 1) {
        for (uint32_t g = 0; g < ng; ++g)
            PHK_LAUNCH(ctx, "phk_knn_i8_general_kernel",
                       (phk_knn_i8_general_kernel<2, 6><<<dim3(gblocks), dim3(64 * I8_NW), I8Shape<2, 6>::LDS_BYTES, ctx->stream>>>(
                           (const uint4 *)bq, nb, (uint32_t)nchunk, (const uint4 *)m->d_A8h, m->rec8h_bytes / 16, (const uint4 *)m->d_T8h, d_rowsum,
                           (const uint32_t *)bg, blk0, nref, npos, nneg, cv, ci, cu, 0x80000000u | (g << 16) | ng, set_bytes)));
    } else if (two_parts) {
        // how a tile's values meet the lists, by the length of the sweep per lane (see the kernel's epilogue)
        const uint32_t blocks_per_group = (nref + npos + nneg) / ng;
        int ins = blocks_per_group <= I8_INS2_BLOCKS ? 2 : 0;
        if (ctx->knobs.i8_insert) ins = ctx->knobs.i8_insert - '0';
#define PHK_I8_LAUNCH2(INS_)                                                                                                     \
        PHK_LAUNCH(ctx, "phk_knn_i8_general_kernel",                                                                             \
                   (phk_knn_i8_general_kernel<2, 6, INS_><<<dim3(gblocks), dim3(64 * I8_NW), I8Shape<2, 6>::LDS_BYTES, ctx->stream>>>( \
                       (const uint4 *)bq, nb, (uint32_t)nchunk, (const uint4 *)m->d_A8h, m->rec8h_bytes / 16, (const uint4 *)m->d_T8h, d_rowsum, \
                       (const uint32_t *)bg, blk0, nref, npos, nneg, cv, ci, cu, ng, set_bytes)))
        if (ins == 2) {
            PHK_I8_LAUNCH2(2);
        } else if (ins == 1) {
            PHK_I8_LAUNCH2(1);
        } else {
            PHK_I8_LAUNCH2(0);
        }
#undef PHK_I8_LAUNCH2
    } else {
        PHK_LAUNCH(ctx, "phk_knn_i8_general_kernel",
                   (phk_knn_i8_general_kernel<3, 4><<<dim3(gblocks), dim3(64 * I8_NW), I8Shape<3, 4>::LDS_BYTES, ctx->stream>>>(
                       (const uint4 *)bq, nb, (uint32_t)nchunk, (const uint4 *)m->d_A8, m->rec8_bytes / 16, (const uint4 *)m->d_T8, d_rowsum,
                       (const uint32_t *)bg, blk0, nref, npos, nneg, cv, ci, cu, ng, set_bytes)));
    }
    if (ng > 1) {
        PHK_LAUNCH(ctx, "phk_merge_list_sets_kernel",
                   phk_merge_list_sets_kernel<<<dim3((unsigned)phk_div_up(2 * nb, 256)), dim3(256), 0, ctx->stream>>>(
                       cv, ci, cu, nb, set_bytes, (int)ng, nullptr, nullptr, 0));
    }
    return PHK_OK;
}

int phk_score_i8_init_device(phk_ctx *ctx) {
    (void)ctx;
    PHK_HIP(hipFuncSetAttribute((const void *)phk_knn_i8_general_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, I8Shape<3, 4>::LDS_BYTES));
    PHK_HIP(hipFuncSetAttribute((const void *)phk_knn_i8_general_kernel<2, 6, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, I8Shape<2, 6>::LDS_BYTES));
    PHK_HIP(hipFuncSetAttribute((const void *)phk_knn_i8_general_kernel<2, 6, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, I8Shape<2, 6>::LDS_BYTES));
    PHK_HIP(hipFuncSetAttribute((const void *)phk_knn_i8_general_kernel<2, 6, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, I8Shape<2, 6>::LDS_BYTES));
    return PHK_OK;
}
