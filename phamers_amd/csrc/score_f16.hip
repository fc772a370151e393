// score_f16.hip -- split-f16 MFMA proposal kernel for D = 256 (k = 4).
//
// Same role as phk_knn_mfma_kernel (score_mfma.hip): PROPOSE the 4 best columns per (query, segment,
// half-list) of  v = q'.r' - |r'|^2/2 ; the float64 decision stage (phk_rerank_kernel) certifies or
// recomputes.  Instead of fp32-input MFMA (64 cycles per 32x32x2 step) it runs on the f16 matrix
// pipe at 1/16 of the cycles per flop:
//
//   each centred operand value x is scaled by S = 2^12 and split into two fp16 numbers,
//   x*S = hi + lo (+ <= 2^-22 |x*S|), and the contraction keeps three of the four cross terms:
//       a.b  ~=  a_hi.b_hi + a_hi.b_lo + a_lo.b_hi              (dropped: a_lo.b_lo <= 2^-22 |a||b|)
//   i.e. 3 x v_mfma_f32_32x32x16_f16 per 16 dimensions, fp32 accumulation inside the MFMA.
//   That is ~22 significant bits per operand -- enough for the decision stage's margin test to
//   certify most orderings (its bound is evaluated with this kernel's own error model).
//
// Column blocks (32 train rows / centroids) are shared by the 4 waves of a workgroup through LDS:
// a block record is 33 pieces of 1 KiB in MFMA fragment order,
//     piece 2s   : a_hi of step s   (lane l: row l&31, dims 128*(l>>5) + 8s .. +7, 8 halves = 16 B)
//     piece 2s+1 : a_lo of step s
//     piece 32   : 32 floats  -S^2 |r~'|^2 / 2  (r~' = (hi+lo)/S, the column as the kernel sees it), then
//                  32 floats  S (mu.r~' + |r~'|^2 / 2)  (bias of the count-exact kernel below)
// streamed with LDS-DMA (global_load_lds_dwordx4: no VGPR staging, lane-linear = fragment order) into
// a double buffer, one barrier per block.  A wave keeps its 32 queries' b_hi / b_lo fragments in 128
// VGPRs for the whole sweep, exactly like the fp32 kernel keeps q'.
#include <stdlib.h>
#include <string.h>

#include "phk_common.h"
#include "score_lists.h"
#include "score_model.h"

#include <cmath>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define F16_PIECES 33
#define F16_BLOCK_BYTES (F16_PIECES * 1024)
#define F16_SCALE_LOG2 12
#define F16_SCALE 4096.0f

// ------------------------------------------------------------------------------------
// host: build the split-f16 fragment-ordered operand
// ------------------------------------------------------------------------------------
static void split_f16(double x, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (double)hi);
}

// Block record for D = 256 * nchunk: nchunk chunks of 32 pieces (hi/lo of the 16 k-steps of that
// chunk: dims 256c + 128*(lane>>5) + 8s .. +7), then one piece with the 32 norm terms.  For D = 256
// this is the 33-piece record of the k = 4 kernel.  cn_all (one float per column slot, padding
// included) duplicates the norm terms for the general-D kernel, which reads them from global memory.
// hi-only extras (D = 256 models): betah_all = S mu.hi + S |r~'|^2 / 2 per column slot, lo_rows = the low parts, one row
// per REAL column (col0 + r), lonorm = |lo| / S per real column
// Position of dimension d inside a low-part row.  D = 256 rows are stored in the order phk_decide_h_kernel's 16-lane
// groups read them (score_mfma.hip, "G16 ownership": lane t holds dimensions 32 i + 2 t + j): lane t's 16 halves sit in two
// 16-byte pieces, [i >> 2][t][(i & 3) * 2 + j], so the group's two loads per row are contiguous 256 B each.  Other D:
// row-major (phk_rerank_h_kernel).
static inline uint64_t lo_pos(uint64_t D, uint64_t d) {
    if (D != FAST_D) return d;
    const uint64_t i = d >> 5, t = (d >> 1) & 15, j = d & 1;
    return ((i >> 2) * 16 + t) * 8 + (i & 3) * 2 + j;
}
struct HiOnlyOut {
    std::vector<float> *betah_all = nullptr;
    std::vector<_Float16> *lo_rows = nullptr;
    std::vector<double> *lonorm = nullptr;
    std::vector<double> *hsum = nullptr;   // |sum_d r~'_d| per real column (the residue of centring the counts, see ErrBound)
    std::vector<double> *xmax = nullptr;   // |r~'|_inf per real column
    uint64_t col0 = 0;
};

static void pack_segment_f16(const double *rows, uint64_t n, uint64_t D, const double *mu, std::vector<uint8_t> &rec,
                             uint64_t rec_bytes, uint64_t cb0, std::vector<float> &cn_all, std::vector<float> &beta_all,
                             HiOnlyOut ho = HiOnlyOut()) {
    const uint64_t nblk = phk_div_up(n, 32);
    const int nchunk = (int)(D / 256);
    phk_parallel_for(nblk, [&, nchunk](uint64_t b) {
        uint8_t *blk = rec.data() + (cb0 + b) * rec_bytes;
        float *cn = reinterpret_cast<float *>(blk + (uint64_t)nchunk * 32 * 1024);
        for (int i = 0; i < 32; ++i) {
            const uint64_t r = b * 32 + i;
            if (r >= n) {  // padding column: zero operand, never selectable
                cn[i] = PAD_V;
                cn[32 + i] = -PAD_V;
                cn_all[(cb0 + b) * 32 + i] = PAD_V;
                beta_all[(cb0 + b) * 32 + i] = -PAD_V;
                if (ho.betah_all) (*ho.betah_all)[(cb0 + b) * 32 + i] = -PAD_V;
                continue;
            }
            // The count-exact kernels multiply the column by the counts MINUS their centre c0 ~ T / D (phk_row_center), so
            // the bias terms are built with mu - 1/D in place of mu: sum_i (c_i - c0) x_i - T [(mu - 1/D).x + ..] =
            // sum_i c_i x_i - T [mu.x + ..] - (c0 - T/D) sum_i x_i, and the last term is the hsum residue.
            double nrm2 = 0.0, mudot = 0.0, mulo = 0.0, lo2 = 0.0, xsum = 0.0, xmx = 0.0;
            const double shift = 1.0 / (double)D;
            for (int c = 0; c < nchunk; ++c)
                for (int h = 0; h < 2; ++h)
                    for (int s = 0; s < 16; ++s)
                        for (int jj = 0; jj < 8; ++jj) {
                            const uint64_t d = 256 * c + 128 * h + 8 * s + jj;
                            const double x = (rows[r * D + d] - mu[d]) * (double)F16_SCALE;
                            _Float16 hi, lo;
                            split_f16(x, hi, lo);
                            const int lane = h * 32 + i;
                            uint8_t *piece = blk + ((uint64_t)c * 32 + 2 * s) * 1024;
                            reinterpret_cast<_Float16 *>(piece + lane * 16)[jj] = hi;
                            reinterpret_cast<_Float16 *>(piece + 1024 + lane * 16)[jj] = lo;
                            const double xt = (double)hi + (double)lo;  // the column as the kernel sees it (scaled)
                            nrm2 += xt * xt;
                            mudot += (mu[d] - shift) * xt;
                            mulo += (mu[d] - shift) * (double)lo;
                            lo2 += (double)lo * (double)lo;
                            xsum += xt;
                            xmx = std::fabs(xt) > xmx ? std::fabs(xt) : xmx;
                            if (ho.lo_rows) (*ho.lo_rows)[(ho.col0 + r) * D + lo_pos(D, d)] = lo;
                        }
            if (ho.betah_all) (*ho.betah_all)[(cb0 + b) * 32 + i] = (float)(mudot - mulo + 0.5 * nrm2 / (double)F16_SCALE);
            if (ho.lonorm) (*ho.lonorm)[ho.col0 + r] = std::sqrt(lo2) / (double)F16_SCALE;
            if (ho.hsum) (*ho.hsum)[ho.col0 + r] = std::fabs(xsum) / (double)F16_SCALE;
            if (ho.xmax) (*ho.xmax)[ho.col0 + r] = xmx / (double)F16_SCALE;
            // count-exact kernel: S beta = S (mu.r~' + |r~'|^2 / 2), the bias per unit of row sum
            cn[32 + i] = (float)(mudot + 0.5 * nrm2 / (double)F16_SCALE);
            beta_all[(cb0 + b) * 32 + i] = cn[32 + i];
            cn[i] = (float)(-0.5 * nrm2);  // already in S^2 units
            cn_all[(cb0 + b) * 32 + i] = cn[i];
        }
    });
}

#define F16H_PIECES 17
#define F16H_BLOCK_BYTES (F16H_PIECES * 1024)
#define F16H_NBUF 3      // LDS ring of block records in phk_knn_f16h_kernel

// ---- the bias as a 17th k-step of the k = 4 sweep (round 5) ----
// w = sum_i (c_i - c0) hi_ji - T b_j.  Rounds 3-4 subtracted T b_j with one v_fma per value after the MFMA chain; one value
// per MFMA and lane, so that fma was a sixth of the kernel's vector instructions.  Now the MFMA does it: piece 16 of a block
// record is one more A fragment, lane (column j, half 0) = [p0 p0 p0 p1 p1 p1 p2 p2], lane (j, half 1) = [p2 0 ..], with
// p0 + p1 + p2 = -b_j 2^e on the 2^-14 grid (29 bits: three float16 pieces hold it exactly, none subnormal), and the query
// side brings [t0 t1 t2 t0 t1 t2 t0 t1 | t2 0 ..], t0 + t1 + t2 = T 2^-e exactly: the nine products sum to -T b~_j.
// e = the model's bias_e.  A padding / masked column carries -65504 (its value can enter no list of a real batch).
__host__ __device__ __forceinline__ void phk_bias_pieces(float bias, double scale, _Float16 (&p)[3]) {
    double x = -(double)bias * scale;
    x = x > -65504.0 ? x : -65504.0;
    x = x < 65504.0 ? x : 65504.0;
    x = rint(x * 16384.0) * (1.0 / 16384.0);
    p[0] = (_Float16)(float)x;
    const double r = x - (double)p[0];
    p[1] = (_Float16)(float)r;
    p[2] = (_Float16)(float)(r - (double)p[1]);
}
// the 16 + 16 bytes of column i in piece 16 of its block record
__host__ __device__ __forceinline__ void phk_bias_piece_store(uint8_t *block, int i, const _Float16 (&p)[3]) {
    _Float16 *lo = reinterpret_cast<_Float16 *>(block + 16 * 1024 + i * 16), *hi = reinterpret_cast<_Float16 *>(block + 16 * 1024 + (32 + i) * 16);
    lo[0] = p[0]; lo[1] = p[0]; lo[2] = p[0]; lo[3] = p[1]; lo[4] = p[1]; lo[5] = p[1]; lo[6] = p[2]; lo[7] = p[2];
    hi[0] = p[2];
    for (int j = 1; j < 8; ++j) hi[j] = (_Float16)0.0f;
}
static int bias_exponent(const std::vector<float> &betah, double *bmax_out) {
    double bmax = 0.0;
    for (float b : betah)
        if (std::fabs((double)b) < 1.0e29) bmax = std::fabs((double)b) > bmax ? std::fabs((double)b) : bmax;   // (not the padding slots)
    *bmax_out = bmax;
    if (!(bmax > 0.0)) return 14;
    int e = (int)std::floor(std::log2(32768.0 / bmax));
    while (e > -20 && bmax * std::ldexp(1.0, e) > 32768.0) --e;
    return e > 14 ? 14 : (e < -20 ? -20 : e);
}
static void write_bias_pieces(std::vector<uint8_t> &rech, const std::vector<float> &betah, uint64_t b0, uint64_t b1, int e) {
    const double scale = std::ldexp(1.0, e);
    for (uint64_t b = b0; b < b1; ++b)
        for (int i = 0; i < 32; ++i) {
            _Float16 p[3];
            phk_bias_pieces(betah[b * 32 + i], scale, p);
            phk_bias_piece_store(rech.data() + b * F16H_BLOCK_BYTES, i, p);
        }
}

int phk_model_build_f16(phk_model *m, const double *pos, const double *neg, const double *cpos,
                        const double *cneg, const double *mu, const double *colnorm) {
    const uint64_t D = m->D;
    const uint64_t nblk = (uint64_t)m->n_rblk_ref + m->n_rblk_pos + m->n_rblk_neg;
    const uint64_t rec_bytes = (D / 256 * 32 + 1) * 1024;
    const uint64_t ncols = m->M + m->n_cpos + m->n_cneg;
    const bool hi_only = true;          // high-parts-only first pass: bias terms, row-major low parts, error tables
    const bool hi_records = D == FAST_D;   // ... and, for the k = 4 kernel, its own 17-piece block records
    std::vector<uint8_t> rec((nblk + 1) * rec_bytes, 0);  // + one block: the DMA prefetch runs one past the end
    std::vector<float> cn_all((nblk + 1) * 32, PAD_V), beta_all((nblk + 1) * 32, -PAD_V), betah_all;
    std::vector<_Float16> lo_rows;
    std::vector<double> lonorm, hsum(ncols, 0.0), xmax(ncols, 0.0);
    HiOnlyOut ho;
    ho.hsum = &hsum;
    ho.xmax = &xmax;
    if (hi_only) {
        betah_all.assign((nblk + 1) * 32, -PAD_V);
        lo_rows.assign(ncols * D, (_Float16)0.0f);
        lonorm.assign(ncols, 0.0);
        ho.betah_all = &betah_all;
        ho.lo_rows = &lo_rows;
        ho.lonorm = &lonorm;
    }
    {
        std::vector<double> train(m->M * D);
        std::copy(pos, pos + m->n_pos * D, train.begin());
        std::copy(neg, neg + m->n_neg * D, train.begin() + m->n_pos * D);
        ho.col0 = 0;
        pack_segment_f16(train.data(), m->M, D, mu, rec, rec_bytes, 0, cn_all, beta_all, ho);
    }
    ho.col0 = m->M;
    if (m->n_cpos) pack_segment_f16(cpos, m->n_cpos, D, mu, rec, rec_bytes, m->n_rblk_ref, cn_all, beta_all, ho);
    ho.col0 = m->M + m->n_cpos;
    if (m->n_cneg) pack_segment_f16(cneg, m->n_cneg, D, mu, rec, rec_bytes, (uint64_t)m->n_rblk_ref + m->n_rblk_pos, cn_all, beta_all, ho);
    m->hsum_train = m->hsum_cen = m->rho_train = m->rho_cen = 0.0;
    for (uint64_t c = 0; c < ncols; ++c) {
        double &dst = c < m->M ? m->hsum_train : m->hsum_cen;
        dst = hsum[c] > dst ? hsum[c] : dst;
        double &rd = c < m->M ? m->rho_train : m->rho_cen;
        const double rr = colnorm[c] > 0.0 ? xmax[c] / colnorm[c] : (xmax[c] > 0.0 ? 1.0 : 0.0);
        rd = rr > rd ? rr : rd;
    }
    m->rho_inf = (m->rho_train > m->rho_cen ? m->rho_train : m->rho_cen) * (1.0 + 1e-6);
    if (hipMalloc(&m->d_Af16, rec.size()) != hipSuccess) return PHK_ERR_NOMEM;
    if (hipMemcpy(m->d_Af16, rec.data(), rec.size(), hipMemcpyHostToDevice) != hipSuccess) return PHK_ERR_HIP;
    if (hipMalloc(&m->d_cn16, cn_all.size() * sizeof(float)) != hipSuccess) return PHK_ERR_NOMEM;
    if (hipMemcpy(m->d_cn16, cn_all.data(), cn_all.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return PHK_ERR_HIP;
    if (hipMalloc(&m->d_beta16, beta_all.size() * sizeof(float)) != hipSuccess) return PHK_ERR_NOMEM;
    if (hipMemcpy(m->d_beta16, beta_all.data(), beta_all.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return PHK_ERR_HIP;
    {   // the train segment's column terms as built (restored when a column mask is lifted)
        const uint64_t ns = (uint64_t)m->n_rblk_ref * 32;
        std::vector<float> orig(3 * ns, 0.0f);
        for (uint64_t i = 0; i < ns; ++i) {
            orig[i] = cn_all[i];
            orig[ns + i] = beta_all[i];
            orig[2 * ns + i] = hi_only ? betah_all[i] : 0.0f;
        }
        if (hipMalloc((void **)&m->d_term_orig, orig.size() * sizeof(float) + 16) != hipSuccess) return PHK_ERR_NOMEM;
        if (hipMemcpy(m->d_term_orig, orig.data(), orig.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return PHK_ERR_HIP;
    }
    if (hi_only) {
        if (hipMalloc((void **)&m->d_betah16, betah_all.size() * sizeof(float)) != hipSuccess) return PHK_ERR_NOMEM;
        if (hipMemcpy(m->d_betah16, betah_all.data(), betah_all.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return PHK_ERR_HIP;
    }
    if (hi_only && hi_records) {
        // 17-piece records: the 16 hi fragments of the full record + one piece of bias terms
        std::vector<uint8_t> &rech = m->h_rech;
        rech.assign((nblk + 2) * F16H_BLOCK_BYTES, 0);   // (two padding records: the sweep's prefetch runs two blocks ahead)
        for (uint64_t b = 0; b < nblk; ++b)
            for (int st = 0; st < 16; ++st)
                memcpy(rech.data() + b * F16H_BLOCK_BYTES + st * 1024, rec.data() + b * rec_bytes + (2 * st) * 1024, 1024);
        m->h_betah = betah_all;
        m->bias_e = bias_exponent(m->h_betah, &m->bias_max);
        write_bias_pieces(rech, m->h_betah, 0, nblk, m->bias_e);
        if (hipMalloc(&m->d_Af16h, rech.size()) != hipSuccess) return PHK_ERR_NOMEM;
        if (hipMemcpy(m->d_Af16h, rech.data(), rech.size(), hipMemcpyHostToDevice) != hipSuccess) return PHK_ERR_HIP;
    }
    if (hi_only) {
        if (hipMalloc((void **)&m->d_lo16, lo_rows.size() * sizeof(_Float16)) != hipSuccess) return PHK_ERR_NOMEM;
        if (hipMemcpy(m->d_lo16, lo_rows.data(), lo_rows.size() * sizeof(_Float16), hipMemcpyHostToDevice) != hipSuccess) return PHK_ERR_HIP;
        // lam_tab: per segment, the largest |lo_j| / S among its columns within a radius, on a 64-step grid of radii
        const uint64_t seg0[4] = {0, m->M, m->M + m->n_cpos, ncols};
        for (int sg = 0; sg < 3; ++sg) {
            double rmin = 1e300, rmax = 0.0;
            for (uint64_t c = seg0[sg]; c < seg0[sg + 1]; ++c) {
                rmin = colnorm[c] < rmin ? colnorm[c] : rmin;
                rmax = colnorm[c] > rmax ? colnorm[c] : rmax;
            }
            if (seg0[sg] == seg0[sg + 1]) rmin = rmax = 0.0;
            m->lam_r0[sg] = rmin;
            m->lam_step[sg] = rmax > rmin ? (rmax - rmin) / 64.0 : 1.0;
            for (int i = 0; i <= 64; ++i) m->lam_tab[sg][i] = 0.0;
            for (uint64_t c = seg0[sg]; c < seg0[sg + 1]; ++c) {
                int i = (int)std::ceil((colnorm[c] - rmin) / m->lam_step[sg] - 1e-12);
                i = i < 0 ? 0 : (i > 64 ? 64 : i);
                if (lonorm[c] > m->lam_tab[sg][i]) m->lam_tab[sg][i] = lonorm[c];
            }
            for (int i = 1; i <= 64; ++i)
                if (m->lam_tab[sg][i - 1] > m->lam_tab[sg][i]) m->lam_tab[sg][i] = m->lam_tab[sg][i - 1];
        }
    }
    return PHK_OK;
}

// Replace the centroid segments of a built model (same counts): block records, column terms, low parts, error tables.
// colnorm_c: |r'| of the new centroids (positive then negative), computed by the caller as at build time.
int phk_model_update_centroids_f16(phk_model *m, const double *cpos, const double *cneg, const double *colnorm_c) {
    const uint64_t D = m->D;
    const uint64_t nbc = (uint64_t)m->n_rblk_pos + m->n_rblk_neg, ncc = m->n_cpos + m->n_cneg;
    const uint64_t rec_bytes = (D / 256 * 32 + 1) * 1024;
    const bool hi_only = m->d_lo16 != nullptr;
    std::vector<uint8_t> rec(nbc * rec_bytes, 0);
    std::vector<float> cn_all(nbc * 32, PAD_V), beta_all(nbc * 32, -PAD_V), betah_all(nbc * 32, -PAD_V);
    std::vector<_Float16> lo_rows;
    std::vector<double> lonorm, hsum(ncc, 0.0), xmax(ncc, 0.0);
    HiOnlyOut ho;
    ho.hsum = &hsum;
    ho.xmax = &xmax;
    if (hi_only) {
        lo_rows.assign(ncc * D, (_Float16)0.0f);
        lonorm.assign(ncc, 0.0);
        ho.betah_all = &betah_all;
        ho.lo_rows = &lo_rows;
        ho.lonorm = &lonorm;
    }
    ho.col0 = 0;
    pack_segment_f16(cpos, m->n_cpos, D, m->h_mu.data(), rec, rec_bytes, 0, cn_all, beta_all, ho);
    ho.col0 = m->n_cpos;
    pack_segment_f16(cneg, m->n_cneg, D, m->h_mu.data(), rec, rec_bytes, m->n_rblk_pos, cn_all, beta_all, ho);
    m->hsum_cen = m->rho_cen = 0.0;
    for (uint64_t c = 0; c < ncc; ++c) {
        m->hsum_cen = hsum[c] > m->hsum_cen ? hsum[c] : m->hsum_cen;
        const double rr = colnorm_c[c] > 0.0 ? xmax[c] / colnorm_c[c] : (xmax[c] > 0.0 ? 1.0 : 0.0);
        m->rho_cen = rr > m->rho_cen ? rr : m->rho_cen;
    }
    m->rho_inf = (m->rho_train > m->rho_cen ? m->rho_train : m->rho_cen) * (1.0 + 1e-6);
    const uint64_t b0 = m->n_rblk_ref;
    if (hipMemcpy((uint8_t *)m->d_Af16 + b0 * rec_bytes, rec.data(), rec.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->d_cn16 + b0 * 32, cn_all.data(), cn_all.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->d_beta16 + b0 * 32, beta_all.data(), beta_all.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return PHK_ERR_HIP;
    if (hi_only) {
        if (m->d_Af16h) {
            // the centroid blocks of the host copy, then the bias pieces: of the centroid blocks alone while the model's
            // exponent still fits the new biases, else of every block (NOTE: the train blocks then carry the biases as
            // built -- a column mask is re-applied by the caller's next phk_model_set_column_mask / apply_mask)
            std::vector<uint8_t> &rech = m->h_rech;
            for (uint64_t b = 0; b < nbc; ++b)
                for (int st = 0; st < 16; ++st)
                    memcpy(rech.data() + (b0 + b) * F16H_BLOCK_BYTES + st * 1024, rec.data() + b * rec_bytes + (2 * st) * 1024, 1024);
            for (uint64_t i = 0; i < nbc * 32; ++i) m->h_betah[b0 * 32 + i] = betah_all[i];
            const int e = bias_exponent(m->h_betah, &m->bias_max);
            const uint64_t nblk = b0 + nbc;
            if (e != m->bias_e) {
                m->bias_e = e;
                write_bias_pieces(rech, m->h_betah, 0, nblk, e);
                if (hipMemcpy(m->d_Af16h, rech.data(), nblk * F16H_BLOCK_BYTES, hipMemcpyHostToDevice) != hipSuccess) return PHK_ERR_HIP;
            } else {
                write_bias_pieces(rech, m->h_betah, b0, nblk, e);
                if (hipMemcpy((uint8_t *)m->d_Af16h + b0 * F16H_BLOCK_BYTES, rech.data() + b0 * F16H_BLOCK_BYTES, nbc * F16H_BLOCK_BYTES,
                              hipMemcpyHostToDevice) != hipSuccess)
                    return PHK_ERR_HIP;
            }
        }
        if (hipMemcpy(m->d_betah16 + b0 * 32, betah_all.data(), betah_all.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(m->d_lo16 + m->M * D, lo_rows.data(), lo_rows.size() * sizeof(_Float16), hipMemcpyHostToDevice) != hipSuccess)
            return PHK_ERR_HIP;
        const uint64_t seg0[3] = {0, m->n_cpos, ncc};
        for (int sg = 1; sg <= 2; ++sg) {
            double rmin = 1e300, rmax = 0.0;
            for (uint64_t c = seg0[sg - 1]; c < seg0[sg]; ++c) {
                rmin = colnorm_c[c] < rmin ? colnorm_c[c] : rmin;
                rmax = colnorm_c[c] > rmax ? colnorm_c[c] : rmax;
            }
            m->lam_r0[sg] = rmin;
            m->lam_step[sg] = rmax > rmin ? (rmax - rmin) / 64.0 : 1.0;
            for (int i = 0; i <= 64; ++i) m->lam_tab[sg][i] = 0.0;
            for (uint64_t c = seg0[sg - 1]; c < seg0[sg]; ++c) {
                int i = (int)std::ceil((colnorm_c[c] - rmin) / m->lam_step[sg] - 1e-12);
                i = i < 0 ? 0 : (i > 64 ? 64 : i);
                if (lonorm[c] > m->lam_tab[sg][i]) m->lam_tab[sg][i] = lonorm[c];
            }
            for (int i = 1; i <= 64; ++i)
                if (m->lam_tab[sg][i - 1] > m->lam_tab[sg][i]) m->lam_tab[sg][i] = m->lam_tab[sg][i - 1];
        }
    }
    return PHK_OK;
}

// column mask -> the train segment's column terms: a masked column carries the terms of a padding column (its value
// can never enter a list), an unmasked one the terms it was built with
__global__ __launch_bounds__(256) void phk_mask_terms_kernel(const uint8_t *__restrict__ mask, uint64_t M, uint64_t nslots,
                                                             const float *__restrict__ orig, uint8_t *__restrict__ rec,
                                                             uint64_t rec_bytes, uint64_t term_off, uint8_t *__restrict__ rech,
                                                             float *__restrict__ cn16, float *__restrict__ beta16,
                                                             float *__restrict__ betah16, double bias_scale) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    const bool masked = s < M && mask && mask[s];
    const uint64_t b = s / 32, i = s % 32;
    const float cn = masked ? PAD_V : orig[s], be = masked ? -PAD_V : orig[nslots + s], bh = masked ? -PAD_V : orig[2 * nslots + s];
    float *t = reinterpret_cast<float *>(rec + b * rec_bytes + term_off);
    t[i] = cn;
    t[32 + i] = be;
    cn16[s] = cn;
    beta16[s] = be;
    if (betah16) betah16[s] = bh;
    if (rech) {
        _Float16 p[3];
        phk_bias_pieces(bh, bias_scale, p);
        phk_bias_piece_store(rech + b * F16H_BLOCK_BYTES, (int)i, p);
    }
}

int phk_model_apply_mask_f16(phk_ctx *ctx, phk_model *m) {
    const uint64_t ns = (uint64_t)m->n_rblk_ref * 32;
    if (ns == 0) return PHK_OK;
    const uint64_t rec_bytes = (m->D / 256 * 32 + 1) * 1024;
    PHK_LAUNCH(ctx, "phk_mask_terms_kernel",
               phk_mask_terms_kernel<<<dim3((unsigned)phk_div_up(ns, 256)), dim3(256), 0, ctx->stream>>>(
                   m->has_mask ? m->d_col_mask : nullptr, m->M, ns, m->d_term_orig, (uint8_t *)m->d_Af16, rec_bytes,
                   (m->D / 256) * 32 * 1024, (uint8_t *)m->d_Af16h, m->d_cn16, m->d_beta16, m->d_betah16, std::ldexp(1.0, m->bias_e)));
    return PHK_OK;
}

// ------------------------------------------------------------------------------------
// device
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void f16_split8(const float (&x)[8], half8 &hi, half8 &lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 h = (_Float16)x[j];
        hi[j] = h;
        lo[j] = (_Float16)(x[j] - (float)h);
    }
}

// Workgroup = 4 waves = 128 queries; two workgroups per CU (one wave of each per SIMD).
#define F16_WAVES 4
template <int SRC>
__global__ __launch_bounds__(256, 2) void phk_knn_f16_kernel(const void *__restrict__ src,
                                                             const uint32_t *__restrict__ rowsum, uint64_t N,
                                                             const uint4 *__restrict__ Af,
                                                             const float *__restrict__ mu32,
                                                             const double *__restrict__ mu64,
                                                             uint32_t nblk_ref, uint32_t nblk_pos,
                                                             uint32_t nblk_neg,
                                                             float *__restrict__ cand_v,
                                                             uint32_t *__restrict__ cand_i,
                                                             float *__restrict__ cand_u,
                                                             const uint32_t *__restrict__ qmap,
                                                             const uint32_t *__restrict__ qcount,
                                                             uint64_t set_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 2 x F16_BLOCK_BYTES
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const uint64_t q0 = ((uint64_t)blockIdx.x * F16_WAVES + wave) * 32;
    // "second chance" launches (phk_score_fast): the queries are rows qmap[0 .. *qcount) of `src`, the lists are
    // written at the dense positions 0 .. *qcount (list stride N = the capacity of the second list set); the grid is
    // sized for N, so workgroups wholly past the device-side count leave at once (before any DMA / barrier)
    const uint64_t Nlist = N;
    if (qmap) {
        const uint64_t cnt = phk_uniform_load(qcount);
        N = cnt < PHK_SECOND_MIN ? 0 : (cnt < N ? cnt : N);   // a handful of rows: left to the brute force
        if ((uint64_t)blockIdx.x * F16_WAVES * 32 >= N) return;
    }
    // NB: every wave of the workgroup takes part in the DMA + barriers even if its queries are padding
    const uint64_t qpos = (q0 + j < N) ? q0 + j : N - 1;
    const uint64_t qrow = qmap ? (uint64_t)qmap[qpos] : qpos;
    // Column split (gridDim.y > 1; the second-chance launches, whose few thousand queries fill a fraction of the chip):
    // workgroup row y sweeps the y-th part of the train blocks -- the last one the centroid blocks as well, which follow
    // the train blocks in the record array -- and writes list set y (set_bytes apart); phk_merge_list_sets_kernel folds
    // the sets into set 0.
    uint32_t col0 = 0;
    if (gridDim.y > 1) {
        const uint32_t S = gridDim.y, y = blockIdx.y;
        const uint32_t b0 = (uint32_t)((uint64_t)nblk_ref * y / S), b1 = (uint32_t)((uint64_t)nblk_ref * (y + 1) / S);
        Af += (uint64_t)b0 * (F16_BLOCK_BYTES / 16);
        col0 = 32u * b0;
        nblk_ref = b1 - b0;
        if (y != S - 1) nblk_pos = nblk_neg = 0;
        cand_v = reinterpret_cast<float *>(reinterpret_cast<char *>(cand_v) + y * set_bytes);
        cand_i = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(cand_i) + y * set_bytes);
        cand_u = reinterpret_cast<float *>(reinterpret_cast<char *>(cand_u) + y * set_bytes);
    }
    const uint32_t total = nblk_ref + nblk_pos + nblk_neg;
    const uint32_t seg_end0 = nblk_ref, seg_end1 = nblk_ref + nblk_pos;

    // one block record -> LDS buffer `buf`: 33 pieces, piece p by wave p % 8.  The LDS-DMA is issued
    // from inline asm: a builtin DMA makes hipcc put s_waitcnt vmcnt(0) in front of the next ds_read
    // (it cannot tell the two LDS buffers apart), which would serialise the prefetch with the MFMAs.
    // The asm loads are invisible to hipcc's counters; they are drained by the explicit vmcnt(0) in
    // front of the barrier below (M0 = LDS byte address of the piece, written in the same statement).
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem;
    auto dma_block = [&](uint32_t blk, int buf) {
        const uint4 *g = Af + (uint64_t)blk * (F16_BLOCK_BYTES / 16) + lane;
        const uint32_t l = lds_base + (uint32_t)buf * F16_BLOCK_BYTES;
        for (int p = wave; p < F16_PIECES; p += F16_WAVES) {
            const uint4 *gp = g + p * 64;
            const uint32_t lp = __builtin_amdgcn_readfirstlane(l + (uint32_t)p * 1024u);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(lp) : "memory");
        }
    };
    if (total) dma_block(0, 0);

    // ---- prologue: this lane's scaled, split query elements b[s] (dims 128h + 8s .. +7) ----
    half8 bh[16], bl[16];
    if (SRC == 0) {
        const uint4 *row = reinterpret_cast<const uint4 *>(static_cast<const uint32_t *>(src) + qrow * FAST_D + 128 * h);
        uint32_t tot;
        if (rowsum) {
            tot = rowsum[qrow];  // the count kernel's row sums: saves a pass over the counts
        } else {
            uint32_t sum = 0;
#pragma unroll
            for (int g = 0; g < 32; ++g) {
                const uint4 c = row[g];
                sum += c.x + c.y + c.z + c.w;
            }
            tot = sum + __shfl_xor(sum, 32);
        }
        const float inv = (float)(1.0 / (double)tot) * F16_SCALE;
        const float4 *mp = reinterpret_cast<const float4 *>(mu32 + 128 * h);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            // (every second k-step the row and centre addresses are made to depend on the fragments of two k-steps before: hoisted
            // to the top, all 64 loads beside the 128 fragment registers spilled 24 registers)
            uint64_t ro = 0;   // (an offset of 0 that only exists once the fragments of two k-steps ago do)
            if (s >= 2 && (s & 1) == 0) asm volatile("" : "+v"(ro) : "v"(bh[s - 2]), "v"(bl[s - 2]));
            const uint4 *rw = row + ro;
            const float4 *mw = mp + ro;
            const uint4 c0 = rw[2 * s], c1 = rw[2 * s + 1];
            const float4 m0 = mw[2 * s], m1 = mw[2 * s + 1];
            float x[8];
            x[0] = fmaf((float)c0.x, inv, -m0.x * F16_SCALE);
            x[1] = fmaf((float)c0.y, inv, -m0.y * F16_SCALE);
            x[2] = fmaf((float)c0.z, inv, -m0.z * F16_SCALE);
            x[3] = fmaf((float)c0.w, inv, -m0.w * F16_SCALE);
            x[4] = fmaf((float)c1.x, inv, -m1.x * F16_SCALE);
            x[5] = fmaf((float)c1.y, inv, -m1.y * F16_SCALE);
            x[6] = fmaf((float)c1.z, inv, -m1.z * F16_SCALE);
            x[7] = fmaf((float)c1.w, inv, -m1.w * F16_SCALE);
            f16_split8(x, bh[s], bl[s]);
        }
    } else {
        const double2 *row = reinterpret_cast<const double2 *>(static_cast<const double *>(src) + qrow * FAST_D + 128 * h);
        const double2 *mp = reinterpret_cast<const double2 *>(mu64 + 128 * h);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float x[8];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double2 c = row[4 * s + t], mm = mp[4 * s + t];
                x[2 * t + 0] = (float)((c.x - mm.x) * (double)F16_SCALE);
                x[2 * t + 1] = (float)((c.y - mm.y) * (double)F16_SCALE);
            }
            // (the float values are made opaque: hipcc otherwise folds double -> float -> half into ONE double -> half
            // conversion -- legal, 24 >= 2 * 11 + 2 bits -- for which gfx950 has no instruction: ~45 instructions of integer
            // arithmetic per element, a 6 000-line prologue and 276 spilled registers; v_cvt_f32_f64 + v_cvt_f16_f32 are two)
#pragma unroll
            for (int e = 0; e < 8; ++e) asm("" : "+v"(x[e]));
            f16_split8(x, bh[s], bl[s]);
        }
    }

    float lv[CAND];
    uint32_t li[CAND];
    float ldrop = -3.0e38f;
#pragma unroll
    for (int c = 0; c < CAND; ++c) {
        lv[c] = -3.0e38f;
        li[c] = 0xFFFFFFFFu;
    }
    int seg = 0;
    uint32_t seg_first = 0;
    // list flush / reset once column block `b` (the last of its segment) has been inserted
    auto flush_if_segment_end = [&](uint32_t b) {
        while (seg < NSEG && b + 1 == (seg == 0 ? seg_end0 : seg == 1 ? seg_end1 : total)) {
            if (q0 + j < N) {
                const uint32_t o = seg == 0 ? col0 : 0u;   // (a column split: indices relative to the whole train segment)
                cand_store(cand_v, cand_i, cand_u, seg, h, q0 + j, Nlist, lv[0], lv[1], lv[2], lv[3],
                           li[0] == 0xFFFFFFFFu ? li[0] : li[0] + o, li[1] == 0xFFFFFFFFu ? li[1] : li[1] + o,
                           li[2] == 0xFFFFFFFFu ? li[2] : li[2] + o, li[3] == 0xFFFFFFFFu ? li[3] : li[3] + o, ldrop);
            }
#pragma unroll
            for (int c = 0; c < CAND; ++c) {
                lv[c] = -3.0e38f;
                li[c] = 0xFFFFFFFFu;
            }
            ldrop = -3.0e38f;
            ++seg;
            seg_first = b + 1;
        }
    };

    // leading segments without columns (method 'kmeans' sweeps no train rows): empty lists; the flush rule above
    // only fires at the end of a block
    while (seg < NSEG && (seg == 0 ? seg_end0 : seg == 1 ? seg_end1 : total) == 0) {
        if (q0 + j < N) cand_store_empty(cand_v, cand_i, cand_u, seg, h, q0 + j, Nlist);
        ++seg;
    }

    // Software pipeline inside the wave: while block blk's 48 MFMAs run on the matrix pipe, the VALU
    // inserts the 16 values of block blk-1 (one branch-free insertion per k-step, i.e. per 3 MFMAs),
    // so the epilogue costs no separate phase.  xs = values of the previous block (acc + norm term);
    // before the first block it is -3e38, which no list accepts.
    f32x16 xs;
#pragma unroll
    for (int r = 0; r < 16; ++r) xs[r] = -3.0e38f;
    uint32_t cbase = 4u * (uint32_t)h;  // index base of the block held in xs
    for (uint32_t blk = 0; blk < total; ++blk) {
        // block blk has landed (every wave waits for its own pieces, then the barrier), and every wave
        // is done reading the other buffer, which the next DMA overwrites
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        dma_block(blk + 1, (blk + 1) & 1);  // one past the end on the last block: the record array is padded
        const uint8_t *buf = smem + (blk & 1) * F16_BLOCK_BYTES;

        // Two accumulators: the hi.hi chain, and the two cross terms (hi.lo + lo.hi), whose running sums are 2^-10 of the
        // first one's.  Every instruction of a chain is charged on that chain's running sum (ErrBound), so the 32 cross
        // instructions cost next to nothing there instead of as much as the 16 hi.hi ones.
        f32x16 acc, accx;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = accx[r] = 0.0f;
        const half8 *fr = reinterpret_cast<const half8 *>(buf) + lane;
        // fragments are read one k-step ahead of their MFMAs (LDS latency ~ one step of MFMA time);
        // the issue order per step is pinned: 2 LDS reads, then MFMA / insertion-VALU interleaved
        half8 ahn = fr[0], aln = fr[64];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const half8 ah = ahn, al = aln;
            if (s < 15) {
                ahn = fr[(2 * s + 2) * 64];
                aln = fr[(2 * s + 3) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);  // the reads stay up here, one step ahead of their MFMAs
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[s], acc, 0, 0, 0);
            accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[s], accx, 0, 0, 0);
            accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[s], accx, 0, 0, 0);
            list_insert(lv, li, ldrop, xs[s], cbase + (uint32_t)((s & 3) + 8 * (s >> 2)));
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);  // VALU (insertion)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        }
        if (blk > 0) flush_if_segment_end(blk - 1);
        // D[i][j]: register r of lane (j, h') holds column row i = (r&3) + 8(r>>2) + 4h'; add its norm term
        // now (the buffer is recycled after the next barrier)
        const float4 *cn = reinterpret_cast<const float4 *>(buf + 32 * 1024) + h;
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) {
            const float4 c4 = cn[2 * m4];
            xs[4 * m4 + 0] = (acc[4 * m4 + 0] + accx[4 * m4 + 0]) + c4.x;
            xs[4 * m4 + 1] = (acc[4 * m4 + 1] + accx[4 * m4 + 1]) + c4.y;
            xs[4 * m4 + 2] = (acc[4 * m4 + 2] + accx[4 * m4 + 2]) + c4.z;
            xs[4 * m4 + 3] = (acc[4 * m4 + 3] + accx[4 * m4 + 3]) + c4.w;
        }
        cbase = 32u * (blk - seg_first) + 4u * (uint32_t)h;
    }
    if (total) {  // the last block's values
#pragma unroll
        for (int r = 0; r < 16; ++r) list_insert(lv, li, ldrop, xs[r], cbase + (uint32_t)((r & 3) + 8 * (r >> 2)));
        flush_if_segment_end(total - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the trailing prefetch before the LDS is released
    for (; seg < NSEG; ++seg) {
        if (q0 + j < N) {
            cand_store_empty(cand_v, cand_i, cand_u, seg, h, q0 + j, Nlist);
        }
    }
}

// Folds the list sets 1 .. S-1 of a column-split launch into set 0: per (query, half-list) the 4 S train candidates
// are re-inserted into one list (the best value dropped = the largest any part dropped or the merge drops); the centroid
// segments, swept by the last part alone, are copied.
// qcount == NULL: all Nlist queries (the column groups of the general-D sweep); ca: the sets' observed running sums
// ([2][Nlist] floats each, ca_set floats apart), folded by maximum.
__global__ __launch_bounds__(256) void phk_merge_list_sets_kernel(float *__restrict__ cv, uint32_t *__restrict__ ci,
                                                                  float *__restrict__ cu, uint64_t Nlist, uint64_t set_bytes,
                                                                  int S, const uint32_t *__restrict__ qcount,
                                                                  float *__restrict__ ca, uint64_t ca_set) {
    uint64_t cnt = Nlist;
    if (qcount) {
        const uint64_t cnt_all = phk_uniform_load(qcount);
        cnt = cnt_all < PHK_SECOND_MIN ? 0 : (cnt_all < Nlist ? cnt_all : Nlist);
    }
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t q = id >> 1;
    const int h = (int)(id & 1);
    if (q >= cnt) return;
    if (ca) {
        float a = ca[(uint64_t)h * Nlist + q];
        for (int y = 1; y < S; ++y) a = fmaxf(a, ca[(uint64_t)y * ca_set + (uint64_t)h * Nlist + q]);
        ca[(uint64_t)h * Nlist + q] = a;
    }
    auto set_v = [&](int y) { return reinterpret_cast<const float *>(reinterpret_cast<const char *>(cv) + (uint64_t)y * set_bytes); };
    auto set_i = [&](int y) { return reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(ci) + (uint64_t)y * set_bytes); };
    auto set_u = [&](int y) { return reinterpret_cast<const float *>(reinterpret_cast<const char *>(cu) + (uint64_t)y * set_bytes); };
    float lv[CAND] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    uint32_t li[CAND] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    float drop = -3.0e38f;
    // Eight sets at a time, every load of the eight requested before the first insertion (and the centroid segments' copies
    // with them): as load - insert - load the kernel was a chain of 8 dependent round trips per thread, 33 us for the ~5 000
    // queued rows of configs[1] -- most of it waiting.
    float cv1[NSEG - 1][CAND], cu1[NSEG - 1];
    uint32_t ci1[NSEG - 1][CAND];
#pragma unroll
    for (int seg = 1; seg < NSEG; ++seg) {
#pragma unroll
        for (int c = 0; c < CAND; ++c) {
            cv1[seg - 1][c] = set_v(S - 1)[cand_at(seg, h, c, q, Nlist)];
            ci1[seg - 1][c] = set_i(S - 1)[cand_at(seg, h, c, q, Nlist)];
        }
        cu1[seg - 1] = set_u(S - 1)[candu_at(seg, h, q, Nlist)];
    }
    for (int y0 = 0; y0 < S; y0 += 8) {
        float v[8][CAND], u[8];
        uint32_t ix[8][CAND];
#pragma unroll
        for (int yy = 0; yy < 8; ++yy) {
            const int y = y0 + yy < S ? y0 + yy : S - 1;   // (a set past the end re-reads the last one: no load under a condition)
            u[yy] = set_u(y)[candu_at(0, h, q, Nlist)];
#pragma unroll
            for (int c = 0; c < CAND; ++c) {
                ix[yy][c] = set_i(y)[cand_at(0, h, c, q, Nlist)];
                v[yy][c] = set_v(y)[cand_at(0, h, c, q, Nlist)];
            }
        }
#pragma unroll
        for (int yy = 0; yy < 8; ++yy) {
            if (y0 + yy >= S) break;
            drop = fmaxf(drop, u[yy]);
#pragma unroll
            for (int c = 0; c < CAND; ++c)
                if (ix[yy][c] != 0xFFFFFFFFu) list_insert(lv, li, drop, v[yy][c], ix[yy][c]);
        }
    }
    cand_store(cv, ci, cu, 0, h, q, Nlist, lv[0], lv[1], lv[2], lv[3], li[0], li[1], li[2], li[3], drop);
#pragma unroll
    for (int seg = 1; seg < NSEG; ++seg) {
#pragma unroll
        for (int c = 0; c < CAND; ++c) {
            cv[cand_at(seg, h, c, q, Nlist)] = cv1[seg - 1][c];
            ci[cand_at(seg, h, c, q, Nlist)] = ci1[seg - 1][c];
        }
        cu[candu_at(seg, h, q, Nlist)] = cu1[seg - 1];
    }
}

// splits > 1 (second-chance launches only): `splits` list sets, set_bytes apart, merged into the first
int phk_launch_proposal_f16(phk_ctx *ctx, const phk_model *m, const void *src, bool src_counts,
                            const uint32_t *d_rowsum, uint64_t nb, uint32_t nref, uint32_t npos, uint32_t nneg,
                            float *cv, uint32_t *ci, float *cu, const uint32_t *qmap, const uint32_t *qcount, int splits,
                            uint64_t set_bytes) {
    const size_t lds = 2 * F16_BLOCK_BYTES;
    if (splits < 1 || !qmap || (uint32_t)splits > nref) splits = 1;
    const uint4 *af = (const uint4 *)m->d_Af16 + (uint64_t)(nref ? 0 : m->n_rblk_ref) * (F16_BLOCK_BYTES / 16);
    const unsigned gblocks = (unsigned)phk_div_up(nb, 32 * F16_WAVES);
    if (src_counts) {
        PHK_LAUNCH(ctx, "phk_knn_f16_kernel",
                   phk_knn_f16_kernel<0><<<dim3(gblocks, (unsigned)splits), dim3(64 * F16_WAVES), lds, ctx->stream>>>(
                       src, d_rowsum, nb, af, m->d_mu32, m->d_mu64, nref, npos, nneg, cv, ci, cu, qmap, qcount, set_bytes));
    } else {
        PHK_LAUNCH(ctx, "phk_knn_f16_kernel",
                   phk_knn_f16_kernel<1><<<dim3(gblocks, (unsigned)splits), dim3(64 * F16_WAVES), lds, ctx->stream>>>(
                       src, nullptr, nb, af, m->d_mu32, m->d_mu64, nref, npos, nneg, cv, ci, cu, qmap, qcount, set_bytes));
    }
    if (splits > 1) {
        PHK_LAUNCH(ctx, "phk_merge_list_sets_kernel",
                   phk_merge_list_sets_kernel<<<dim3((unsigned)phk_div_up(2 * nb, 256)), dim3(256), 0, ctx->stream>>>(
                       cv, ci, cu, nb, set_bytes, splits, qcount, nullptr, 0));
    }
    return PHK_OK;
}

// ====================================================================================
// Count-exact proposal (k = 4, queries given as uint32 counts): the idea the kernel below builds on.
//
// The query operand is the count vector ITSELF: integers <= 2048 are exact in fp16, so no split and no
// normalisation are needed on the query side and a k-step costs 2 MFMAs (c.r_hi, c.r_lo) instead of 3.
// With T = row sum:   sum_i c_i (r_hi + r_lo)_ji = T S (q . r~'_j)   and
//     w_j = acc_j - T * [S (mu . r~'_j + |r~'_j|^2 / 2)]  =  T S (q'. r~'_j - |r~'_j|^2 / 2)  =  T S v_j ,
// the same ranking quantity in per-row units (the decision stage divides by T S).  A row with a count
// above 2048 (a contig of hundreds of kb) is handed to the exact brute-force queue: its lists are written empty.
//
// Candidate lists without index registers in the hot loop: the low 5 mantissa bits of a value carry
// (r << 1) | fresh, r = which of the lane's 16 rows of the block it is, fresh = inserted during this
// block.  The sorted 5-deep value list (4 candidates + the best dropped value) is maintained with
// v_med3 only; once per block the block number is shifted into the id list at the positions whose
// fresh bit is set, and the bits are cleared (clearing keeps the list sorted: fresh is the lowest bit).
// The 31-ulp perturbation is part of the decision stage's error bound.
// ====================================================================================
#define CX_SENT 0x03FFFFFFu

// ====================================================================================
// High-parts-only proposal kernel (k = 4, uint32 counts; the default first pass).
//
// The count-exact kernel above spends 2 MFMAs per k-step because the reference column is split in two fp16 numbers
// (r' S = hi + lo).  The lo product only carries the last 11 of 22 bits; it matters for a handful of columns per query
// -- those whose value lies within the error of the hi product of the decisive (need-th) value.  This kernel runs the
// hi product ALONE (1 MFMA per k-step, block records of 16 hi fragments + one piece of bias terms) and the decision
// stage (phk_decide_h_kernel) adds the lo product, sum_i (c_i - T mu_i) lo_ji in float64, to just those candidates:
//     w^h_j = sum_i c_i hi_ji - T [S mu.hi_j + S |r~'_j|^2 / 2]      (this kernel),
//     w_j   = w^h_j + sum_i (c_i - T mu_i) lo_ji                     (= the count-exact kernel's value, refined later).
// |w_j - w^h_j| <= T S |q'| |lo_j| / S by Cauchy-Schwarz: that is the window the decision stage refines inside.
// List maintenance, index bits in the value, the two-tile ping-pong and the segment pipeline are those of the
// count-exact kernel; with one MFMA per value the loop would be bound by the VALU issue of the insertions (7 operations
// per value), so an insertion no lane of the wave needs is skipped (see insert()).  In-kernel cycle stamps per block
// iteration of a wave (32 MFMAs): 83 % of the time in the k-step loop, 8 % issuing its 5 DMA pieces, 4 % at the barrier,
// 4 % between iterations; by the counters the matrix pipe is 48 % busy at the 2.15 GHz the chip holds under this kernel.
// ====================================================================================
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void phk_knn_f16h_kernel(
    const uint32_t *__restrict__ counts, const uint32_t *__restrict__ rowsum, uint64_t N, const uint4 *__restrict__ Af,
    uint32_t nblk_ref, uint32_t nblk_pos, uint32_t nblk_neg, float *__restrict__ cand_v, uint32_t *__restrict__ cand_i,
    float *__restrict__ cand_u, float tscale /* 2^-bias_e */) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // F16H_NBUF x F16H_BLOCK_BYTES
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const uint64_t q0 = ((uint64_t)blockIdx.x * NW + wave) * (32 * NT);
    const uint32_t total = nblk_ref + nblk_pos + nblk_neg;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    // One block record -> LDS buffer `buf`, as 16 / NW fragment pieces + the bias terms per wave (32 floats of the 17th piece;
    // every wave fetches the same 256 bytes, so that each wave has issued exactly 16 / NW + 1 DMAs per block and can count
    // them).  Scalar base + the lane's offset: the address of a piece costs two scalar additions.  piece(k): the k-th of
    // this wave's requests for a block -- they are issued one at a time between the MFMAs of the k-step loop, not as a
    // burst behind the barrier (in-kernel stamps of the burst version: 8 % of a block iteration spent issuing it).
    constexpr int NPW = 16 / NW + 1;
    const uint32_t lane4 = (uint32_t)lane * 4u;
    auto dma_piece = [&](uint32_t blk, uint32_t buf, int k) {
        const char *g = reinterpret_cast<const char *>(Af) + (uint64_t)blk * F16H_BLOCK_BYTES;
        const uint32_t l = lds_base + buf * (uint32_t)F16H_BLOCK_BYTES;
        if (k < 16 / NW) {
            const int p = wave + NW * k;
            const char *gp = g + p * 1024;
            const uint32_t lp = __builtin_amdgcn_readfirstlane(l + (uint32_t)p * 1024u);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(gp), "s"(lp) : "memory");
        } else {   // piece 16 (the bias fragment): 256 bytes of it per wave
            const uint32_t quarter = 256u * ((uint32_t)wave & 3u);
            const char *gp = g + 16 * 1024 + quarter;
            const uint32_t lp = __builtin_amdgcn_readfirstlane(l + 16u * 1024u + quarter);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(lane4), "s"(gp), "s"(lp) : "memory");
        }
    };
    // the ring is F16H_NBUF = 3 buffers deep: block g + 2 is requested while block g computes, and a wave waits for ITS
    // pieces of block g with s_waitcnt vmcnt(NPW) -- those of block g + 1 may still be in flight (with two buffers and
    // vmcnt(0) a block had one iteration, ~1.2 us, to arrive: an L2 / fabric round trip under load)
    if (total) {
#pragma unroll
        for (int k = 0; k < NPW; ++k) dma_piece(0, 0, k);
#pragma unroll
        for (int k = 0; k < NPW; ++k) dma_piece(1, 1, k);   // (one past the end on a one-block sweep: the record array is padded)
    }

    // ---- prologue: centred counts -> fp16 (exact up to 2048 in magnitude), row sum (from the caller) ----
    half8 bq[NT][16];
    half8 bqx[NT];      // the 17th k-step's operand: the pieces of T 2^-e (see phk_bias_pieces)
    bool big[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint64_t qi = q0 + 32 * t + j;
        const uint64_t qrow = qi < N ? qi : N - 1;
        const uint4 *row = reinterpret_cast<const uint4 *>(counts + qrow * FAST_D + 128 * h);
        // the operand is the count minus the row's centre (phk_row_center): an integer of magnitude <= 2048, exact in fp16
        const uint32_t tot = rowsum[qrow];
        const int cen = (int)phk_row_center(tot, FAST_D);
        // T 2^-e = t0 + t1 + t2 exactly, or the row is `big` (a row sum the pieces cannot carry: beyond 65504 x 2^e): the three
        // bit fields of T, each at most 11 bits wide, times the power of two -- exact in float32 and, inside the float16 range
        // (the lowest non-zero piece is >= 2^-e >= 2^-14), in float16.  (Formed before the row is read: nothing of it but the
        // four registers of bqx stays live.)
        bool fits;
        {
            const float f0 = (float)(tot & 0xFFE00000u) * tscale, f1 = (float)(tot & 0x001FFC00u) * tscale, f2 = (float)(tot & 0x000003FFu) * tscale;
            fits = fmaxf(fmaxf(f0, f1), f2) < 65504.0f;
            const _Float16 z16 = (_Float16)0.0f;
            const _Float16 tp[3] = {fits ? (_Float16)f0 : z16, fits ? (_Float16)f1 : z16, fits ? (_Float16)f2 : z16};
            if (h == 0) bqx[t] = half8{tp[0], tp[1], tp[2], tp[0], tp[1], tp[2], tp[0], tp[1]};
            else bqx[t] = half8{tp[2], z16, z16, z16, z16, z16, z16, z16};
        }
        __builtin_amdgcn_sched_barrier(0);
        uint32_t mx = 0;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const uint4 c0 = row[2 * s], c1 = row[2 * s + 1];
            const uint32_t c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = (int)c[e] - cen;           // (a count above 2^31 wraps to a negative d: the row is `big` below)
                const uint32_t ad = (uint32_t)(d < 0 ? -d : d);
                mx = max(mx, max(ad, c[e] >> 31 ? 0xFFFFFFFFu : 0u));
                bq[t][s][e] = (_Float16)(float)(d < -2048 ? -2048 : (d > 2048 ? 2048 : d));
            }
        }
        const uint32_t mo = __shfl_xor(mx, 32);
        big[t] = (mx > mo ? mx : mo) > 2048u || !fits;
        __builtin_amdgcn_sched_barrier(0);
    }

    float lv[NT][5];
    uint32_t lb[NT][4];
    const float vempty = __uint_as_float(__float_as_uint(-3.0e38f) & ~31u);  // empty slot: fresh bit clear
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int c = 0; c < 5; ++c) lv[t][c] = vempty;
#pragma unroll
        for (int c = 0; c < 4; ++c) lb[t][c] = CX_SENT;
    }
    const float fbig = 3.3e38f;  // above every value: med3(v0, x, fbig) = max(v0, x) without the canonicalising v_max pair

    // after a block's 16 insertions: block number -> id list at the fresh positions, fresh bits cleared.
    // A no-op when no fresh bit is set.
    auto settle = [&](uint32_t cur) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            uint32_t m[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(m[c]) : "v"(lv[t][c]));  // 0 / ~0 from bit 0
            lb[t][3] = phk_bfi_hw(m[0], lb[t][2], lb[t][3]);
            lb[t][2] = phk_bfi_hw(m[0], lb[t][1], lb[t][2]);
            lb[t][1] = phk_bfi_hw(m[0], lb[t][0], lb[t][1]);
            lb[t][0] = phk_bfi_hw(m[0], cur, lb[t][0]);
            lb[t][3] = phk_bfi_hw(m[1], lb[t][2], lb[t][3]);
            lb[t][2] = phk_bfi_hw(m[1], lb[t][1], lb[t][2]);
            lb[t][1] = phk_bfi_hw(m[1], cur, lb[t][1]);
            lb[t][3] = phk_bfi_hw(m[2], lb[t][2], lb[t][3]);
            lb[t][2] = phk_bfi_hw(m[2], cur, lb[t][2]);
            lb[t][3] = phk_bfi_hw(m[3], cur, lb[t][3]);
#pragma unroll
            for (int c = 0; c < 5; ++c) lv[t][c] = __uint_as_float(__float_as_uint(lv[t][c]) & ~1u);
        }
    };
    // insertion of one value (w = acc, the bias is in it since round 5: the MFMA's 17th k-step), index bits embedded, 5 x v_med3
    auto insert = [&](int t, float w, int r) {
        // A value no lane of the wave can place (w <= the best value its list has dropped, in every lane) changes nothing:
        // wave-uniform skip of the index bits and the five v_med3.  A lane's list holds the 5 best of the n values it has
        // seen, so a value enters with probability 5 / n and some lane of the wave takes one with 1 - (1 - 5/n)^64: 63 % of
        // the insertions of a 4000-column sweep are skipped (2.39 -> 2.23 ms on configs[1]).  The branch is uniform; the
        // MFMAs stay one per basic block.  (The test is on the value before its index bits: both are "the computed value"
        // within the 31 ulp the error model charges.)
        if (__builtin_amdgcn_ballot_w64(w > lv[t][4]) == 0) return;
        const float x = __uint_as_float((__float_as_uint(w) & ~31u) | (uint32_t)(2 * r + 1));
        // in place, last slot first (slot c takes med3(slot c-1, slot c, x), both still the old values): as builtins the
        // five results are temporaries that the join after the skip has to move into the list's registers -- 129 v_mov
        // per block iteration, a quarter of its vector instructions
        asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(lv[t][4]) : "v"(lv[t][3]), "v"(x));
        asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(lv[t][3]) : "v"(lv[t][2]), "v"(x));
        asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(lv[t][2]) : "v"(lv[t][1]), "v"(x));
        asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(lv[t][1]) : "v"(lv[t][0]), "v"(x));
        asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(lv[t][0]) : "v"(x), "v"(fbig));
    };

    // Software pipeline inside the wave, per segment: while block i's MFMAs fill one accumulator set (16 k-steps of the
    // counts, then the bias step), the VALU inserts block i-1's values out of the other set and settles the ids of block
    // i-2's insertions under the first fragment read.  Nothing but the workgroup barrier is left outside the MFMA stream;
    // the only data-dependent control flow of the hot loop is the wave-uniform skip inside insert().  The pipeline drains at
    // the end of a segment (3 times per sweep); the DMA stream does not.
    f32x16 accA[NT], accB[NT];
    uint32_t g = 0;  // global block number: DMA source
    uint32_t cur = 0, nxt = 2;   // ring positions of the block being read / requested

    auto block_iter = [&](uint32_t settle_id, f32x16 (&cur_acc)[NT], const f32x16 (&prev)[NT]) {
        // block g has landed (every wave waits for its own pieces -- those of block g + 1 may be outstanding -- then the
        // barrier), and every wave is done reading the buffer of block g - 1, which block g + 2's DMAs overwrite
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        __syncthreads();
        const uint8_t *buf = smem + cur * F16H_BLOCK_BYTES;
        const half8 *fr = reinterpret_cast<const half8 *>(buf + lane16);   // (lane16 is live for the DMAs: `lane` itself was spilled for this)
        half8 ahn = fr[0];
        // block i - 2's ids are settled here, under the latency of the first fragment read (38 vector instructions that need
        // nothing from LDS), not behind the first MFMAs
        __builtin_amdgcn_sched_barrier(0);
        settle(settle_id);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 17; ++s) {
            const half8 ah = ahn;
            if (s < 16) ahn = fr[(s + 1) * 64];   // (piece 16: the bias fragment)
            __builtin_amdgcn_sched_barrier(0);  // the reads stay up here, one step ahead of their MFMAs (two steps ahead: 4 spilled registers, 2.17 ms instead of 2.10)
            // (also measured and not kept: s_setprio 2 around the two MFMAs, 2.42 ms against 2.21 on the same box; the first
            // tile's insertion between the two MFMAs instead of behind them, 2.19 against 2.21: within the noise)
            if (s == 0) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) cur_acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq[t][s], z, 0, 0, 0);
            } else if (s < 16) {
#pragma unroll
                for (int t = 0; t < NT; ++t) cur_acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq[t][s], cur_acc[t], 0, 0, 0);
            } else {   // the bias: - T b~_j, LAST, so that the sixteen steps before it run on the small sums of the centred counts
#pragma unroll
                for (int t = 0; t < NT; ++t) cur_acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bqx[t], cur_acc[t], 0, 0, 0);
            }
            // block g + 2's pieces, one behind the MFMAs of every third k-step (one past the end on the last blocks: padded)
            if (s % 3 == 1 && s / 3 < NPW) dma_piece(g + 2, nxt, s / 3);
            // (D[i][j]: register r of lane (j, h') holds column row i = (r&3) + 8(r>>2) + 4h')
            if (s < 16) {
#pragma unroll
                for (int t = 0; t < NT; ++t) insert(t, prev[t][s], s);
            }
        }
        ++g;
        cur = cur == F16H_NBUF - 1 ? 0 : cur + 1;
        nxt = nxt == F16H_NBUF - 1 ? 0 : nxt + 1;
    };
    // drain: the last block's values (in `last`), ids, then the segment's lists go out
    auto finish = [&](int seg, uint32_t nb, const f32x16 (&last)[NT]) {
        settle(nb - 2);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 16; ++s) insert(t, last[t][s], s);
        settle(nb - 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // (the query number is formed here, three times per sweep, from the DMA offset register by an opaque instruction:
            // as the prologue's value it is kept -- spilled -- across the block loop)
            uint32_t jj;
            asm volatile("v_bfe_u32 %0, %1, 4, 5" : "=v"(jj) : "v"(lane16));
            const uint64_t qi = q0 + 32 * t + jj;
            if (qi < N) {
                                uint32_t ix[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t r = (__float_as_uint(lv[t][c]) >> 1) & 15u;
                    ix[c] = (lb[t][c] == CX_SENT || big[t]) ? 0xFFFFFFFFu
                                                             : lb[t][c] * 32u + (r & 3u) + 8u * (r >> 2) + 4u * (uint32_t)h;
                }
                cand_store(cand_v, cand_i, cand_u, seg, h, qi, N, lv[t][0], lv[t][1], lv[t][2], lv[t][3], ix[0], ix[1], ix[2], ix[3], big[t] ? 3.0e38f : lv[t][4]);
            }
        }
    };
#pragma unroll 1
    for (int seg = 0; seg < NSEG; ++seg) {
        const uint32_t nb = seg == 0 ? nblk_ref : seg == 1 ? nblk_pos : nblk_neg;
        if (nb == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                uint32_t jj;   // (as in finish(): not the prologue's value)
                asm volatile("v_bfe_u32 %0, %1, 4, 5" : "=v"(jj) : "v"(lane16));
                const uint64_t qi = q0 + 32 * t + jj;
                if (qi < N) {
                                        cand_store_empty(cand_v, cand_i, cand_u, seg, h, qi, N);
                }
            }
            continue;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int c = 0; c < 5; ++c) lv[t][c] = vempty;
#pragma unroll
            for (int c = 0; c < 4; ++c) lb[t][c] = CX_SENT;
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[t][r] = -3.35e38f;  // "block -1": below the empty slots, never accepted
        }
        uint32_t i = 0;
#pragma unroll 1
        for (; i + 1 < nb; i += 2) {
            block_iter(i - 2, accA, accB);
            block_iter(i - 1, accB, accA);
        }
        if (i < nb) {
            block_iter(i - 2, accA, accB);
            finish(seg, nb, accA);
        } else {
            finish(seg, nb, accB);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the trailing prefetch before the LDS is released
}

__global__ void phk_rowsum_kernel(const uint32_t *__restrict__ counts, uint64_t N, uint64_t D, uint32_t *__restrict__ out);
// the count-exact kernels centre a row by a function of its sum: callers that hold no row sums get them computed here
static int ensure_rowsum(phk_ctx *ctx, const uint32_t *d_counts, uint64_t nb, uint64_t D, const uint32_t *&d_rowsum) {
    if (d_rowsum) return PHK_OK;
    void *rs;
    PHK_TRY(phk_ws(ctx, WS_NWIN, nb * sizeof(uint32_t), &rs));
    PHK_LAUNCH(ctx, "phk_rowsum_kernel",
               phk_rowsum_kernel<<<dim3((unsigned)phk_div_up(nb, 4)), dim3(256), 0, ctx->stream>>>(d_counts, nb, D, (uint32_t *)rs));
    d_rowsum = (const uint32_t *)rs;
    return PHK_OK;
}

int phk_launch_proposal_f16h(phk_ctx *ctx, const phk_model *m, const uint32_t *d_counts, const uint32_t *d_rowsum,
                             uint64_t nb, uint32_t nref, uint32_t npos, uint32_t nneg, float *cv, uint32_t *ci, float *cu) {
    PHK_TRY(ensure_rowsum(ctx, d_counts, nb, m->D, d_rowsum));
    const size_t lds = F16H_NBUF * F16H_BLOCK_BYTES;
    const uint4 *af = (const uint4 *)m->d_Af16h + (uint64_t)(nref ? 0 : m->n_rblk_ref) * (F16H_BLOCK_BYTES / 16);
    // Two 4-wave workgroups per CU (default, "24") or one 8-wave workgroup ("28"): with two workgroups the two waves of a
    // SIMD share no barrier, and the one that lost the issue arbitration does not hold the other up at every block
    // (2.44 vs 2.62 ms on configs[1]; one wave per SIMD with 3 or 4 tiles, tried through AGPRs, ran 2.9 / 3.3 ms, and
    // 2-wave workgroups quadruple the L2 -> LDS record traffic: 13.6 ms).
    const char *e = ctx->knobs.cx_cfg;
    const float tscale = std::ldexp(1.0f, -m->bias_e);
    if (e[0] == '2' && e[1] == '8') {
        PHK_LAUNCH(ctx, "phk_knn_f16h_kernel",
                   (phk_knn_f16h_kernel<2, 8><<<dim3((unsigned)phk_div_up(nb, 32 * 8 * 2)), dim3(64 * 8), lds, ctx->stream>>>(
                       d_counts, d_rowsum, nb, af, nref, npos, nneg, cv, ci, cu, tscale)));
    } else {
        PHK_LAUNCH(ctx, "phk_knn_f16h_kernel",
                   (phk_knn_f16h_kernel<2, 4><<<dim3((unsigned)phk_div_up(nb, 32 * 4 * 2)), dim3(64 * 4), lds, ctx->stream>>>(
                       d_counts, d_rowsum, nb, af, nref, npos, nneg, cv, ci, cu, tscale)));
    }
    return PHK_OK;
}

// ====================================================================================
// General D = 256 * nchunk (k = 5: 1024, k = 6: 4096): the query operand no longer fits in
// registers, so (i) queries are split once into fragment order in a workspace (phk_split_queries),
// (ii) a wave keeps CT = 4 column-block accumulators and sweeps the K chunks, re-loading its 32
// queries' chunk fragments (32 KiB, coalesced) once per (tile, chunk) = once per 192 MFMAs, while
// the column-chunk records stream through the same LDS double buffer as above.
// ====================================================================================
__global__ __launch_bounds__(256) void phk_rowsum_kernel(const uint32_t *__restrict__ counts, uint64_t N, uint64_t D, uint32_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= N) return;
    uint32_t s = 0;
    for (uint64_t d = lane; d < D; d += 64) s += counts[r * D + d];
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh);
    if (lane == 0) out[r] = s;
}

// one wave per (query block of 32, chunk): Bq[(qb * nchunk + c) * 32 + piece][lane] (uint4)
template <int SRC>
__global__ __launch_bounds__(256) void phk_split_queries_kernel(const void *__restrict__ src,
                                                                const uint32_t *__restrict__ rowsum, uint64_t N,
                                                                uint64_t D, const float *__restrict__ mu32,
                                                                const double *__restrict__ mu64,
                                                                uint4 *__restrict__ Bq, uint32_t *__restrict__ big) {
    // SRC 0: counts -> normalised, centred, split (hi, lo);  SRC 1: float64 rows -> centred, split;
    // SRC 2 (count-exact): counts -> fp16 as they are (hi slots only; lo slots unused), rows holding a count
    // above 2048 are flagged in `big`
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const uint64_t nchunk = D / 256;
    const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nqb = (N + 31) / 32;
    if (w >= nqb * nchunk) return;
    const uint64_t qb = w / nchunk, c = w % nchunk;
    const uint64_t qrow = (qb * 32 + j < N) ? qb * 32 + j : N - 1;
    const uint64_t d0 = 256 * c + 128 * h;
    uint4 *out = Bq + (w * 32) * 64 + lane;
    float inv = 0.f;
    if (SRC == 0) inv = (float)(1.0 / (double)rowsum[qrow]) * F16_SCALE;
    if (SRC == 2) {   // the counts minus the row's centre (phk_row_center), exact in fp16 up to 2048 in magnitude
        const int cen = (int)phk_row_center(rowsum[qrow], (uint32_t)D);
        uint32_t mx = 0;
        // a lane owns 128 consecutive counts of its query's row (512 B): fetched as whole 64-byte pieces, four loads in
        // flight per piece (32-byte pieces, two loads at a time, made the memory system fetch every sector twice)
        const uint4 *row = reinterpret_cast<const uint4 *>(static_cast<const uint32_t *>(src) + qrow * D + d0);
#pragma unroll 2
        for (int s2 = 0; s2 < 8; ++s2) {
            uint4 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = row[4 * s2 + e];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const uint32_t c[8] = {v[2 * half].x, v[2 * half].y, v[2 * half].z, v[2 * half].w,
                                       v[2 * half + 1].x, v[2 * half + 1].y, v[2 * half + 1].z, v[2 * half + 1].w};
                half8 hi;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int d = (int)c[e] - cen;
                    const uint32_t ad = (uint32_t)(d < 0 ? -d : d);
                    mx = max(mx, max(ad, c[e] >> 31 ? 0xFFFFFFFFu : 0u));
                    hi[e] = (_Float16)(float)(d < -2048 ? -2048 : (d > 2048 ? 2048 : d));
                }
                out[(2 * (2 * s2 + half)) * 64] = *reinterpret_cast<uint4 *>(&hi);
            }
        }
        if (mx > 2048u) atomicOr(big + qrow, 1u);
        return;
    }
    for (int s = 0; s < 16; ++s) {
        float x[8];
        if (SRC == 0) {
            const uint4 *row = reinterpret_cast<const uint4 *>(static_cast<const uint32_t *>(src) + qrow * D + d0 + 8 * s);
            const float4 *mp = reinterpret_cast<const float4 *>(mu32 + d0 + 8 * s);
            const uint4 c0 = row[0], c1 = row[1];
            const float4 m0 = mp[0], m1 = mp[1];
            x[0] = fmaf((float)c0.x, inv, -m0.x * F16_SCALE);
            x[1] = fmaf((float)c0.y, inv, -m0.y * F16_SCALE);
            x[2] = fmaf((float)c0.z, inv, -m0.z * F16_SCALE);
            x[3] = fmaf((float)c0.w, inv, -m0.w * F16_SCALE);
            x[4] = fmaf((float)c1.x, inv, -m1.x * F16_SCALE);
            x[5] = fmaf((float)c1.y, inv, -m1.y * F16_SCALE);
            x[6] = fmaf((float)c1.z, inv, -m1.z * F16_SCALE);
            x[7] = fmaf((float)c1.w, inv, -m1.w * F16_SCALE);
        } else {
            const double *row = static_cast<const double *>(src) + qrow * D + d0 + 8 * s;
            const double *mp = mu64 + d0 + 8 * s;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                x[t] = (float)((row[t] - mp[t]) * (double)F16_SCALE);
                asm("" : "+v"(x[t]));   // (opaque: no fold into a software double -> half conversion; see phk_knn_f16_kernel)
            }
        }
        half8 hi, lo;
        f16_split8(x, hi, lo);
        out[(2 * s) * 64] = *reinterpret_cast<uint4 *>(&hi);
        out[(2 * s + 1) * 64] = *reinterpret_cast<uint4 *>(&lo);
    }
}

// CX (count-exact): Bq holds the fp16 counts (hi slots), 2 MFMAs per k-step, value = acc - T * bias (rowsum,
// beta_all), rows flagged in `big` get empty lists (-> exact brute-force queue)
// GEN_CT = column blocks per tile (accumulators per wave): a query chunk fetched from memory serves GEN_CT x 16 k-steps;
// the count-exact flavour has no lo query fragments and spends the registers on 8 accumulators instead of 4
// HI (with CX): high parts of the columns only -- ONE MFMA per k-step, only the 16 hi pieces of a (block, chunk) item are
// streamed, the bias is the high-part one (beta_all = betah); the decision stage (phk_rerank_h_kernel) adds the low
// product to the window's members
// GEN_NW = waves per workgroup: the (block, chunk) items streamed into LDS are shared by all of them, so 8 waves (one
// workgroup per CU) halve the L2 -> LDS traffic per MFMA of two 4-wave workgroups that each stream their own copy --
// and that stream is what bounds the kernel (32 KiB per 32 MFMAs per wave: ~64 GB/s per CU at 4 waves)
template <bool CX, int GEN_CT, bool HI = false, int GEN_NW = 8>
__global__ __launch_bounds__(64 * GEN_NW, GEN_NW == 8 ? 1 : 2) void phk_knn_f16_general_kernel(const uint4 *__restrict__ Bq, uint64_t N,
                                                                     uint32_t nchunk,
                                                                     const uint4 *__restrict__ Af,
                                                                     uint64_t rec_u4,  // uint4 per block record
                                                                     const float *__restrict__ cn_all,
                                                                     const float *__restrict__ beta_all,
                                                                     const uint32_t *__restrict__ rowsum,
                                                                     const uint32_t *__restrict__ big,
                                                                     uint32_t blk0,    // first block swept
                                                                     uint32_t nblk_ref, uint32_t nblk_pos,
                                                                     uint32_t nblk_neg,
                                                                     float *__restrict__ cand_v,
                                                                     uint32_t *__restrict__ cand_i,
                                                                     float *__restrict__ cand_u,
                                                                     float *__restrict__ cand_a,
                                                                     uint32_t ngroups,   // column groups (2-D launch), or 1
                                                                     uint64_t set_bytes, uint64_t ca_set) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 2 x 32 KiB
    // (a third buffer with the DMA running two items ahead and counted vmcnt waits was measured: 106 instead of 96 ms at
    // config 4 -- the deeper queue does not pay)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const uint64_t nqb = (N + 31) / 32;
    // 2-D launch (ngroups > 1; D >= 2048): a workgroup sweeps ONE of `ngroups` groups of train blocks for its 256 queries
    // and writes list set g (set_bytes apart; the last group also sweeps the centroid blocks), phk_merge_list_sets_kernel
    // folds the sets.  Workgroups are numbered so that the `ngroups` workgroups of a query block have equal
    // blockIdx.x % 8 and neighbouring numbers: under the dispatcher's round-robin they share an XCD -- and its L2 serves
    // the query fragments that every one of them re-reads once per 8 column blocks (a speed matter only; nothing
    // depends on the placement).  With one group per workgroup those fragments came from HBM: 283 GB per configs[4] step.
    // The number of groups is a balance (score_model.h, PHK_GEN_GROUPS): every group more also splits the column stream
    // that the workgroups of an XCD otherwise pull through its L2 in step.
    uint64_t wg = blockIdx.x;
    uint32_t col0 = 0;
    if (ngroups > 1) {
        const uint32_t li = blockIdx.x >> 3, g = li % ngroups;
        wg = (uint64_t)(li / ngroups) * 8 + (blockIdx.x & 7u);
        const uint32_t b0 = (uint32_t)((uint64_t)nblk_ref * g / ngroups), b1 = (uint32_t)((uint64_t)nblk_ref * (g + 1) / ngroups);
        blk0 += b0;
        col0 = 32u * b0;
        nblk_ref = b1 - b0;
        if (g + 1 != ngroups) nblk_pos = nblk_neg = 0;
        cand_v = reinterpret_cast<float *>(reinterpret_cast<char *>(cand_v) + (uint64_t)g * set_bytes);
        cand_i = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(cand_i) + (uint64_t)g * set_bytes);
        cand_u = reinterpret_cast<float *>(reinterpret_cast<char *>(cand_u) + (uint64_t)g * set_bytes);
        if (cand_a) cand_a += (uint64_t)g * ca_set;
    }
    if (wg * GEN_NW >= nqb) return;   // padding workgroup of the 2-D numbering (uniform: before any barrier)
    const uint64_t qb = wg * GEN_NW + wave;
    const uint64_t q0 = qb * 32;
    const uint64_t qbc = qb < nqb ? qb : nqb - 1;  // padding waves re-read the last block; nothing is written
    const uint32_t total = nblk_ref + nblk_pos + nblk_neg;
    const uint32_t seg_end0 = nblk_ref, seg_end1 = nblk_ref + nblk_pos;
    const uint32_t ntile = (total + GEN_CT - 1) / GEN_CT;
    const uint64_t nitem = (uint64_t)ntile * nchunk * GEN_CT;

    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem;
    // item it = ((tile * nchunk) + chunk) * CT + cb  ->  32 pieces of (block, chunk)
    // Source of the pieces of item (tile, chunk, cb); this wave's are p = wave, wave + GEN_NW, ..  (Coordinates, not the
    // item number: the 64-bit divisions that recover them from it cost ~400 cycles per item.)
    constexpr int NPIECE = (HI ? 16 : 32) / GEN_NW;   // pieces per wave and item
    auto item_src = [&](uint32_t tile, uint32_t c, uint32_t cb) {
        if (tile >= ntile) { tile = ntile - 1; c = nchunk - 1; cb = GEN_CT - 1; }   // one past the end: harmless re-read
        uint32_t blk = tile * GEN_CT + cb;
        if (blk >= total) blk = total - 1;  // partial last tile: harmless re-read, result discarded
        return Af + (uint64_t)(blk0 + blk) * rec_u4 + (uint64_t)c * 32 * 64 + lane;
    };
    auto dma_piece = [&](const uint4 *g, int buf, int i) {   // i-th piece of this wave
        const int p = wave + GEN_NW * i;   // HI: source pieces 0, 2, 4, .. (hi) land as pieces 0, 1, 2, ..
        const uint4 *gp = g + (HI ? 2 * p : p) * 64;
        const uint32_t lp = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)buf * 32768u + (uint32_t)p * 1024u);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(lp) : "memory");
    };
    if (nitem) {
        const uint4 *g0 = item_src(0, 0, 0);
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) dma_piece(g0, 0, i);
    }

    float lv[CAND];
    uint32_t li[CAND];
    float ldrop = -3.0e38f;
#pragma unroll
    for (int c = 0; c < CAND; ++c) {
        lv[c] = -3.0e38f;
        li[c] = 0xFFFFFFFFu;
    }
    float negT = 0.f;
    bool isbig = false;
    // the largest |accumulator| this lane meets at the (block, chunk) item boundaries: what the chain's rounding and
    // alignment errors scale with (ErrBound, D > 256); the decision stage reads it from cand_a
    float amax = 0.f;
    if (CX) {
        const uint64_t qr = (qb < nqb && q0 + j < N) ? q0 + j : N - 1;
        negT = -(float)rowsum[qr];
        isbig = big[qr] != 0;
    }
    int seg = 0;
    uint32_t seg_first = 0;
    uint64_t it = 0;
    // leading segments without columns (method 'kmeans': no train rows are swept): their lists are empty, and the
    // flush rule below -- "block b closes segment seg when b + 1 is its end" -- would never fire for them
    while (seg < NSEG && (seg == 0 ? seg_end0 : seg == 1 ? seg_end1 : total) == 0) {
        if (qb < nqb && q0 + j < N) cand_store_empty(cand_v, cand_i, cand_u, seg, h, q0 + j, N);
        ++seg;
    }
    for (uint32_t t = 0; t < ntile; ++t) {
        f32x16 acc[GEN_CT];
#pragma unroll
        for (int cb = 0; cb < GEN_CT; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
        for (uint32_t c = 0; c < nchunk; ++c) {
            // this wave's query fragments of chunk c (16 x hi + 16 x lo, 1 KiB coalesced loads)
            half8 bh[16], bl[16];
            const uint4 *bq = Bq + ((qbc * nchunk + c) * 32) * 64 + lane;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const uint4 uh = bq[(2 * s) * 64];
                bh[s] = *reinterpret_cast<const half8 *>(&uh);
                if (!CX) {
                    const uint4 ul = bq[(2 * s + 1) * 64];
                    bl[s] = *reinterpret_cast<const half8 *>(&ul);
                }
            }
#pragma unroll
            for (int cb = 0; cb < GEN_CT; ++cb, ++it) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                // The next item's pieces are issued BETWEEN this item's first k-steps, not in a burst after the barrier:
                // there the 32 pieces of the 8 waves queued at the CU's DMA path and every wave spent ~950 cycles per
                // item (of 4200, by in-kernel stamps) doing nothing else.
                const uint4 *gnext = cb + 1 < GEN_CT ? item_src(t, c, cb + 1) : (c + 1 < nchunk ? item_src(t, c + 1, 0) : item_src(t + 1, 0, 0));
                const int nbuf = (int)((it + 1) & 1);
                const half8 *fr = reinterpret_cast<const half8 *>(smem + (it & 1) * 32768u) + lane;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    if (s < NPIECE) dma_piece(gnext, nbuf, s);
                    if (HI) {
                        const half8 ah = fr[s * 64];
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[s], acc[cb], 0, 0, 0);
                    } else {
                        const half8 ah = fr[(2 * s) * 64];
                        const half8 al = fr[(2 * s + 1) * 64];
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[s], acc[cb], 0, 0, 0);
                        if (!CX) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[s], acc[cb], 0, 0, 0);
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[s], acc[cb], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) amax = fmaxf(amax, fmaxf(fabsf(acc[cb][r]), fabsf(acc[cb][r + 1])));
            }
        }
        // epilogue of the tile: norm terms from global memory, insertion (skipped by the whole wave when no lane can place
        // the value), segment flushes
#pragma unroll
        for (int cb = 0; cb < GEN_CT; ++cb) {
            const uint32_t blk = t * GEN_CT + cb;
            if (blk < total) {
                const float4 *cn = reinterpret_cast<const float4 *>((CX ? beta_all : cn_all) + (uint64_t)(blk0 + blk) * 32) + h;
                const uint32_t cbase = 32u * (blk - seg_first) + 4u * (uint32_t)h;
                const float mul = CX ? negT : 1.0f;   // CX: acc - T * bias;  else acc + norm term
#pragma unroll
                for (int m4 = 0; m4 < 4; ++m4) {
                    const float4 c4 = cn[2 * m4];
                    list_insert_needed(lv, li, ldrop, fmaf(mul, c4.x, acc[cb][4 * m4 + 0]), cbase + 8u * m4 + 0u);
                    list_insert_needed(lv, li, ldrop, fmaf(mul, c4.y, acc[cb][4 * m4 + 1]), cbase + 8u * m4 + 1u);
                    list_insert_needed(lv, li, ldrop, fmaf(mul, c4.z, acc[cb][4 * m4 + 2]), cbase + 8u * m4 + 2u);
                    list_insert_needed(lv, li, ldrop, fmaf(mul, c4.w, acc[cb][4 * m4 + 3]), cbase + 8u * m4 + 3u);
                }
                while (seg < NSEG && blk + 1 == (seg == 0 ? seg_end0 : seg == 1 ? seg_end1 : total)) {
                    if (qb < nqb && q0 + j < N) {
                        if (CX && isbig)
                            cand_store(cand_v, cand_i, cand_u, seg, h, q0 + j, N, -3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f,
                                       0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 3.0e38f);
                        else {
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
                    ++seg;
                    seg_first = blk + 1;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (; seg < NSEG; ++seg) {
        if (qb < nqb && q0 + j < N) {
            cand_store_empty(cand_v, cand_i, cand_u, seg, h, q0 + j, N);
        }
    }
    if (cand_a && qb < nqb && q0 + j < N) cand_a[(uint64_t)h * N + q0 + j] = amax;
}

// proposal pass for D = 256 * nchunk > 256: row sums (if needed) -> split queries -> sweep
int phk_launch_proposal_f16_general(phk_ctx *ctx, const phk_model *m, const void *src, bool src_counts, bool count_exact,
                                    const uint32_t *d_rowsum, uint64_t nb, uint32_t nref, uint32_t npos, uint32_t nneg,
                                    float *cv, uint32_t *ci, float *cu, float *ca, bool hi_only, uint32_t groups,
                                    uint64_t set_bytes) {
    const uint64_t D = m->D, nchunk = D / 256;
    const uint64_t nqb = phk_div_up(nb, 32);
    void *bq, *rs = nullptr;
    PHK_TRY(phk_ws(ctx, WS_Q64, nqb * nchunk * 32 * 1024, &bq));
    if (src_counts && !d_rowsum) {
        PHK_TRY(phk_ws(ctx, WS_NWIN, nb * sizeof(uint32_t), &rs));
        PHK_LAUNCH(ctx, "phk_rowsum_kernel",
                   phk_rowsum_kernel<<<dim3((unsigned)phk_div_up(nb, 4)), dim3(256), 0, ctx->stream>>>(
                       (const uint32_t *)src, nb, D, (uint32_t *)rs));
        d_rowsum = (const uint32_t *)rs;
    }
    const unsigned sblocks = (unsigned)phk_div_up(nqb * nchunk, 4);
    uint32_t *d_big = nullptr;
    if (src_counts && count_exact) {
        void *bg;
        PHK_TRY(phk_ws(ctx, WS_LONG, nb * sizeof(uint32_t), &bg));
        d_big = (uint32_t *)bg;
        PHK_HIP(hipMemsetAsync(d_big, 0, nb * sizeof(uint32_t), ctx->stream));
        PHK_LAUNCH(ctx, "phk_split_queries_kernel",
                   phk_split_queries_kernel<2><<<dim3(sblocks), dim3(256), 0, ctx->stream>>>(
                       src, d_rowsum, nb, D, m->d_mu32, m->d_mu64, (uint4 *)bq, d_big));
    } else if (src_counts) {
        PHK_LAUNCH(ctx, "phk_split_queries_kernel",
                   phk_split_queries_kernel<0><<<dim3(sblocks), dim3(256), 0, ctx->stream>>>(
                       src, d_rowsum, nb, D, m->d_mu32, m->d_mu64, (uint4 *)bq, nullptr));
    } else {
        PHK_LAUNCH(ctx, "phk_split_queries_kernel",
                   phk_split_queries_kernel<1><<<dim3(sblocks), dim3(256), 0, ctx->stream>>>(
                       src, nullptr, nb, D, m->d_mu32, m->d_mu64, (uint4 *)bq, nullptr));
    }
    const uint64_t rec_u4 = (nchunk * 32 + 1) * 64;
    const uint32_t blk0 = nref ? 0 : m->n_rblk_ref;
    // 2-D launch for D >= 2048 (see the kernel): `groups` column groups per query block, list sets set_bytes apart
    const uint32_t ng = (groups > 1 && nref >= 8u * groups && nqb >= 64) ? groups : 1u;
    const uint64_t nqg = phk_div_up(nqb, 8);
    const unsigned gblocks = ng > 1 ? (unsigned)(phk_div_up(nqg, 8) * 8 * ng) : (unsigned)nqg;
    const uint64_t ca_set = 2 * nb;
    if (d_big && hi_only) {
        PHK_LAUNCH(ctx, "phk_knn_f16_general_kernel",
                   (phk_knn_f16_general_kernel<true, 8, true><<<dim3(gblocks), dim3(512), 65536, ctx->stream>>>(
                       (const uint4 *)bq, nb, (uint32_t)nchunk, (const uint4 *)m->d_Af16, rec_u4, m->d_cn16, m->d_betah16, d_rowsum,
                       d_big, blk0, nref, npos, nneg, cv, ci, cu, ca, ng, set_bytes, ca_set)));
    } else if (d_big) {
        PHK_LAUNCH(ctx, "phk_knn_f16_general_kernel",
                   (phk_knn_f16_general_kernel<true, 8><<<dim3(gblocks), dim3(512), 65536, ctx->stream>>>(
                       (const uint4 *)bq, nb, (uint32_t)nchunk, (const uint4 *)m->d_Af16, rec_u4, m->d_cn16, m->d_beta16, d_rowsum,
                       d_big, blk0, nref, npos, nneg, cv, ci, cu, ca, ng, set_bytes, ca_set)));
    } else {
        PHK_LAUNCH(ctx, "phk_knn_f16_general_kernel",
                   (phk_knn_f16_general_kernel<false, 4><<<dim3(gblocks), dim3(512), 65536, ctx->stream>>>(
                       (const uint4 *)bq, nb, (uint32_t)nchunk, (const uint4 *)m->d_Af16, rec_u4, m->d_cn16, m->d_beta16, nullptr,
                       nullptr, blk0, nref, npos, nneg, cv, ci, cu, ca, ng, set_bytes, ca_set)));
    }
    if (ng > 1) {
        PHK_LAUNCH(ctx, "phk_merge_list_sets_kernel",
                   phk_merge_list_sets_kernel<<<dim3((unsigned)phk_div_up(2 * nb, 256)), dim3(256), 0, ctx->stream>>>(
                       cv, ci, cu, nb, set_bytes, (int)ng, nullptr, ca, ca_set));
    }
    return PHK_OK;
}

// ------------------------------------------------------------------------------------
// Diagnostic: the instruction the proposal kernels are built on, on caller-supplied tiles.  The certification charges
// every v_mfma_f32_32x32x16_f16 with u (11 A + 18 p) (score_lists.h) -- a measured model of an
// instruction whose internal accumulation the ISA does not state; tests/test_gpu_score.py drives this entry with
// structured worst cases (cut-maximising and cancellation-heavy tiles, fp16 subnormals, counts at the 2048 limit) and
// tests/mfma_fuzz_worker.py with 650 000 random instructions, and both assert that bound, step by step over chains as
// long as the kernels' (16 k-steps at k = 4, 256 at D = 4096).
// One wave per tile: acc = C; for s < steps: acc = mfma(A[s], B[s], acc), stored after every step.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void phk_mfma_f16_probe_kernel(const _Float16 *__restrict__ A, const _Float16 *__restrict__ B,
                                                                const float *__restrict__ C, float *__restrict__ Dout,
                                                                uint32_t steps) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const uint64_t tile = blockIdx.x;
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = C[tile * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i];
    for (uint32_t s = 0; s < steps; ++s) {
        const _Float16 *a = A + (tile * steps + s) * 512, *b = B + (tile * steps + s) * 512;
        half8 av, bv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {   // A[32][16] row-major, B[16][32] row-major (lane: row / column i, k = 8 h + j)
            av[j] = a[i * 16 + 8 * h + j];
            bv[j] = b[(8 * h + j) * 32 + i];
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) Dout[(tile * steps + s) * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = c[r];
    }
}

extern "C" int phk_mfma_f16_probe(phk_ctx *ctx, const uint16_t *A, const uint16_t *B, const float *C, uint64_t n_tiles,
                                  uint32_t steps, float *D) {
    PHK_ENTER(ctx, "phk_mfma_f16_probe");
    PHK_REQUIRE(A && B && C && D && n_tiles > 0 && steps > 0, "phk_mfma_f16_probe: NULL pointer / empty problem");
    const uint64_t ab = n_tiles * steps * 512 * 2, cb = n_tiles * 1024 * 4, db = n_tiles * steps * 1024 * 4;
    void *dA, *dB, *dC, *dD;
    PHK_TRY(phk_ws(ctx, WS_WIDE, ab, &dA));
    PHK_TRY(phk_ws(ctx, WS_Q64, ab, &dB));
    PHK_TRY(phk_ws(ctx, WS_COUNTS, cb, &dC));
    PHK_TRY(phk_ws(ctx, WS_OUT, db, &dD));
    PHK_HIP(hipMemcpyAsync(dA, A, ab, hipMemcpyHostToDevice, ctx->stream));
    PHK_HIP(hipMemcpyAsync(dB, B, ab, hipMemcpyHostToDevice, ctx->stream));
    PHK_HIP(hipMemcpyAsync(dC, C, cb, hipMemcpyHostToDevice, ctx->stream));
    PHK_LAUNCH(ctx, "phk_mfma_f16_probe_kernel",
               phk_mfma_f16_probe_kernel<<<dim3((unsigned)n_tiles), dim3(64), 0, ctx->stream>>>(
                   (const _Float16 *)dA, (const _Float16 *)dB, (const float *)dC, (float *)dD, steps));
    PHK_HIP(hipMemcpyAsync(D, dD, db, hipMemcpyDeviceToHost, ctx->stream));
    PHK_HIP(hipStreamSynchronize(ctx->stream));
    return PHK_OK;
}

// per-device kernel attributes (dynamic LDS above the 64 KiB default), called from phk_create
int phk_score_f16_init_device(phk_ctx *ctx) {
    (void)ctx;
    const int lds = 2 * F16_BLOCK_BYTES;
    PHK_HIP(hipFuncSetAttribute((const void *)phk_knn_f16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    PHK_HIP(hipFuncSetAttribute((const void *)phk_knn_f16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    PHK_HIP(hipFuncSetAttribute((const void *)phk_knn_f16h_kernel<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, F16H_NBUF * F16H_BLOCK_BYTES));
    PHK_HIP(hipFuncSetAttribute((const void *)phk_knn_f16h_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, F16H_NBUF * F16H_BLOCK_BYTES));
    PHK_HIP(hipFuncSetAttribute((const void *)phk_knn_f16_general_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    PHK_HIP(hipFuncSetAttribute((const void *)phk_knn_f16_general_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    PHK_HIP(hipFuncSetAttribute((const void *)phk_knn_f16_general_kernel<true, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    return PHK_OK;
}
