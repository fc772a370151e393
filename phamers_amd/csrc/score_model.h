// score_model.h -- the device-resident scoring model (train rows, labels, centroids and the
// MFMA path's derived operands).
#pragma once
#include "phk_common.h"

struct phk_model {
    uint64_t D = 0, M = 0, n_pos = 0, n_neg = 0, n_cpos = 0, n_cneg = 0;
    int kn = 3;
    // exact path
    double *d_R64 = nullptr;      // [M][D] train = vstack(pos, neg)  (scripts/phamer.py:186)
    uint8_t *d_labels = nullptr;  // [M] 1 = positive row           (scripts/phamer.py:187)
    double *d_C64 = nullptr;      // [n_cpos + n_cneg][D] centroids
    // MFMA path (score_mfma.hip); null when the shape is outside it
    bool fast = false;
    double *d_colnorm = nullptr;  // |r'| per real column (train rows, pos centroids, neg centroids)
    void *d_Af16 = nullptr;       // split-f16 fragment-ordered block records (score_f16.hip)
    float *d_cn16 = nullptr;      // split-f16 norm terms per column slot (general-D kernel)
    float *d_beta16 = nullptr;    // count-exact bias terms S (mu.r~' + |r~'|^2/2) per column slot (general-D kernel)
    // high-parts-only proposal (k = 4, phk_knn_f16h_kernel): 17-piece block records (16 hi fragments + the bias terms
    // S mu.hi + S |r~'|^2/2 per unit of row sum), the low parts row-major for the decision stage's refinement, and
    // lam_tab[s][i] = max |lo_j| / S over segment s's columns with |r'_j| <= lam_r0[s] + i * lam_step[s] (error of a high-parts-only value)
    float *d_betah16 = nullptr;   // high-part bias terms per column slot (general-D kernel; mask restore target)
    void *d_Af16h = nullptr;
    // (round 5) the bias enters the k = 4 sweep as a 17th k-step of the MFMA: piece 16 of a block record holds, per column,
    // the three float16 pieces of  -bias x 2^bias_e  (on the 2^-14 grid) in fragment order; the kernel multiplies them by
    // the three pieces of  T x 2^-bias_e.  bias_e: the largest exponent that keeps every real column's bias below 2^15.
    int bias_e = 0;
    double bias_max = 0.0;        // max |bias| over the real columns (what the 17th step's products are bounded by)
    std::vector<float> h_betah;   // host copy of the high-part bias terms, per column slot (piece rebuilds)
    std::vector<uint8_t> h_rech;  // host copy of the 17-piece records
    _Float16 *d_lo16 = nullptr;   // [M + n_cpos + n_cneg][D]
    double lam_tab[3][65] = {{0}};     // per segment (train rows, positive centroids, negative centroids)
    double lam_r0[3] = {0, 0, 0}, lam_step[3] = {1, 1, 1};
    // int8 proposal for count rows at D >= 512 (score_i8.hip): block records of 24-bit fixed-point columns in three int8
    // parts + (quantum, bias) per column; kappa8 = max_j |r'_j - r~'_j| / |r'_j| of that quantisation, hsum8 as hsum_*
    void *d_A8 = nullptr;
    float *d_T8 = nullptr;        // [blocks + padding][64]: quanta and bias terms
    float *d_T8h = nullptr;       // the same with the quanta times 256 (exact): what the two-part sweep multiplies 256 S_H + S_M by
    uint64_t rec8_bytes = 0;
    double kappa8 = 0.0, hsum8 = 0.0;
    // two-part sweep (the default): records of the H and M parts only; the L digits row-major for the decision stage,
    // [M + n_cpos + n_cneg][D]; lam8[s] = max_j g_j |L_j|_2 over segment s's columns (what a two-part value can lack)
    void *d_A8h = nullptr;
    int8_t *d_L8 = nullptr;
    uint64_t rec8h_bytes = 0;
    double lam8[3] = {0, 0, 0};
    float *d_mu32 = nullptr;      // centring vector, fp32
    double *d_mu64 = nullptr;     // centring vector, fp64
    // cross-validation service (phk_model_set_centroids / phk_model_set_column_mask): the train segment's column terms as
    // built (a masked column's terms are overwritten with the padding values and restored from here), the mask itself
    // for the float64 kernels, the host copy of the centring vector and the train segment's largest column norm
    float *d_term_orig = nullptr;     // [3][n_rblk_ref * 32]: norm terms, bias terms, high-part bias terms
    uint8_t *d_col_mask = nullptr;    // [M], 1 = excluded; null until a mask is set
    bool has_mask = false;
    bool bf_stale = false;            // the fp32 / int8 MFMA operands no longer match: centroids replaced, or a column mask set
    bool cen_replaced = false;        // phk_model_set_centroids was called (stays: those operands are not rebuilt); a mask can be cleared
    std::vector<double> h_mu;
    double max_colnorm_train = 0.0;
    uint32_t n_rblk_ref = 0, n_rblk_pos = 0, n_rblk_neg = 0;  // 32-column blocks per segment
    double rho_inf = 1.0;         // max_j |r~'_j|_inf / |r'_j| over the real columns: |x|_inf |y_j|_inf <= I rho_inf R in ErrBound
    double rho_train = 0.0, rho_cen = 0.0;
    double hsum_train = 0.0, hsum_cen = 0.0;   // max_j |sum_i r~'_ji| over the train rows / the centroids (see ErrBound: habs)
    double max_colnorm = 0.0;     // max ||r'|| over real columns (error bound)
    double mu_norm = 0.0;         // ||mu||
    double mu_tilde_norm = 0.0;   // ||mu - 1/D||: what the count-exact bias terms multiply the column by
};

// second-chance pass of phk_score_fast: below this many queued rows the float64 brute force is the cheaper last resort
#define PHK_SECOND_MIN 24
#define PHK_SECOND_SPLITS 8   // column parts of a second-chance sweep (its few thousand queries alone fill ~1/6 of the CUs)

// exact float64 batch scorer (score.hip)
int phk_score_exact_batch(phk_ctx *ctx, const phk_model *m, const double *d_Q, uint64_t nq, int method,
                          double *d_knn, double *d_cen, uint32_t *d_status);

// MFMA path hooks (score_mfma.hip)
int phk_model_build_fast(phk_ctx *ctx, phk_model *m, const double *pos, const double *neg,
                         const double *cpos, const double *cneg);
void phk_model_free_fast(phk_model *m);
static inline bool phk_model_has_fast(const phk_model *m) { return m->fast; }
// split-f16 proposal (score_f16.hip)
int phk_model_build_f16(phk_model *m, const double *pos, const double *neg, const double *cpos,
                        const double *cneg, const double *mu, const double *colnorm);
int phk_model_update_centroids_f16(phk_model *m, const double *cpos, const double *cneg, const double *colnorm_c);
int phk_model_apply_mask_f16(phk_ctx *ctx, phk_model *m);
int phk_launch_proposal_f16(phk_ctx *ctx, const phk_model *m, const void *src, bool src_counts,
                            const uint32_t *d_rowsum, uint64_t nb, uint32_t nref, uint32_t npos, uint32_t nneg,
                            float *cv, uint32_t *ci, float *cu, const uint32_t *qmap = nullptr,
                            const uint32_t *qcount = nullptr, int splits = 1, uint64_t set_bytes = 0);
int phk_launch_proposal_f16h(phk_ctx *ctx, const phk_model *m, const uint32_t *d_counts, const uint32_t *d_rowsum,
                             uint64_t nb, uint32_t nref, uint32_t npos, uint32_t nneg, float *cv, uint32_t *ci, float *cu);
int phk_launch_proposal_f16_general(phk_ctx *ctx, const phk_model *m, const void *src, bool src_counts, bool count_exact,
                                    const uint32_t *d_rowsum, uint64_t nb, uint32_t nref, uint32_t npos, uint32_t nneg,
                                    float *cv, uint32_t *ci, float *cu, float *ca, bool hi_only = false, uint32_t groups = 1,
                                    uint64_t set_bytes = 0);
// int8 proposal (score_i8.hip): count rows, D >= 512, models without a column mask / replaced centroids
int phk_model_build_i8(phk_model *m, const double *pos, const double *neg, const double *cpos, const double *cneg,
                       const double *mu, const double *colnorm);
int phk_launch_proposal_i8_general(phk_ctx *ctx, const phk_model *m, const uint32_t *d_counts, const uint32_t *d_rowsum, uint64_t nb,
                                   uint32_t nref, uint32_t npos, uint32_t nneg, float *cv, uint32_t *ci, float *cu,
                                   uint32_t groups, uint64_t set_bytes, bool two_parts);
int phk_score_i8_init_device(phk_ctx *ctx);
// Column groups of the general-D sweep's 2-D launch (D >= 2048).  Measured on configs[4] (kernel ms): 1 group 92.9, 2: 89.8,
// 4: 86.4, 8: 95.1 (and 3 / 5 / 6: no better) -- more groups share a query block's fragments through one XCD's L2, but
// split the column stream that ALL workgroups of an XCD otherwise pull through it in step.  D = 1024 (configs[2]): 184 /
// 185 / 210 ms with 1 / 2 / 4 groups: one group.
#define PHK_GEN_GROUPS 4
int phk_score_fast(phk_ctx *ctx, const phk_model *m, const double *d_Q, const uint32_t *d_counts,
                   const uint32_t *d_rowsum, uint64_t N, int method, double *d_scores, uint32_t *d_status);
