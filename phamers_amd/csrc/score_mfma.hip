// score_mfma.hip -- the fast scoring path's model and driver (k = 4 .. 6: D = 256 .. 4096):
//
//   1. a proposal pass (score_f16.hip, score_i8.hip): an MFMA "distance GEMM" of the centred count rows against the centred
//      reference rows + centroids with a fused running top-4 per (query, column segment, half-list) kept in registers --
//      nothing of the N x (M + C) product ever reaches memory.  This is the dense contraction behind scikit-learn's
//      brute-force k-NN (scripts/learning.py:127) and behind the nearest-centroid search (scripts/learning.py:59-66).
//   2. the decision stage (score_decide.hip, score_rerank.hip): per query, certify the candidate order against a rigorous
//      error bound; where certified take the vote from the labels, otherwise (and always for the two nearest-centroid
//      distances the proximity metric needs, scripts/phamer.py:198-210) recompute direct-difference float64 distances for
//      the candidates.  A query whose candidate set cannot be certified to contain the true neighbours is queued for
//   3. the exact float64 brute force over every column (score_fallback.hip).
//
// So the MFMA pass only ever PROPOSES candidates; every emitted number is decided by float64 arithmetic of the same form as
// the reference's (direct differences), or by a certified ordering.
//
// Ranking quantity.  With mu = mean train row, r' = r - mu, q' = q - mu (distances are translation invariant) the proposal
// maximises  v = q'.r' - |r'|^2/2  = (|q'|^2 - |q-r|^2)/2; the -|r'|^2/2 term rides through the MFMA as one more k-step
// (score_f16.hip: phk_bias_pieces).
//
// This file: the host-side model build (centring, norms, the float64 copies the decision stage reads) and phk_score_fast,
// which sizes the workspaces, fills the parameter blocks (score_decide.h) and enqueues the launch chain of a batch.
#include <stdlib.h>

#include <cmath>
#include <vector>

#include "score_decide.h"

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
// driver
// ------------------------------------------------------------------------------------
// One decision pass over the lists of a (sub-)batch.  src_kind: 0 = count rows, 1 = normalised float64 rows.
static int launch_rerank(phk_ctx *ctx, int src_kind, unsigned blocks, const void *src, const RerankParams &p) {
    // phk_rerank_kernel walks the queries grid-stride (its workgroups keep the training mean in LDS): a few workgroups per CU
    const unsigned cap = (unsigned)ctx->num_cus * 16u;
    const unsigned wblocks = (p.D >= 2048 && blocks > cap) ? cap : blocks;
    if (p.D == FAST_D) {
        const char rr = ctx->knobs.rerank;
        if (rr == 'w')   // one wave per query (the general kernel), for A/B comparison
            return phk_launch_rerank_wave(ctx, src_kind, 1, false, wblocks, src, p);
        if (rr == 'g')   // four queries per wave for every query (the decision kernel off)
            return phk_launch_rerank16(ctx, src_kind, 0, (unsigned)phk_div_up(p.N, 16), src, p);
        // one lane per query for what the margin test certifies, then four per wave for the rest
        PHK_TRY(phk_launch_decide(ctx, src_kind, src, p));
        return phk_launch_rerank16(ctx, src_kind, 1, (unsigned)phk_div_up(p.N, 16), src, p);
    }
    if (!phk_fast_supports_dim(p.D)) {
        phk_set_error("phk_score: no decision kernel for D = %llu", (unsigned long long)p.D);
        return PHK_ERR_UNSUPPORTED;
    }
    const int dsub = (int)(p.D / 256);
    const bool i8h = src_kind == 0 && p.L8;
    if (i8h && p.rowsum && ctx->knobs.rerank != 'w') {
        // the lane-per-query decision kernel first; what it hands on, listed, to the wave-per-query kernel
        PHK_TRY(phk_launch_decide_gen(ctx, dsub, static_cast<const uint32_t *>(src), p));
        RerankParams pl = p;
        pl.slow_back = 2;
        return phk_launch_rerank_wave(ctx, 0, dsub, true, wblocks, src, pl);
    }
    return phk_launch_rerank_wave(ctx, src_kind, dsub, i8h, wblocks, src, p);
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
            PHK_TRY(phk_launch_decide_h(ctx, d_knn, d_cen, dg, db, (const uint32_t *)src, pd, hp));
            // what it passes on is decided from the same lists by exact candidate distances where possible.  For that
            // kernel the lists' error model is the count-exact one plus the missing low product, |q'| |lo_j| / S with
            // |lo_j| <= 2^-11 (1 + 2^-11) S |r'_j| + sqrt(D) 2^-25 (half an ulp of the high part per element; the
            // second term covers fp16 subnormals): 2^-11 / u = 8192 more on cP, the absolute term doubled
            RerankParams ph = pd;
            ph.eb_cP += 8192.0 * (1.0 + 1.0 / 2048.0) + 1.0;
            ph.eb_abs *= 2.0;
            ph.slow_back = 3;   // front and back list in one launch
            // (waves: sub_lists x (sub_cap / 4 + 2) local ones, four per workgroup)
            PHK_TRY(phk_launch_rerank16(ctx, 0, 1, (unsigned)phk_div_up((uint64_t)PHK_SUB_LISTS * (pd.sub_cap / 4 + 2), 4), src, ph));
        } else if (hi_gen) {
            HiParams hp;
            hp.lo16 = m->d_lo16;
            for (int sg = 0; sg < 3; ++sg) {
                for (int i = 0; i <= 64; ++i) hp.lam_tab[sg][i] = m->lam_tab[sg][i];
                hp.lam_r0[sg] = m->lam_r0[sg];
                hp.lam_inv_step[sg] = 1.0 / m->lam_step[sg];
            }
            PHK_TRY(phk_launch_rerank_h(ctx, (int)(D / 256), rblocks, (const uint32_t *)src, p, hp));
            // what it passes on: the one-wave-per-query kernel on the listed queries, the lists under the high-part error
            // model (see the k = 4 path above)
            RerankParams ph = p;
            ph.eb_cP += 8192.0 * (1.0 + 1.0 / 2048.0) + 1.0;
            ph.eb_abs *= 2.0;
            ph.slow_back = 2;
            PHK_TRY(launch_rerank(ctx, 0, rblocks, src, ph));
        } else {
            RerankParams pr = p;
            if (pend && ctx->knobs.rerank != 'w') {   // general D: the proximity metric is finished by a lane-per-query kernel
                PHK_HIP(hipMemsetAsync(pend, 0xFF, nb * 2 * sizeof(double), ctx->stream));   // NaN: not decided here
                pr.pend = pend;
            }
            PHK_TRY(launch_rerank(ctx, d_counts ? 0 : 1, rblocks, src, pr));
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
                        PHK_TRY(phk_launch_append_queue(ctx, list, (uint32_t)nq, fb_list, fbc, q2c + pass));
                        continue;
                    }
                    void *sub;
                    PHK_TRY(phk_ws(ctx, WS_SUB, nq * (D + 1) * sizeof(uint32_t), &sub));
                    uint32_t *sub_counts = (uint32_t *)sub, *sub_sum = sub_counts + nq * D;
                    PHK_TRY(phk_launch_gather_list_rows(ctx, (const uint32_t *)src, rsum, list, nq, D, sub_counts, sub_sum));
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
                    PHK_TRY(launch_rerank(ctx, 0, (unsigned)phk_div_up(nq, 4), sub_counts, p2));
                }
            }
            if (pr.pend) PHK_TRY(phk_launch_finish_cen(ctx, nb, pend, d_scores + s));
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
            PHK_TRY(phk_launch_rerank16(ctx, 0, 2, (unsigned)phk_div_up(cap, 16), src, p2));
            pf = p2;
            pf.N = nb;
            pf.fb_rec_cap = nb_max * FB_CHUNKS;
        }
        if (D == FAST_D) {
            PHK_TRY(phk_launch_fallback_partial(ctx, d_counts ? 0 : 1, fb_lds, src, pf));
        } else {
            // general D: the queued queries eight at a time against a chunk of the reference (see the kernel)
            pf.fb_rec_cap = nb_max * FB_CHUNKS;
            PHK_TRY(phk_launch_fallback_group(ctx, d_counts ? 0 : 1, (int)(D / 256), src, pf));
        }
        pf.clean_counters = fbc;   // the set's last kernel leaves its counters and stripes zeroed for batch b + 2 / the next call
        pf.clean_stripes = stripes;
        PHK_TRY(phk_launch_fallback_merge(ctx, pf));
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
int phk_score_mfma_init_device(phk_ctx *ctx) { return phk_score_fallback_init_device(ctx); }

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
