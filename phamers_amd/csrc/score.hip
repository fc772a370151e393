// score.hip -- phage scoring of normalised k-mer rows against the reference matrix (gfx950).
//
// Replaces the arithmetic of learning.knn (scripts/learning.py:118-128, scikit-learn brute
// force Euclidean k-NN + uniform majority vote), the nearest-centroid loop of
// phamer_scorer.kmeans_score_points (scripts/phamer.py:250-256, learning.closest_to
// scripts/learning.py:59-66), phamer_scorer.proximity_metric (scripts/phamer.py:198-210) and
// combo_score_points (scripts/phamer.py:303-313).
//
// This file holds the EXACT float64 path: direct-difference squared distances
// (scripts/learning.py:56 form) tiled through LDS, then per-query selection.  It serves any
// D / M / kn and is the fall-back of the MFMA path (score_mfma.hip) for queries whose
// candidate margin cannot be certified.
#include <stdlib.h>

#include "phk_common.h"
#include "score_model.h"

// ------------------------------------------------------------------------------------
// squared distances, float64, direct differences:  out[q][x] = sum_d (Q[q][d]-X[x][d])^2
// block = 256 threads, tile = 64 queries x 64 rows, 4x4 per thread, K chunk = 16
// ------------------------------------------------------------------------------------
#define DT 64
#define DK 16
__global__ __launch_bounds__(256) void phk_dist2_f64_kernel(const double *__restrict__ Q, uint64_t nq,
                                                            const double *__restrict__ X, uint64_t nx,
                                                            uint64_t D, double *__restrict__ out,
                                                            uint64_t ld_out) {
    __shared__ double Qs[DK][DT + 2];
    __shared__ double Xs[DK][DT + 2];
    const int t = threadIdx.x;
    const int tq = t >> 4, tx = t & 15;  // 16 x 16 threads, each 4 q x 4 x
    const uint64_t q0 = (uint64_t)blockIdx.y * DT, x0 = (uint64_t)blockIdx.x * DT;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;

    const int lr = t >> 2, lk = (t & 3) * 4;  // loader: row lr, 4 consecutive k at lk
    for (uint64_t k0 = 0; k0 < D; k0 += DK) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t k = k0 + lk + j;
            const uint64_t qr = q0 + lr, xr = x0 + lr;
            Qs[lk + j][lr] = (qr < nq && k < D) ? Q[qr * D + k] : 0.0;
            Xs[lk + j][lr] = (xr < nx && k < D) ? X[xr * D + k] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < DK; ++kk) {
            double qv[4], xv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) qv[i] = Qs[kk][tq * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = Xs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double d = qv[i] - xv[j];
                    acc[i][j] = fma(d, d, acc[i][j]);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint64_t q = q0 + tq * 4 + i;
        if (q >= nq) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t x = x0 + tx * 4 + j;
            if (x < nx) out[q * ld_out + x] = acc[i][j];
        }
    }
}

// ------------------------------------------------------------------------------------
// k-NN vote: one wavefront per query over its distance row.  Neighbours are taken in
// (distance, index) lexicographic order, i.e. ties go to the lower train index.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void phk_mask_dist_kernel(double *__restrict__ dist, uint64_t nq, uint64_t M,
                                                            const uint8_t *__restrict__ mask) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * M) return;
    if (mask[i % M] && dist[i] == dist[i]) dist[i] = __builtin_inf();   // (a NaN query row stays NaN)
}

__global__ __launch_bounds__(256) void phk_knn_vote_kernel(const double *__restrict__ dist, uint64_t nq,
                                                           uint64_t M, uint64_t ld,
                                                           const uint8_t *__restrict__ labels, int kn,
                                                           double *__restrict__ knn_out,
                                                           uint32_t *__restrict__ nan_rows) {
    const int lane = threadIdx.x & 63;
    const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (q >= nq) return;
    const double *row = dist + q * ld;
    if (row[0] != row[0]) {  // NaN query row (zero-count contig): every distance is NaN
        if (lane == 0) {
            knn_out[q] = __builtin_nan("");
            if (nan_rows) atomicAdd(nan_rows, 1u);
        }
        return;
    }
    double last_d = -1.0;
    uint64_t last_i = 0;
    bool first = true;
    int votes = 0;
    for (int r = 0; r < kn; ++r) {
        double bd = __builtin_inf();
        uint64_t bi = ~0ull;
        for (uint64_t j = lane; j < M; j += 64) {
            const double d = row[j];
            const bool after = first || d > last_d || (d == last_d && j > last_i);
            if (after && (d < bd || (d == bd && j < bi))) {
                bd = d;
                bi = j;
            }
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            const double od = __shfl_xor(bd, s);
            const uint64_t oi = __shfl_xor(bi, s);
            if (od < bd || (od == bd && oi < bi)) {
                bd = od;
                bi = oi;
            }
        }
        last_d = bd;
        last_i = bi;
        first = false;
        votes += labels[bi] ? 1 : 0;
    }
    if (lane == 0) knn_out[q] = (2 * votes > kn) ? 1.0 : -1.0;
}

// ------------------------------------------------------------------------------------
// nearest positive / negative centroid + proximity metric; one thread per query
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void phk_centroid_metric_kernel(const double *__restrict__ dist, uint64_t nq,
                                                                  uint64_t n_cpos, uint64_t n_cneg,
                                                                  uint64_t ld, double *__restrict__ cen_out,
                                                                  uint32_t *__restrict__ nan_rows,
                                                                  int count_nan) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const double *row = dist + q * ld;
    if (row[0] != row[0]) {
        cen_out[q] = __builtin_nan("");
        if (count_nan && nan_rows) atomicAdd(nan_rows, 1u);
        return;
    }
    double bp = __builtin_inf(), bn = __builtin_inf();
    for (uint64_t c = 0; c < n_cpos; ++c) bp = row[c] < bp ? row[c] : bp;
    for (uint64_t c = 0; c < n_cneg; ++c) bn = row[n_cpos + c] < bn ? row[n_cpos + c] : bn;
    const double ep = sqrt(bp), en = sqrt(bn);
    cen_out[q] = tanh((en - ep) / (ep + en));  // scripts/phamer.py:206-209
}

__global__ __launch_bounds__(256) void phk_combine_kernel(const double *__restrict__ a,
                                                          const double *__restrict__ b, uint64_t n,
                                                          double *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (a ? a[i] : 0.0) + (b ? b[i] : 0.0);  // scripts/phamer.py:313
}

// ------------------------------------------------------------------------------------
// model
// ------------------------------------------------------------------------------------
extern "C" int phk_model_create(phk_ctx *ctx, const double *pos, uint64_t n_pos, const double *neg,
                                uint64_t n_neg, const double *cpos, uint64_t n_cpos,
                                const double *cneg, uint64_t n_cneg, uint64_t D, int kn,
                                phk_model **out) {
    PHK_ENTER(ctx, "phk_model_create");
    PHK_REQUIRE(out, "phk_model_create: NULL out");
    PHK_REQUIRE(D >= 1 && pos && neg && n_pos + n_neg >= 1, "phk_model_create: empty reference data");
    PHK_REQUIRE(kn >= 1 && (uint64_t)kn <= n_pos + n_neg && kn <= 64,
                "phk_model_create: k_neighbors=%d out of range (1..min(64, rows))", kn);
    PHK_REQUIRE((n_cpos == 0) == (n_cneg == 0), "phk_model_create: give both centroid sets or neither");
    PHK_REQUIRE(n_cpos == 0 || (cpos && cneg), "phk_model_create: NULL centroid pointer");
    phk_model *m = new phk_model();
    m->D = D;
    m->n_pos = n_pos;
    m->n_neg = n_neg;
    m->M = n_pos + n_neg;
    m->n_cpos = n_cpos;
    m->n_cneg = n_cneg;
    m->kn = kn;
    int rc = PHK_OK;
    do {
        if (hipMalloc(&m->d_R64, m->M * D * sizeof(double)) != hipSuccess) { rc = PHK_ERR_NOMEM; break; }
        if (hipMalloc(&m->d_labels, m->M) != hipSuccess) { rc = PHK_ERR_NOMEM; break; }
        if (hipMemcpy(m->d_R64, pos, n_pos * D * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(m->d_R64 + n_pos * D, neg, n_neg * D * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
            rc = PHK_ERR_HIP; break;
        }
        std::vector<uint8_t> lab(m->M, 0);
        for (uint64_t i = 0; i < n_pos; ++i) lab[i] = 1;  // scripts/phamer.py:187
        if (hipMemcpy(m->d_labels, lab.data(), m->M, hipMemcpyHostToDevice) != hipSuccess) { rc = PHK_ERR_HIP; break; }
        if (n_cpos) {
            if (hipMalloc(&m->d_C64, (n_cpos + n_cneg) * D * sizeof(double)) != hipSuccess) { rc = PHK_ERR_NOMEM; break; }
            if (hipMemcpy(m->d_C64, cpos, n_cpos * D * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(m->d_C64 + n_cpos * D, cneg, n_cneg * D * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
                rc = PHK_ERR_HIP; break;
            }
        }
        rc = phk_model_build_fast(ctx, m, pos, neg, cpos, cneg);
    } while (0);
    if (rc != PHK_OK) {
        if (rc != PHK_ERR_ARG) phk_set_error("phk_model_create: device allocation / copy failed (rc=%d)", rc);
        phk_model_destroy(ctx, m);
        return rc;
    }
    *out = m;
    return PHK_OK;
}

extern "C" int phk_model_destroy(phk_ctx *ctx, phk_model *m) {
    if (!m) return PHK_OK;
    if (ctx) {  // kernels that still read the model may be in flight on the context's stream
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (m->d_R64) (void)hipFree(m->d_R64);
    if (m->d_labels) (void)hipFree(m->d_labels);
    if (m->d_C64) (void)hipFree(m->d_C64);
    phk_model_free_fast(m);
    delete m;
    return PHK_OK;
}

// ------------------------------------------------------------------------------------
// exact scoring of a batch of float64 rows
// ------------------------------------------------------------------------------------
int phk_score_exact_batch(phk_ctx *ctx, const phk_model *m, const double *d_Q, uint64_t nq, int method,
                          double *d_knn, double *d_cen, uint32_t *d_status) {
    const uint64_t D = m->D;
    const bool want_knn = method & PHK_METHOD_KNN, want_cen = method & PHK_METHOD_KMEANS;
    if (want_knn) {
        void *dist;
        PHK_TRY(phk_ws(ctx, WS_DIST, nq * m->M * sizeof(double), &dist));
        dim3 grid((unsigned)phk_div_up(m->M, DT), (unsigned)phk_div_up(nq, DT));
        PHK_LAUNCH(ctx, "phk_dist2_f64_kernel",
                   phk_dist2_f64_kernel<<<grid, dim3(256), 0, ctx->stream>>>(d_Q, nq, m->d_R64, m->M, D,
                                                                            (double *)dist, m->M));
        if (m->has_mask) {   // cross-validation fold: excluded train rows are infinitely far
            PHK_LAUNCH(ctx, "phk_mask_dist_kernel",
                       phk_mask_dist_kernel<<<dim3((unsigned)phk_div_up(nq * m->M, 256)), dim3(256), 0, ctx->stream>>>(
                           (double *)dist, nq, m->M, m->d_col_mask));
        }
        PHK_LAUNCH(ctx, "phk_knn_vote_kernel",
                   phk_knn_vote_kernel<<<dim3((unsigned)phk_div_up(nq, 4)), dim3(256), 0, ctx->stream>>>(
                       (const double *)dist, nq, m->M, m->M, m->d_labels, m->kn, d_knn, d_status));
    }
    if (want_cen) {
        const uint64_t nc = m->n_cpos + m->n_cneg;
        void *dist;
        PHK_TRY(phk_ws(ctx, WS_DIST, nq * (want_knn ? (m->M > nc ? m->M : nc) : nc) * sizeof(double), &dist));
        dim3 grid((unsigned)phk_div_up(nc, DT), (unsigned)phk_div_up(nq, DT));
        PHK_LAUNCH(ctx, "phk_dist2_f64_kernel",
                   phk_dist2_f64_kernel<<<grid, dim3(256), 0, ctx->stream>>>(d_Q, nq, m->d_C64, nc, D,
                                                                            (double *)dist, nc));
        PHK_LAUNCH(ctx, "phk_centroid_metric_kernel",
                   phk_centroid_metric_kernel<<<dim3((unsigned)phk_div_up(nq, 256)), dim3(256), 0, ctx->stream>>>(
                       (const double *)dist, nq, m->n_cpos, m->n_cneg, nc, d_cen, d_status, want_knn ? 0 : 1));
    }
    return PHK_OK;
}

__global__ __launch_bounds__(256) void phk_sqrt_kernel(double *__restrict__ x, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = sqrt(x[i]);
}

// learning.distances (scripts/learning.py:47-56): out[q][x] = || Q[q] - X[x] ||_2 in the reference's direct-difference
// form, float64.  Host pointers; row blocks of Q so that the distance tile stays within a fixed workspace.
extern "C" int phk_distances(phk_ctx *ctx, const double *Q, uint64_t N, const double *X, uint64_t M, uint64_t D, double *out) {
    PHK_ENTER(ctx, "phk_distances");
    if (N == 0 || M == 0) return PHK_OK;
    PHK_REQUIRE(Q && X && out && D > 0, "phk_distances: NULL pointer / zero dimension");
    void *d_x, *d_q, *d_o;
    uint64_t rows = (256ull << 20) / (M * sizeof(double));   // <= 256 MiB of distances per block of queries
    rows = rows < 1 ? 1 : (rows > N ? N : rows);
    while (rows > 1 && rows * D * sizeof(double) > (1ull << 30)) rows >>= 1;
    PHK_TRY(phk_ws(ctx, WS_WIDE, M * D * sizeof(double), &d_x));
    PHK_TRY(phk_ws(ctx, WS_Q64, rows * D * sizeof(double), &d_q));
    PHK_TRY(phk_ws(ctx, WS_DIST, rows * M * sizeof(double), &d_o));
    PHK_HIP(hipMemcpyAsync(d_x, X, M * D * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    for (uint64_t s = 0; s < N; s += rows) {
        const uint64_t nb = N - s < rows ? N - s : rows;
        PHK_HIP(hipMemcpyAsync(d_q, Q + s * D, nb * D * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        dim3 grid((unsigned)phk_div_up(M, DT), (unsigned)phk_div_up(nb, DT));
        PHK_LAUNCH(ctx, "phk_dist2_f64_kernel",
                   phk_dist2_f64_kernel<<<grid, dim3(256), 0, ctx->stream>>>((const double *)d_q, nb, (const double *)d_x, M, D, (double *)d_o, M));
        PHK_LAUNCH(ctx, "phk_sqrt_kernel", phk_sqrt_kernel<<<dim3((unsigned)phk_div_up(nb * M, 256)), dim3(256), 0, ctx->stream>>>((double *)d_o, nb * M));
        PHK_HIP(hipMemcpyAsync(out + s * M, d_o, nb * M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PHK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return PHK_OK;
}

static int check_method(const phk_model *m, int method) {
    PHK_REQUIRE(method == PHK_METHOD_KNN || method == PHK_METHOD_KMEANS || method == PHK_METHOD_COMBO,
                "phk_score: unknown method %d", method);
    PHK_REQUIRE(!(method & PHK_METHOD_KMEANS) || m->n_cpos > 0,
                "phk_score: method needs centroids but the model was created without them");
    return PHK_OK;
}

// Scores N rows given either as float64 rows (d_Q) or as uint32 count rows (d_counts, normalised
// on the fly as kmer.normalize_counts would).  Exactly one of d_Q / d_counts is non-NULL.
int phk_score_rows(phk_ctx *ctx, const phk_model *m, const double *d_Q, const uint32_t *d_counts,
                   const uint32_t *d_rowsum, uint64_t N, int method, double *d_scores, uint32_t *d_status) {
    PHK_REQUIRE(m && d_scores && (d_Q || d_counts), "phk_score: NULL pointer");
    PHK_TRY(check_method(m, method));
    // (phk_count_score_dev: the count planner's kernel has zeroed the NaN counter -- and the statistics totals -- already)
    const bool prezeroed = ctx->score_totals_zeroed;
    ctx->score_totals_zeroed = prezeroed && !ctx->score_totals_only_status;   // what phk_score_fast may rely on: the totals
    if (d_status && !prezeroed) PHK_HIP(hipMemsetAsync(d_status, 0, sizeof(uint32_t), ctx->stream));
    if (N == 0) {
        ctx->score_totals_zeroed = false;
        return PHK_OK;
    }
    const uint64_t D = m->D;

    // PHK_FORCE_EXACT=1 routes every model through the float64 path (used by the parity tests to
    // cross-check the two GPU paths against each other)
    ctx->last_score_fast = false;
    if (!(phk_model_has_fast(m) && !ctx->knobs.force_exact)) ctx->score_totals_zeroed = false;
    if (phk_model_has_fast(m) && !ctx->knobs.force_exact) {
        ctx->last_score_fast = true;
        return phk_score_fast(ctx, m, d_Q, d_counts, d_rowsum, N, method, d_scores, d_status);
    }

    // exact path in batches sized to a 512 MiB distance scratch
    const uint64_t widest = m->M > (m->n_cpos + m->n_cneg) ? m->M : (m->n_cpos + m->n_cneg);
    uint64_t B = (512ull << 20) / (widest * sizeof(double));
    B = B < 64 ? 64 : (B / 64) * 64;
    if (B > N) B = N;
    void *tmp;
    PHK_TRY(phk_ws(ctx, WS_SCORES, 2 * B * sizeof(double), &tmp));
    double *d_knn = (double *)tmp, *d_cen = d_knn + B;
    void *q64 = nullptr;
    if (d_counts) PHK_TRY(phk_ws(ctx, WS_Q64, B * D * sizeof(double), &q64));
    for (uint64_t s = 0; s < N; s += B) {
        const uint64_t nb = N - s < B ? N - s : B;
        const double *q = d_Q ? d_Q + s * D : (const double *)q64;
        if (d_counts) PHK_TRY(phk_launch_normalize_u32(ctx, d_counts + s * D, nb, D, (double *)q64));
        PHK_TRY(phk_score_exact_batch(ctx, m, q, nb, method, d_knn, d_cen, d_status));
        PHK_LAUNCH(ctx, "phk_combine_kernel",
                   phk_combine_kernel<<<dim3((unsigned)phk_div_up(nb, 256)), dim3(256), 0, ctx->stream>>>(
                       (method & PHK_METHOD_KNN) ? d_knn : nullptr,
                       (method & PHK_METHOD_KMEANS) ? d_cen : nullptr, nb, d_scores + s));
    }
    return PHK_OK;
}
