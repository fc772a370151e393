// kmeans.hip -- deterministic Lloyd k-means on the device.
//
// The reference gets its centroids from scikit-learn's KMeans(n_clusters=k, random_state=10)
// (scripts/learning.py:131-146), whose result depends on the scikit-learn version.  This is an opt-in,
// version-independent alternative with a fully specified algorithm (restated in oracle/oracle.py
// :kmeans_lloyd for the tests):
//   init   : k-means++ -- first centre = point floor(u_0 * n); each next centre is drawn with probability
//            proportional to the squared distance to the nearest chosen centre, by inverse-CDF on the
//            running sum in index order with u_j = splitmix64(seed + j) / 2^64 (one candidate per step);
//   Lloyd  : assign every point to its nearest centre (float64 direct differences, ties to the lower
//            centre index); new centre = mean of its members summed in index order; an empty cluster
//            takes the point farthest from its assigned centre (ties to the lower index); stop when no
//            label changes or after max_iter sweeps.
// Everything is float64 and order-deterministic (no atomics), so the result is bit-reproducible.
#include "phk_common.h"

__device__ __forceinline__ uint64_t km_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// squared distance of every point to centre row `c`; keeps the running minimum (init) --
// one wave per point
__global__ __launch_bounds__(256) void km_update_mind2_kernel(const double *__restrict__ X, uint64_t n, uint64_t D,
                                                              const double *__restrict__ centre, double *__restrict__ mind2,
                                                              int first) {
    const int lane = threadIdx.x & 63;
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    double acc = 0.0;
    for (uint64_t d = lane; d < D; d += 64) {
        const double t = X[i * D + d] - centre[d];
        acc = fma(t, t, acc);
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
    if (lane == 0) mind2[i] = first ? acc : fmin(mind2[i], acc);
}

// inverse-CDF draw over mind2 in index order (single block; n is small: reference classes have ~2.3k rows)
__global__ __launch_bounds__(256) void km_pick_kernel(const double *__restrict__ mind2, uint64_t n, uint64_t seed, uint64_t step,
                                                      const double *__restrict__ X, uint64_t D, double *__restrict__ centres,
                                                      uint32_t *__restrict__ picked) {
    __shared__ double part[256];
    __shared__ uint64_t chosen;
    const int t = threadIdx.x;
    const double u = (double)(km_splitmix64(seed + step) >> 11) * (1.0 / 9007199254740992.0);  // [0,1)
    if (step == 0) {
        if (t == 0) chosen = (uint64_t)(u * (double)n) < n ? (uint64_t)(u * (double)n) : n - 1;
    } else {
        // each thread owns a contiguous slice; sums are formed in index order (slice sums, then prefix)
        const uint64_t per = (n + 255) / 256, lo = per * t, hi = lo + per < n ? lo + per : n;
        double s = 0.0;
        for (uint64_t i = lo; i < hi; ++i) s += mind2[i];
        part[t] = s;
        __syncthreads();
        if (t == 0) {
            double total = 0.0;
            for (int k = 0; k < 256; ++k) total += part[k];
            const double target = u * total;
            double run = 0.0;
            uint64_t pick = n - 1;
            bool done = false;
            for (int k = 0; k < 256 && !done; ++k) {
                if (run + part[k] > target) {
                    const uint64_t l2 = per * k, h2 = l2 + per < n ? l2 + per : n;
                    for (uint64_t i = l2; i < h2; ++i) {
                        run += mind2[i];
                        if (run > target) { pick = i; done = true; break; }
                    }
                    if (!done) { pick = h2 ? h2 - 1 : 0; done = true; }
                } else {
                    run += part[k];
                }
            }
            chosen = pick;
        }
    }
    __syncthreads();
    const uint64_t c = chosen;
    for (uint64_t d = t; d < D; d += 256) centres[step * D + d] = X[c * D + d];
    if (t == 0) picked[step] = (uint32_t)c;
}

// nearest centre per point: one wave per point, centres streamed from L2
__global__ __launch_bounds__(256) void km_assign_kernel(const double *__restrict__ X, uint64_t n, uint64_t D,
                                                        const double *__restrict__ centres, uint32_t k,
                                                        uint32_t *__restrict__ labels, double *__restrict__ d2own,
                                                        uint32_t *__restrict__ changed,
                                                        unsigned long long *__restrict__ gapbits = nullptr) {
    const int lane = threadIdx.x & 63;
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    double best = INFINITY, second = INFINITY;
    uint32_t bi = 0;
    for (uint32_t c = 0; c < k; ++c) {
        double acc = 0.0;
        for (uint64_t d = lane; d < D; d += 64) {
            const double t = X[i * D + d] - centres[(uint64_t)c * D + d];
            acc = fma(t, t, acc);
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
        if (acc < best) { second = best; best = acc; bi = c; }  // strict: ties keep the lower centre index
        else if (acc < second) second = acc;
    }
    if (lane == 0) {
        if (labels[i] != bi) atomicAdd(changed, 1u);
        labels[i] = bi;
        d2own[i] = best;
        // the closest call of the whole fit: (second - best) / second, the smallest over points and sweeps (non-negative
        // doubles order as their bit patterns)
        if (gapbits && k > 1 && second > 0.0 && second < INFINITY)
            atomicMin(gapbits, (unsigned long long)__double_as_longlong((second - best) / second));
    }
}

// new centre = mean of members in index order; one block per centre, thread = dimension stripe
__global__ __launch_bounds__(256) void km_update_kernel(const double *__restrict__ X, uint64_t n, uint64_t D,
                                                        const uint32_t *__restrict__ labels, double *__restrict__ centres,
                                                        uint32_t *__restrict__ sizes) {
    const uint32_t c = blockIdx.x;
    uint32_t cnt = 0;
    for (uint64_t d = threadIdx.x; d < D; d += 256) {
        double s = 0.0;
        uint32_t m = 0;
        for (uint64_t i = 0; i < n; ++i)
            if (labels[i] == c) { s += X[i * D + d]; ++m; }
        if (m) centres[(uint64_t)c * D + d] = s / (double)m;
        cnt = m;
    }
    if (threadIdx.x == 0) {
        if (D <= 0) cnt = 0;
        sizes[c] = cnt;
    }
}

// empty clusters (in index order) take the points farthest from their own centre (single thread; rare)
__global__ void km_fix_empty_kernel(const double *__restrict__ X, uint64_t n, uint64_t D, uint32_t k,
                                    uint32_t *__restrict__ labels, double *__restrict__ d2own, double *__restrict__ centres,
                                    uint32_t *__restrict__ sizes, uint32_t *__restrict__ changed) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (uint32_t c = 0; c < k; ++c) {
        if (sizes[c] != 0) continue;
        uint64_t far = 0;
        double fd = -1.0;
        for (uint64_t i = 0; i < n; ++i)
            if (d2own[i] > fd && sizes[labels[i]] > 1) { fd = d2own[i]; far = i; }
        if (fd < 0.0) continue;
        sizes[labels[far]] -= 1;
        labels[far] = c;
        sizes[c] = 1;
        d2own[far] = 0.0;
        for (uint64_t d = 0; d < D; ++d) centres[(uint64_t)c * D + d] = X[far * D + d];
        atomicAdd(changed, 1u);
    }
}

extern "C" int phk_kmeans(phk_ctx *ctx, const double *X, uint64_t n, uint64_t D, uint32_t k, uint64_t seed,
                          int max_iter, double *centroids, uint32_t *labels, int *n_iter) {
    PHK_REQUIRE(ctx && X && centroids, "phk_kmeans: NULL pointer");
    PHK_REQUIRE(k >= 1 && k <= n, "phk_kmeans: need 1 <= k <= n (k=%u, n=%llu)", k, (unsigned long long)n);
    PHK_REQUIRE(D >= 1 && max_iter >= 1, "phk_kmeans: bad D / max_iter");
    PHK_HIP(hipSetDevice(ctx->device));
    void *dX, *dC, *dm, *dl, *dsz, *dflag;
    PHK_TRY(phk_ws(ctx, WS_WIDE, n * D * 8, &dX));
    PHK_TRY(phk_ws(ctx, WS_Q64, (uint64_t)k * D * 8, &dC));
    PHK_TRY(phk_ws(ctx, WS_SCORES, n * 8, &dm));
    PHK_TRY(phk_ws(ctx, WS_COUNTS, n * 4 + (uint64_t)k * 4, &dl));
    PHK_TRY(phk_ws(ctx, WS_FLAGS, 64, &dflag));
    dsz = (uint32_t *)dl + n;
    PHK_HIP(hipMemcpyAsync(dX, X, n * D * 8, hipMemcpyHostToDevice, ctx->stream));
    PHK_HIP(hipMemsetAsync(dl, 0xFF, n * 4, ctx->stream));
    const unsigned wblocks = (unsigned)phk_div_up(n, 4);
    // k-means++ initialisation (picked indices land in the first k entries of the sizes buffer, unused until Lloyd)
    for (uint32_t j = 0; j < k; ++j) {
        PHK_LAUNCH(ctx, "km_pick_kernel",
                   km_pick_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>((const double *)dm, n, seed, j, (const double *)dX, D,
                                                                        (double *)dC, (uint32_t *)dsz));
        PHK_LAUNCH(ctx, "km_update_mind2_kernel",
                   km_update_mind2_kernel<<<dim3(wblocks), dim3(256), 0, ctx->stream>>>(
                       (const double *)dX, n, D, (const double *)dC + (uint64_t)j * D, (double *)dm, j == 0 ? 1 : 0));
    }
    int it = 0;
    for (; it < max_iter; ++it) {
        PHK_HIP(hipMemsetAsync(dflag, 0, 4, ctx->stream));
        PHK_LAUNCH(ctx, "km_assign_kernel",
                   km_assign_kernel<<<dim3(wblocks), dim3(256), 0, ctx->stream>>>((const double *)dX, n, D, (const double *)dC, k,
                                                                                (uint32_t *)dl, (double *)dm, (uint32_t *)dflag));
        PHK_LAUNCH(ctx, "km_update_kernel",
                   km_update_kernel<<<dim3(k), dim3(256), 0, ctx->stream>>>((const double *)dX, n, D, (const uint32_t *)dl,
                                                                          (double *)dC, (uint32_t *)dsz));
        PHK_LAUNCH(ctx, "km_fix_empty_kernel",
                   km_fix_empty_kernel<<<dim3(1), dim3(64), 0, ctx->stream>>>((const double *)dX, n, D, k, (uint32_t *)dl, (double *)dm,
                                                                            (double *)dC, (uint32_t *)dsz, (uint32_t *)dflag));
        uint32_t changed = 1;
        PHK_HIP(hipMemcpyAsync(&changed, dflag, 4, hipMemcpyDeviceToHost, ctx->stream));
        PHK_HIP(hipStreamSynchronize(ctx->stream));
        if (changed == 0) { ++it; break; }
    }
    PHK_HIP(hipMemcpyAsync(centroids, dC, (uint64_t)k * D * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (labels) PHK_HIP(hipMemcpyAsync(labels, dl, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PHK_HIP(hipStreamSynchronize(ctx->stream));
    if (n_iter) *n_iter = it;
    return PHK_OK;
}

// ---- Lloyd sweeps from GIVEN initial centres, with scikit-learn's stopping rule (reference-equal centroids
// without the host fit) ----
// sum over the centres of |new - old|^2, and the number of empty clusters: one block
__global__ __launch_bounds__(256) void km_shift_kernel(const double *__restrict__ a, const double *__restrict__ b, uint64_t count,
                                                       const uint32_t *__restrict__ sizes, uint32_t k, double *__restrict__ out_shift,
                                                       uint32_t *__restrict__ out_empty) {
    __shared__ double part[256];
    double s = 0.0;
    for (uint64_t i = threadIdx.x; i < count; i += 256) {
        const double d = a[i] - b[i];
        s = fma(d, d, s);
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 256; ++i) t += part[i];
        *out_shift = t;
        uint32_t e = 0;
        for (uint32_t c = 0; c < k; ++c) e += sizes[c] == 0 ? 1u : 0u;
        *out_empty = e;
    }
}

extern "C" int phk_kmeans_lloyd(phk_ctx *ctx, const double *X, uint64_t n, uint64_t D, uint32_t k, const double *init, double tol,
                                int max_iter, double *centres_out, uint32_t *labels, int *n_iter, int *n_empty, double *min_gap) {
    PHK_REQUIRE(ctx && X && init && labels, "phk_kmeans_lloyd: NULL pointer");
    PHK_REQUIRE(k >= 1 && k <= n, "phk_kmeans_lloyd: need 1 <= k <= n (k=%u, n=%llu)", k, (unsigned long long)n);
    PHK_REQUIRE(D >= 1 && max_iter >= 1 && tol >= 0.0, "phk_kmeans_lloyd: bad D / max_iter / tol");
    PHK_HIP(hipSetDevice(ctx->device));
    void *dX, *dC, *dm, *dl, *dsz, *dflag;
    PHK_TRY(phk_ws(ctx, WS_WIDE, n * D * 8, &dX));
    PHK_TRY(phk_ws(ctx, WS_Q64, 2 * (uint64_t)k * D * 8, &dC));      // the centres and the previous sweep's
    PHK_TRY(phk_ws(ctx, WS_SCORES, n * 8, &dm));
    PHK_TRY(phk_ws(ctx, WS_COUNTS, 2 * n * 4 + (uint64_t)k * 4, &dl));
    PHK_TRY(phk_ws(ctx, WS_FLAGS, 64, &dflag));
    dsz = (uint32_t *)dl + 2 * n;
    double *cen = (double *)dC, *old = cen + (uint64_t)k * D;
    PHK_HIP(hipMemcpyAsync(dX, X, n * D * 8, hipMemcpyHostToDevice, ctx->stream));
    PHK_HIP(hipMemcpyAsync(cen, init, (uint64_t)k * D * 8, hipMemcpyHostToDevice, ctx->stream));
    PHK_HIP(hipMemsetAsync(dl, 0xFF, n * 4, ctx->stream));
    unsigned long long *gapbits = (unsigned long long *)((uint32_t *)dflag + 4);   // (behind the 16 bytes zeroed per sweep)
    {
        const double inf = INFINITY;
        PHK_HIP(hipMemcpyAsync(gapbits, &inf, 8, hipMemcpyHostToDevice, ctx->stream));
    }
    const unsigned wblocks = (unsigned)phk_div_up(n, 4);
    struct { uint32_t changed, empty; double shift; } flag;
    int it = 0, empties = 0;
    bool strict = false;
    // scikit-learn's _kmeans_single_lloyd: E-step against the current centres, M-step, then "labels unchanged" (strict
    // convergence) or "total squared centre shift <= tol"; in the second case one more E-step so that the labels match the
    // final centres.  An empty cluster keeps its centre here (scikit-learn relocates it): reported, and the caller falls back.
    for (; it < max_iter; ++it) {
        PHK_HIP(hipMemsetAsync(dflag, 0, 16, ctx->stream));
        PHK_HIP(hipMemcpyAsync(old, cen, (uint64_t)k * D * 8, hipMemcpyDeviceToDevice, ctx->stream));
        PHK_LAUNCH(ctx, "km_assign_kernel",
                   km_assign_kernel<<<dim3(wblocks), dim3(256), 0, ctx->stream>>>((const double *)dX, n, D, cen, k, (uint32_t *)dl,
                                                                                (double *)dm, (uint32_t *)dflag, gapbits));
        PHK_LAUNCH(ctx, "km_update_kernel",
                   km_update_kernel<<<dim3(k), dim3(256), 0, ctx->stream>>>((const double *)dX, n, D, (const uint32_t *)dl, cen,
                                                                          (uint32_t *)dsz));
        PHK_LAUNCH(ctx, "km_shift_kernel",
                   km_shift_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(cen, old, (uint64_t)k * D, (const uint32_t *)dsz, k,
                                                                         (double *)((uint32_t *)dflag + 2), (uint32_t *)dflag + 1));
        PHK_HIP(hipMemcpyAsync(&flag, dflag, 16, hipMemcpyDeviceToHost, ctx->stream));
        PHK_HIP(hipStreamSynchronize(ctx->stream));
        empties += (int)flag.empty;
        if (flag.changed == 0) { strict = true; ++it; break; }
        if (flag.shift <= tol) { ++it; break; }
    }
    if (!strict) {
        PHK_LAUNCH(ctx, "km_assign_kernel",
                   km_assign_kernel<<<dim3(wblocks), dim3(256), 0, ctx->stream>>>((const double *)dX, n, D, cen, k, (uint32_t *)dl,
                                                                                (double *)dm, (uint32_t *)dflag, gapbits));
    }
    double gap = INFINITY;
    PHK_HIP(hipMemcpyAsync(&gap, gapbits, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (centres_out) PHK_HIP(hipMemcpyAsync(centres_out, cen, (uint64_t)k * D * 8, hipMemcpyDeviceToHost, ctx->stream));
    PHK_HIP(hipMemcpyAsync(labels, dl, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PHK_HIP(hipStreamSynchronize(ctx->stream));
    if (n_iter) *n_iter = it;
    if (n_empty) *n_empty = empties;
    if (min_gap) *min_gap = gap;
    return PHK_OK;
}
