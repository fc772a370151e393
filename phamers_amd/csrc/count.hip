// count.hip -- ASCII -> 2-bit packer, per-contig 4^k histogram, widening and row
// normalisation kernels (gfx950).
//
// Replaces the arithmetic of kmer.sequence_to_integers (scripts/kmer.py:183-196),
// kmer.count_string's window loop (scripts/kmer.py:47-50) and kmer.normalize_counts
// (scripts/kmer.py:209-221).  Packed-stream layout: include/phamers_hip.h.
#include "phk_common.h"

// ------------------------------------------------------------------------------------
// pack: one thread per 32 bases -> two packed words + one mask word
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int phk_code_of(uint32_t ch, uint32_t sym) {
    // sym = symbols4 packed little-endian: code i <-> byte i.  Case-sensitive exact match
    // (scripts/kmer.py:190-191: every character outside `symbols` is a no-read).
    int code = -1;
    code = (ch == (sym & 0xFF)) ? 0 : code;
    code = (ch == ((sym >> 8) & 0xFF)) ? 1 : code;
    code = (ch == ((sym >> 16) & 0xFF)) ? 2 : code;
    code = (ch == (sym >> 24)) ? 3 : code;
    return code;
}

__global__ __launch_bounds__(256) void phk_pack_kernel(const uint8_t *__restrict__ bases, uint64_t T,
                                                       uint32_t sym, uint32_t *__restrict__ packed,
                                                       uint32_t *__restrict__ mask,
                                                       uint64_t packed_words, uint64_t mask_words,
                                                       uint32_t *any_invalid) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= mask_words) return;
    uint64_t g0 = t * 32;
    uint32_t w[2] = {0u, 0u};
    uint32_t m = 0u;
    bool bad = false;
    if (g0 + 32 <= T) {
        const uint4 *p = reinterpret_cast<const uint4 *>(bases + g0);  // 32-byte aligned
        uint4 v[2] = {p[0], p[1]};
        const uint32_t *d = reinterpret_cast<const uint32_t *>(v);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            uint32_t ch = (d[i >> 2] >> (8 * (i & 3))) & 0xFF;
            int code = phk_code_of(ch, sym);
            bool ok = code >= 0;
            bad |= !ok;
            w[i >> 4] |= (uint32_t)(ok ? code : 0) << (30 - 2 * (i & 15));
            m |= (uint32_t)ok << (31 - i);
        }
    } else {
        for (int i = 0; i < 32; ++i) {
            uint64_t g = g0 + i;
            if (g < T) {
                int code = phk_code_of(bases[g], sym);
                bool ok = code >= 0;
                bad |= !ok;
                w[i >> 4] |= (uint32_t)(ok ? code : 0) << (30 - 2 * (i & 15));
                m |= (uint32_t)ok << (31 - i);
            }
        }
    }
    mask[t] = m;
    if (2 * t < packed_words) packed[2 * t] = w[0];
    if (2 * t + 1 < packed_words) packed[2 * t + 1] = w[1];
    if (bad && any_invalid) atomicOr(any_invalid, 1u);
}

int phk_launch_pack(phk_ctx *ctx, const char *d_bases, uint64_t T, const char *symbols4,
                    uint32_t *d_packed, uint32_t *d_mask, uint32_t *d_any_invalid) {
    PHK_REQUIRE(d_packed && d_mask, "phk_pack: packed and mask outputs are required");
    PHK_REQUIRE(T == 0 || d_bases, "phk_pack: bases is NULL");
    PHK_REQUIRE(((uintptr_t)d_bases & 15) == 0, "phk_pack: bases must be 16-byte aligned");
    uint32_t sym = (uint32_t)(uint8_t)symbols4[0] | ((uint32_t)(uint8_t)symbols4[1] << 8) |
                   ((uint32_t)(uint8_t)symbols4[2] << 16) | ((uint32_t)(uint8_t)symbols4[3] << 24);
    uint64_t packed_words = phk_div_up(T, 16) + 1, mask_words = phk_div_up(T, 32) + 1;
    if (d_any_invalid) PHK_HIP(hipMemsetAsync(d_any_invalid, 0, sizeof(uint32_t), ctx->stream));
    uint64_t blocks = phk_div_up(mask_words, 256);
    PHK_LAUNCH(ctx, "phk_pack_kernel",
               phk_pack_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(
                   (const uint8_t *)d_bases, T, sym, d_packed, d_mask, packed_words, mask_words,
                   d_any_invalid));
    return PHK_OK;
}

// ------------------------------------------------------------------------------------
// count: one wavefront per contig, 4^K uint32 bins per wave in LDS
// ------------------------------------------------------------------------------------
// Lane l of a wave-iteration owns packed word w = w0 + l (16 window starts) and reads word
// w+1 for the K-1 bases a window may reach into; the 64-bit funnel X = w:w+1 makes window i
// the bit field X[63-2i .. 64-2K-2i], which IS the reference's bin index
// int(window, 4) (first base most significant, scripts/kmer.py:50).
template <int K, bool MASK>
__global__ __launch_bounds__(256) void phk_count_kernel(const uint32_t *__restrict__ packed,
                                                        const uint32_t *__restrict__ mask,
                                                        const uint64_t *__restrict__ offsets,
                                                        uint64_t n, uint32_t *__restrict__ counts,
                                                        uint32_t *__restrict__ nwin) {
    constexpr uint32_t D = 1u << (2 * K);
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    uint32_t *bins = lds + (size_t)wave * D;
    for (uint32_t b = lane * 4; b < D; b += 256) *reinterpret_cast<uint4 *>(bins + b) = make_uint4(0, 0, 0, 0);

    const uint64_t total_waves = (uint64_t)gridDim.x * wpb;
    for (uint64_t c = (uint64_t)blockIdx.x * wpb + wave; c < n; c += total_waves) {
        const uint64_t start = offsets[c], end = offsets[c + 1];
        uint32_t cnt = 0;
        if (end >= start + K) {
            const uint64_t last = end - K;  // last window start
            const uint64_t wb = start >> 4, we = last >> 4;
            for (uint64_t w0 = wb; w0 <= we; w0 += 64) {
                const uint64_t w = w0 + lane;
                if (w <= we) {
                    const uint32_t a = packed[w], b = packed[w + 1];
                    const uint64_t X = ((uint64_t)a << 32) | b;
                    const uint64_t g0 = w << 4;
                    // window starts i in [lo, hi] of this word belong to the contig
                    const int lo = start > g0 ? (int)(start - g0) : 0;
                    const int hi = last - g0 < 15 ? (int)(last - g0) : 15;
                    uint32_t VB = 0xFFFFFFFFu;
                    if (MASK) {
                        const uint64_t mi = w >> 1;
                        const uint64_t V = ((uint64_t)mask[mi] << 32) | mask[mi + 1];
                        VB = (uint32_t)(V >> (32 - 16 * (int)(w & 1)));
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        bool ok = (i >= lo) & (i <= hi);
                        if (MASK) ok &= ((VB << i) >> (32 - K)) == ((1u << K) - 1u);
                        const uint32_t idx = (uint32_t)(X >> (64 - 2 * K - 2 * i)) & (D - 1);
                        if (ok) {
                            atomicAdd(&bins[idx], 1u);
                            ++cnt;
                        }
                    }
                }
            }
        }
        // flush this contig's histogram (coalesced 16-byte stores) and clear the bins
        uint32_t *row = counts + c * D;
        for (uint32_t b = lane * 4; b < D; b += 256) {
            uint4 v = *reinterpret_cast<uint4 *>(bins + b);
            *reinterpret_cast<uint4 *>(row + b) = v;
            *reinterpret_cast<uint4 *>(bins + b) = make_uint4(0, 0, 0, 0);
        }
        if (nwin) {
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) cnt += __shfl_xor(cnt, s);
            if (lane == 0) nwin[c] = cnt;
        }
    }
}

template <int K>
static int launch_count_k(phk_ctx *ctx, const uint32_t *d_packed, const uint32_t *d_mask,
                          const uint64_t *d_offsets, uint64_t n, uint32_t *d_counts,
                          uint32_t *d_nwin) {
    constexpr uint32_t D = 1u << (2 * K);
    // waves per block so that a block's bins stay <= 64 KiB
    const int wpb = (D * 4u * 4u <= 65536u) ? 4 : (D * 4u * 2u <= 65536u ? 2 : 1);
    const size_t lds = (size_t)wpb * D * 4u;
    uint64_t blocks = phk_div_up(n, wpb);
    const uint64_t cap = (uint64_t)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) return PHK_OK;
    if (d_mask) {
        PHK_LAUNCH(ctx, "phk_count_kernel",
                   phk_count_kernel<K, true><<<dim3((unsigned)blocks), dim3(64 * wpb), lds, ctx->stream>>>(
                       d_packed, d_mask, d_offsets, n, d_counts, d_nwin));
    } else {
        PHK_LAUNCH(ctx, "phk_count_kernel",
                   phk_count_kernel<K, false><<<dim3((unsigned)blocks), dim3(64 * wpb), lds, ctx->stream>>>(
                       d_packed, d_mask, d_offsets, n, d_counts, d_nwin));
    }
    return PHK_OK;
}

int phk_launch_count(phk_ctx *ctx, const uint32_t *d_packed, const uint32_t *d_mask, uint64_t T,
                     const uint64_t *d_offsets, uint64_t n, int k, uint32_t *d_counts,
                     uint32_t *d_nwin) {
    (void)T;
    PHK_REQUIRE(k >= 1, "phk_count: k must be >= 1 (got %d)", k);
    if (k > PHK_MAX_K) {
        phk_set_error("phk_count: k=%d is above PHK_MAX_K=%d (4^k bins no longer fit LDS)", k, PHK_MAX_K);
        return PHK_ERR_UNSUPPORTED;
    }
    if (n == 0) return PHK_OK;
    PHK_REQUIRE(d_packed && d_offsets && d_counts, "phk_count: NULL device pointer");
    PHK_REQUIRE(((uintptr_t)d_counts & 15) == 0, "phk_count: counts must be 16-byte aligned");
    switch (k) {
        case 1: return launch_count_k<1>(ctx, d_packed, d_mask, d_offsets, n, d_counts, d_nwin);
        case 2: return launch_count_k<2>(ctx, d_packed, d_mask, d_offsets, n, d_counts, d_nwin);
        case 3: return launch_count_k<3>(ctx, d_packed, d_mask, d_offsets, n, d_counts, d_nwin);
        case 4: return launch_count_k<4>(ctx, d_packed, d_mask, d_offsets, n, d_counts, d_nwin);
        case 5: return launch_count_k<5>(ctx, d_packed, d_mask, d_offsets, n, d_counts, d_nwin);
        case 6: return launch_count_k<6>(ctx, d_packed, d_mask, d_offsets, n, d_counts, d_nwin);
        default: return launch_count_k<7>(ctx, d_packed, d_mask, d_offsets, n, d_counts, d_nwin);
    }
}

// ------------------------------------------------------------------------------------
// widen uint32 -> int64 (the reference returns NumPy's default int, scripts/kmer.py:46)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void phk_widen_kernel(const uint32_t *__restrict__ in, uint64_t count,
                                                        int64_t *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) out[i] = (int64_t)in[i];
}

int phk_launch_widen(phk_ctx *ctx, const uint32_t *d_in, uint64_t count, int64_t *d_out) {
    if (count == 0) return PHK_OK;
    uint64_t blocks = phk_div_up(count, 256);
    if (blocks > (uint64_t)ctx->num_cus * 16) blocks = (uint64_t)ctx->num_cus * 16;
    PHK_LAUNCH(ctx, "phk_widen_kernel",
               phk_widen_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(d_in, count, d_out));
    return PHK_OK;
}

// ------------------------------------------------------------------------------------
// normalise: one wavefront per row; out = (double)c / (double)rowsum  (0/0 -> NaN)
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void phk_normalize_int_kernel(const T *__restrict__ counts, uint64_t n,
                                                                uint64_t D, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t total = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n; r += total) {
        const T *row = counts + r * D;
        long long s = 0;  // exact: integer row sums of a count matrix are far below 2^63
        for (uint64_t j = lane; j < D; j += 64) s += (long long)row[j];
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh);
        const double ds = (double)s;
        for (uint64_t j = lane; j < D; j += 64) out[r * D + j] = (double)row[j] / ds;
    }
}

// float rows: the row sum follows NumPy's pairwise summation (what np.sum does on a
// contiguous float64 row: 8 running partial sums per <=128-element block, blocks combined by
// halving) so that renormalising float rows matches the reference bit for bit.
__device__ double phk_np_pairwise_sum(const double *a, uint64_t n) {
    if (n < 8) {
        double r = 0.0;  // NumPy starts from the first element; 0.0 + a0 is exact
        for (uint64_t i = 0; i < n; ++i) r += a[i];
        return r;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        uint64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        uint64_t n2 = n / 2;
        n2 -= n2 % 8;
        return phk_np_pairwise_sum(a, n2) + phk_np_pairwise_sum(a + n2, n - n2);
    }
}

__global__ __launch_bounds__(256) void phk_normalize_f64_kernel(const double *__restrict__ rows, uint64_t n,
                                                                uint64_t D, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t total = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n; r += total) {
        const double *row = rows + r * D;
        double s = 0.0;
        if (lane == 0) s = phk_np_pairwise_sum(row, D);
        s = __shfl(s, 0);
        for (uint64_t j = lane; j < D; j += 64) out[r * D + j] = row[j] / s;
    }
}

static unsigned norm_blocks(phk_ctx *ctx, uint64_t n) {
    uint64_t blocks = phk_div_up(n, 4);
    if (blocks > (uint64_t)ctx->num_cus * 8) blocks = (uint64_t)ctx->num_cus * 8;
    return (unsigned)blocks;
}

int phk_launch_normalize_u32(phk_ctx *ctx, const uint32_t *d_counts, uint64_t n, uint64_t D,
                             double *d_out) {
    if (n == 0 || D == 0) return PHK_OK;
    PHK_LAUNCH(ctx, "phk_normalize_int_kernel",
               phk_normalize_int_kernel<uint32_t><<<dim3(norm_blocks(ctx, n)), dim3(256), 0, ctx->stream>>>(
                   d_counts, n, D, d_out));
    return PHK_OK;
}

int phk_launch_normalize_i64(phk_ctx *ctx, const int64_t *d_counts, uint64_t n, uint64_t D,
                             double *d_out) {
    if (n == 0 || D == 0) return PHK_OK;
    PHK_LAUNCH(ctx, "phk_normalize_int_kernel",
               phk_normalize_int_kernel<int64_t><<<dim3(norm_blocks(ctx, n)), dim3(256), 0, ctx->stream>>>(
                   d_counts, n, D, d_out));
    return PHK_OK;
}

int phk_launch_normalize_f64(phk_ctx *ctx, const double *d_rows, uint64_t n, uint64_t D,
                             double *d_out) {
    if (n == 0 || D == 0) return PHK_OK;
    PHK_LAUNCH(ctx, "phk_normalize_f64_kernel",
               phk_normalize_f64_kernel<<<dim3(norm_blocks(ctx, n)), dim3(256), 0, ctx->stream>>>(
                   d_rows, n, D, d_out));
    return PHK_OK;
}
