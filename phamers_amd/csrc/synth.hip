// synth.hip -- device-side seeded synthetic contig batch, bit-identical to
// phamers_amd/synth.py (benchmark inputs are generated resident in HBM;
// 200 Gbases/s of packed input would exceed PCIe).
#include "phk_common.h"

__device__ __forceinline__ uint64_t phk_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// one thread per 32 stream bases (two packed words + one mask word), uniform contig length L
__global__ __launch_bounds__(256) void phk_synth_kernel(uint64_t seed_mix, uint64_t first, uint64_t n,
                                                        uint64_t L, uint32_t inv_thresh,
                                                        uint32_t *__restrict__ packed,
                                                        uint32_t *__restrict__ mask,
                                                        uint64_t *__restrict__ offsets,
                                                        uint64_t packed_words, uint64_t mask_words, uint64_t block0) {
    const uint64_t t = (block0 + blockIdx.x) * blockDim.x + threadIdx.x;
    if (t <= n) offsets[t] = t * L;
    if (t >= mask_words) return;
    const uint64_t T = n * L;
    const uint64_t g0 = t * 32;
    uint32_t w[2] = {0u, 0u};
    uint32_t m = 0u;
    uint64_t c = L ? g0 / L : 0;
    uint64_t i = L ? g0 % L : 0;
    uint64_t key = phk_splitmix64(seed_mix ^ (first + c));
    uint64_t word_j = ~0ull, word = 0;
    for (int b = 0; b < 32; ++b) {
        if (g0 + b >= T) break;
        if (i == L) {  // next contig
            i = 0;
            ++c;
            key = phk_splitmix64(seed_mix ^ (first + c));
            word_j = ~0ull;
        }
        const uint64_t j = i >> 5;
        if (j != word_j) {
            word_j = j;
            word = phk_splitmix64(key + j);
        }
        uint32_t code = (uint32_t)(word >> (62 - 2 * (i & 31))) & 3u;
        bool ok = true;
        if (inv_thresh) {
            const uint64_t h = phk_splitmix64((key ^ 0xA5A5A5A5A5A5A5A5ull) + i);
            ok = (uint32_t)(h >> 32) >= inv_thresh;
        }
        w[b >> 4] |= (ok ? code : 0u) << (30 - 2 * (b & 15));
        m |= (uint32_t)ok << (31 - b);
        ++i;
    }
    if (mask) mask[t] = m;
    if (2 * t < packed_words) packed[2 * t] = w[0];
    if (2 * t + 1 < packed_words) packed[2 * t + 1] = w[1];
}

int phk_launch_synth(phk_ctx *ctx, uint64_t seed, uint64_t first, uint64_t n, uint64_t L,
                     uint32_t invalid_ppm, uint32_t *d_packed, uint32_t *d_mask,
                     uint64_t *d_offsets) {
    PHK_REQUIRE(d_packed && d_offsets, "phk_synth: NULL output pointer");
    PHK_REQUIRE(invalid_ppm == 0 || d_mask, "phk_synth: invalid_ppm > 0 needs a mask buffer");
    PHK_REQUIRE(invalid_ppm <= 1000000u, "phk_synth: invalid_ppm out of range");
    const uint64_t T = n * L;
    const uint64_t packed_words = phk_div_up(T, 16) + 1, mask_words = phk_div_up(T, 32) + 1;
    // host-side splitmix64(seed): same arithmetic as the device function
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    const uint64_t seed_mix = z ^ (z >> 31);
    const uint32_t thresh = (uint32_t)(((uint64_t)invalid_ppm << 32) / 1000000ull);
    uint64_t threads = mask_words > n + 1 ? mask_words : n + 1;
    PHK_LAUNCH_SLICED(ctx, "phk_synth_kernel", phk_div_up(threads, 256), b0, nblk,
                      phk_synth_kernel<<<dim3(nblk), dim3(256), 0, ctx->stream>>>(
                          seed_mix, first, n, L, thresh, d_packed, d_mask, d_offsets, packed_words, mask_words, b0));
    return PHK_OK;
}

// ------------------------------------------------------------------------------------
// Ragged, composition-skewed batch (the second, clearly labelled bench workload: what real assemblies look like).
// Contig boundaries come from the caller (offsets[n+1], any lengths); contig c has its own GC fraction
//     gc(c) = 1/2 + spread * (u(c) - 1/2),  u(c) = bits 63..48 of key(c) / 65536,
// and base i of contig c is drawn with ONE hash per base:  h = splitmix64(key(c) + GOLD2 + i);
//     G/C if (h >> 32) < gc(c) * 2^32 else A/T, the low bit of h picks within the pair  (codes A0 T1 G2 C3).
// Invalid bases as in the uniform generator.  phamers_amd/synth.py restates this bit for bit.
// ------------------------------------------------------------------------------------
#define PHK_SYNTH_GOLD2 0xD1B54A32D192ED03ull

__global__ __launch_bounds__(256) void phk_synth_ragged_kernel(uint64_t seed_mix, uint64_t first, uint64_t n,
                                                               const uint64_t *__restrict__ offsets, uint32_t spread_permille,
                                                               uint32_t inv_thresh, uint32_t *__restrict__ packed,
                                                               uint32_t *__restrict__ mask, uint64_t packed_words,
                                                               uint64_t mask_words, uint64_t block0) {
    const uint64_t t = (block0 + blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= mask_words) return;
    const uint64_t T = offsets[n];
    const uint64_t g0 = t * 32;
    uint32_t w[2] = {0u, 0u};
    uint32_t m = 0u;
    if (g0 < T) {
        // contig of base g0: last c with offsets[c] <= g0
        uint64_t lo = 0, hi = n;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (offsets[mid] <= g0) lo = mid; else hi = mid;
        }
        uint64_t c = lo, end = offsets[c + 1];
        uint64_t key = 0;
        uint32_t gc_thr = 0;
        bool fresh = true;
        for (int b = 0; b < 32; ++b) {
            const uint64_t g = g0 + b;
            if (g >= T) break;
            while (g >= end) {   // next non-empty contig
                ++c;
                end = offsets[c + 1];
                fresh = true;
            }
            if (fresh) {
                fresh = false;
                key = phk_splitmix64(seed_mix ^ (first + c));
                const long long dev = ((long long)(key >> 48) - 32768) * (long long)spread_permille * 65536 / 1000;
                gc_thr = (uint32_t)(2147483648ll + dev);
            }
            const uint64_t i = g - offsets[c];
            const uint64_t h = phk_splitmix64(key + PHK_SYNTH_GOLD2 + i);
            const uint32_t code = ((uint32_t)(h >> 32) < gc_thr ? 2u : 0u) | (uint32_t)(h & 1u);
            bool ok = true;
            if (inv_thresh) {
                const uint64_t hv = phk_splitmix64((key ^ 0xA5A5A5A5A5A5A5A5ull) + i);
                ok = (uint32_t)(hv >> 32) >= inv_thresh;
            }
            w[b >> 4] |= (ok ? code : 0u) << (30 - 2 * (b & 15));
            m |= (uint32_t)ok << (31 - b);
        }
    }
    if (mask) mask[t] = m;
    if (2 * t < packed_words) packed[2 * t] = w[0];
    if (2 * t + 1 < packed_words) packed[2 * t + 1] = w[1];
}

int phk_launch_synth_ragged(phk_ctx *ctx, uint64_t seed, uint64_t first, uint64_t n, const uint64_t *d_offsets,
                            uint64_t total_bases, uint32_t gc_spread_permille, uint32_t invalid_ppm, uint32_t *d_packed,
                            uint32_t *d_mask) {
    PHK_REQUIRE(d_packed && d_offsets, "phk_synth_ragged: NULL pointer");
    PHK_REQUIRE(invalid_ppm == 0 || d_mask, "phk_synth_ragged: invalid_ppm > 0 needs a mask buffer");
    PHK_REQUIRE(invalid_ppm <= 1000000u && gc_spread_permille <= 1000u, "phk_synth_ragged: parameter out of range");
    const uint64_t packed_words = phk_div_up(total_bases, 16) + 1, mask_words = phk_div_up(total_bases, 32) + 1;
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    const uint64_t seed_mix = z ^ (z >> 31);
    const uint32_t thresh = (uint32_t)(((uint64_t)invalid_ppm << 32) / 1000000ull);
    PHK_LAUNCH_SLICED(ctx, "phk_synth_ragged_kernel", phk_div_up(mask_words, 256), b0, nblk,
                      phk_synth_ragged_kernel<<<dim3(nblk), dim3(256), 0, ctx->stream>>>(
                          seed_mix, first, n, d_offsets, gc_spread_permille, thresh, d_packed, d_mask, packed_words, mask_words, b0));
    return PHK_OK;
}
