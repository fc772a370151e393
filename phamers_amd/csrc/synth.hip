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
                                                        uint64_t packed_words, uint64_t mask_words) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
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
    uint64_t blocks = phk_div_up(threads, 256);
    PHK_LAUNCH(ctx, "phk_synth_kernel",
               phk_synth_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(
                   seed_mix, first, n, L, thresh, d_packed, d_mask, d_offsets, packed_words, mask_words));
    return PHK_OK;
}
