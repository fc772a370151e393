// probe_mfma_f16.hip -- characterise how v_mfma_f32_32x32x16_f16 accumulates its 16 products + C.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_f16.hip -o probe_mfma_f16 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A[32][16] row-major, B[16][32] row-major, C[32][32]; one wave
__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* D) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[i * 16 + 8 * h + j]; b[j] = B[(8 * h + j) * 32 + i]; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i];
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = c[r];
}

int main() {
    _Float16 hA[32 * 16], hB[16 * 32]; float hC[32 * 32], hD[32 * 32];
    _Float16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&dD, sizeof(hD));
    auto run = [&]() {
        hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
        hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dC, dD); hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    };
    auto clear = [&]() { for (auto& x : hA) x = 0; for (auto& x : hB) x = 0; for (auto& x : hC) x = 0; };
    // probe positions: element (row 0, col 0) uses A[0][k], B[k][0]
    // A: 2^24 + 15 x 1  (exact 16777231; sequential fp32 RN gives 16777216; single rounding 16777232)
    for (int pos = 0; pos < 16; pos += 5) {
        clear();
        for (int kk = 0; kk < 16; ++kk) { hA[kk] = 1; hB[kk * 32] = 1; }
        hA[pos] = 4096; hB[pos * 32] = 4096;
        run();
        printf("probeA big at k=%2d : %.1f   (exact 16777231, single-rounding 16777232, sequential 16777216)\n", pos, hD[0]);
    }
    // B: C = 2^24, products sum to 3 (16 x 3/16): exact 16777219 -> RN 16777220, trunc 16777218, sequential 16777216
    clear();
    for (int kk = 0; kk < 16; ++kk) { hA[kk] = (_Float16)0.375f; hB[kk * 32] = (_Float16)0.5f; }
    hC[0] = 16777216.f; run();
    printf("probeB C=2^24 + 16*(3/16): %.1f   (RN-of-exact 16777220, trunc 16777218, per-product 16777216)\n", hD[0]);
    // C: cancellation +2^24 - 2^24 + 1 (+ zeros)
    for (int p1 = 0; p1 < 16; p1 += 7) {
        clear();
        hA[0] = 4096; hB[0] = 4096; hA[8] = -4096; hB[8 * 32] = 4096; hA[(p1 % 14) + 1 == 8 ? 9 : (p1 % 14) + 1] = 1; hB[((p1 % 14) + 1 == 8 ? 9 : (p1 % 14) + 1) * 32] = 1;
        run();
        printf("probeC 2^24 - 2^24 + 1 (1 at k=%d): %.3f (exact 1)\n", (p1 % 14) + 1 == 8 ? 9 : (p1 % 14) + 1, hD[0]);
    }
    // D: fine-grained: 2^24 + 2^-? many small: 2^12*2^12 + 15 * (2^-6 * 2^-6 = 2^-12): exact 16777216 + 15*2^-12
    clear();
    for (int kk = 1; kk < 16; ++kk) { hA[kk] = (_Float16)0.015625f; hB[kk * 32] = (_Float16)0.015625f; }
    hA[0] = 4096; hB[0] = 4096; hC[0] = -16777216.f; run();
    printf("probeD C=-2^24 + 2^24 + 15*2^-12: %.9g (exact %.9g; wide internal accumulator keeps it)\n", hD[0], 15.0 / 4096.0);
    // E: random: error vs exact in units of u*|result| and u*sum|terms|
    srand(1); double worst_rel = 0, worst_abs = 0;
    for (int trial = 0; trial < 200; ++trial) {
        for (auto& x : hA) x = (_Float16)((rand() / (double)RAND_MAX - 0.5) * 200.0);
        for (auto& x : hB) x = (_Float16)((rand() / (double)RAND_MAX - 0.5) * 200.0);
        for (auto& x : hC) x = (float)((rand() / (double)RAND_MAX - 0.5) * 1e4);
        run();
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double ex = hC[i * 32 + j], sa = fabs((double)hC[i * 32 + j]);
            for (int kk = 0; kk < 16; ++kk) { double p = (double)hA[i * 16 + kk] * (double)hB[kk * 32 + j]; ex += p; sa += fabs(p); }
            double err = fabs((double)hD[i * 32 + j] - ex);
            const double u = 5.9604644775390625e-08;
            if (fabs(ex) > 0) worst_rel = fmax(worst_rel, err / (u * fabs(ex)));
            worst_abs = fmax(worst_abs, err / (u * sa));
        }
    }
    printf("probeE random (200 tiles): max err = %.3f u*|exact result|, %.3f u*sum|terms|  (single RN rounding gives <= 1.0 u*|result|)\n", worst_rel, worst_abs);
    // F: fp16 subnormal inputs (the lo parts of the split operands can be subnormal): kept or flushed?
    clear();
    hA[0] = (_Float16)9.5367431640625e-07f;   // 2^-20, subnormal in fp16 (min normal 2^-14)
    hB[0] = (_Float16)1024.0f;
    run();
    printf("probeF subnormal A (2^-20) x 2^10: %.9g (kept: 0.0009765625 = 2^-10; flushed: 0)\n", hD[0]);
    clear();
    hA[0] = (_Float16)1024.0f;
    hB[0] = (_Float16)5.9604644775390625e-08f;  // 2^-24, smallest fp16 subnormal
    run();
    printf("probeF subnormal B (2^-24) x 2^10: %.9g (kept: 6.10351562e-05 = 2^-14; flushed: 0)\n", hD[0]);
    return 0;
}
