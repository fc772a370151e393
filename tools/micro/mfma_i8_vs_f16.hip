// mfma_i8_vs_f16.hip -- (1) operand lane map of v_mfma_i32_32x32x32_i8 checked with exact integer data; (2) sustained rate
// of bare MFMA loops on the whole chip, random operands: f16 32x32x16 vs i8 32x32x32, operands in registers and re-read
// from LDS.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_i8_vs_f16.hip -o tools/micro/mfma_i8_vs_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// one wave: D = A[32][32] x B[32][32] (int8), lane l: row / col r = l & 31, k = 16 (l >> 5) + j
__global__ void layout_check(const int8_t *A, const int8_t *B, int *D) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    i32x4 a, b;
    int8_t *ap = reinterpret_cast<int8_t *>(&a), *bp = reinterpret_cast<int8_t *>(&b);
    for (int j = 0; j < 16; ++j) { ap[j] = A[r * 32 + 16 * h + j]; bp[j] = B[(16 * h + j) * 32 + r]; }
    i32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0;
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}

template <int KIND, int LDSR>   // KIND 0: f16 32x32x16, 1: i8 32x32x32
__global__ __launch_bounds__(512, 1) void loop(const uint4 *in, float *out, int iters) {
    __shared__ __attribute__((aligned(16))) uint4 sm[16 * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 64; i += blockDim.x) sm[i] = in[i];
    __syncthreads();
    uint4 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(lane * 5 + i * 131) & 1023]; b[i] = in[(lane * 3 + i * 17 + 7) & 1023]; }
    f32x16 accf[4]; i32x16 acci[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) { accf[t][r] = 0.f; acci[t][r] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            uint4 av = a[s & 3];
            if (LDSR) av = sm[s * 64 + lane];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (KIND == 0) accf[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8 *>(&av), *reinterpret_cast<half8 *>(&b[t]), accf[t], 0, 0, 0);
                else acci[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<i32x4 *>(&av), *reinterpret_cast<i32x4 *>(&b[t]), acci[t], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += accf[t][r] + (float)acci[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    // ---- layout ----
    std::vector<int8_t> A(1024), B(1024); std::vector<int> D(1024), W(1024);
    srand(3);
    for (auto &x : A) x = (int8_t)(rand() % 255 - 127);
    for (auto &x : B) x = (int8_t)(rand() % 255 - 127);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int k = 0; k < 32; ++k) s += (int)A[i * 32 + k] * (int)B[k * 32 + j]; W[i * 32 + j] = s; }
    int8_t *dA, *dB; int *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    layout_check<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += D[i] != W[i];
    printf("i8 32x32x32 lane map (row/col = lane & 31, k = 16 (lane >> 5) + j; C/D as the f32 forms): %d of 1024 elements wrong (exact integer data, asymmetric operands)\n", bad);
    // ---- rate ----
    std::vector<uint32_t> rnd(4096);
    for (auto &x : rnd) { // random halves in a moderate range / random int8
        _Float16 h0 = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 8.f), h1 = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 8.f);
        x = (uint32_t)(*reinterpret_cast<uint16_t *>(&h0)) | ((uint32_t)(*reinterpret_cast<uint16_t *>(&h1)) << 16);
    }
    uint4 *din; float *dout;
    hipMalloc(&din, 16384); hipMalloc(&dout, 256 * 4 * 512 * 4);
    hipMemcpy(din, rnd.data(), 16384, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 256;
    auto run = [&](const char *name, void (*kern)(const uint4 *, float *, int), double flop_per_mfma) {
        kern<<<blocks, 512>>>(din, dout, 200);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0); kern<<<blocks, 512>>>(din, dout, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        const double mfmas = (double)blocks * 8 * iters * 64;
        printf("%-44s %8.3f ms  %7.1f TOP/s  (%.1f ns per MFMA per SIMD)\n", name, best, mfmas * flop_per_mfma / (best * 1e-3) / 1e12,
               best * 1e6 / (iters * 64.0 * 2));
    };
    run("f16 32x32x16, operands in registers", loop<0, 0>, 2.0 * 32 * 32 * 16);
    run("i8  32x32x32, operands in registers", loop<1, 0>, 2.0 * 32 * 32 * 32);
    run("f16 32x32x16, A re-read from LDS", loop<0, 1>, 2.0 * 32 * 32 * 16);
    run("i8  32x32x32, A re-read from LDS", loop<1, 1>, 2.0 * 32 * 32 * 32);
    return 0;
}
