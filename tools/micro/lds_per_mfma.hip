// How many LDS fragment reads (ds_read_b128, 1 KiB per wave) per MFMA can a CU sustain beside v_mfma_i32_32x32x32_i8 at
// 2 waves per SIMD?  The int8 sweep (phamers_amd/csrc/score_i8.hip) reads its operands from LDS; R = reads per 12 MFMAs:
// 13 = one column block per query block and wave (1 x 4 x 3 accumulators), 8 = 2 query blocks x 2 column blocks per wave,
// 7 = 4 x 1.  The reads are asm volatile (a plain LDS load at a loop-invariant address is hoisted out of the loop).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/lds_per_mfma.hip -o tools/micro/lds_per_mfma && tools/micro/lds_per_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int R>
__global__ __launch_bounds__(512, 1) void loop(const uint4 *in, float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint4 sm[];   // 40 KiB
    for (int i = threadIdx.x; i < 40 * 64; i += blockDim.x) sm[i] = in[i & 1023];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4 *)sm + lane * 16;
    i32x16 acc[12];
    for (int t = 0; t < 12; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    i32x4 f[4], b;
    for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const i32x4 *>(&in[(lane * 5 + i * 131) & 1023]);
    b = *reinterpret_cast<const i32x4 *>(&in[(lane * 3 + 7) & 1023]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int x = 0; x < 12; ++x) {
#pragma unroll
            for (int k = x * R / 12; k < (x + 1) * R / 12; ++k)   // this MFMA's share of the R reads
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[k & 3]) : "v"(la), "n"(((k * 3) % 40) * 1024));
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[x & 3]));
            acc[x] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[x & 3], b, acc[x], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
    float s = 0.f;
    for (int t = 0; t < 12; ++t) for (int r = 0; r < 16; ++r) s += (float)acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)f[0][0] + (float)f[1][0] + (float)f[2][0] + (float)f[3][0];
}

int main() {
    std::vector<uint32_t> rnd(4096);
    srand(5);
    for (auto &x : rnd) x = (uint32_t)rand();
    uint4 *din; float *dout;
    hipMalloc(&din, 16384); hipMalloc(&dout, 256 * 512 * 4);
    hipMemcpy(din, rnd.data(), 16384, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256;
    auto run = [&](int R, void (*kern)(const uint4 *, float *, int)) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
        kern<<<blocks, 512, 40960>>>(din, dout, 200);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0); kern<<<blocks, 512, 40960>>>(din, dout, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        const double per = best * 1e6 / (iters * 12.0 * 2);   // ns per MFMA per SIMD (2 waves per SIMD)
        printf("%2d LDS reads per 12 MFMAs: %8.3f ms  %5.1f ns per MFMA per SIMD  %6.0f TOP/s  LDS %5.1f B/ns per CU\n", R, best, per,
               (double)blocks * 8 * iters * 12 * 65536.0 / (best * 1e-3) / 1e12, 8.0 * R * 1024 / (best * 1e6 / iters));
    };
    run(0, loop<0>); run(3, loop<3>); run(6, loop<6>); run(7, loop<7>); run(8, loop<8>); run(10, loop<10>); run(12, loop<12>); run(13, loop<13>);
    run(16, loop<16>); run(24, loop<24>);
    return 0;
}
