// probe_overlap.hip -- how much VALU work hides behind a dependent v_mfma_f32_32x32x16_f16 chain on
// gfx950?  For NV independent VALU instructions issued after every MFMA (same wave), 1 / 2 / 4 waves
// per SIMD, prints the time per MFMA.  Everything in the loop body is inline asm, so the instruction
// order is exactly what is written here.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_overlap.hip -o tools/probe_overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA_PER_ITER 48

template <int OP>
__device__ __forceinline__ void valu(float &x, float y, float z) {
    if (OP == 0) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == 1) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    if (OP == 2) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(y), "v"(z));
    if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
}

template <int NV, int OP, int NMFMA>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)(0.001f * (threadIdx.x + j));
        b[j] = (_Float16)(0.002f * (threadIdx.x - j));
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x * 8 + i);
    float y = 1.0f + threadIdx.x, z = 2.0f + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < MFMA_PER_ITER; ++s) {
            if (NMFMA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < NV; ++v) valu<OP>(x[v & 7], y, z);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int OP, int NMFMA>
static void run(float *d, int wps, const char *opname) {
    const int iters = 4000;
    const int grid = 256 * wps;  // 256 CUs x wps workgroups of 4 waves -> wps waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NV, OP, NMFMA><<<grid, 256>>>(d, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NV, OP, NMFMA><<<grid, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double steps = (double)iters * MFMA_PER_ITER;       // per wave
    const double ns_per_slot = ms * 1e6 / (steps * wps);      // SIMD time per (MFMA + NV VALU) slot
    const double tflops = NMFMA ? 32768.0 * steps * grid * 4 / (ms * 1e-3) / 1e12 : 0.0;
    printf("%-6s mfma=%d NV=%2d waves/SIMD=%d : %8.3f ms  %7.2f ns per slot per SIMD  %7.1f TFLOP/s\n", opname, NMFMA, NV, wps,
           ms, ns_per_slot, tflops);
    fflush(stdout);
}

template <int OP>
static void sweep(float *d, const char *name) {
    for (int wps = 1; wps <= 4; wps *= 2) {
        run<0, OP, 1>(d, wps, name);
        run<2, OP, 1>(d, wps, name);
        run<4, OP, 1>(d, wps, name);
        run<6, OP, 1>(d, wps, name);
        run<7, OP, 1>(d, wps, name);
        run<8, OP, 1>(d, wps, name);
        run<10, OP, 1>(d, wps, name);
        run<12, OP, 1>(d, wps, name);
        run<8, OP, 0>(d, wps, name);
    }
}

int main() {
    float *d;
    hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    sweep<0>(d, "max");
    sweep<1>(d, "med3");
    sweep<2>(d, "bfi");
    sweep<3>(d, "fma");
    return 0;
}
