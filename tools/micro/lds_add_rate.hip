// Micro-benchmark: the rate at which a CU retires conflict-free wave-wide ds_add_u32 (no return), the operation the
// k-mer slot kernel is built on.  Each lane adds into its own 4-byte column (bank = lane & 31) at 16 row addresses it
// keeps in registers, so the loop has no address arithmetic: what is measured is the LDS atomic path alone.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_add_rate.hip -o gpurun_out/lds_add_rate && gpurun_out/lds_add_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int VALU_PER_ADD, int COLS>
__global__ __launch_bounds__(512) void k_adds(uint32_t *out, int iters) {
    extern __shared__ uint32_t lds[];   // [256][COLS]: 32 KiB (two lanes of a wave per column) or 64 KiB (one)
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    for (int i = threadIdx.x; i < 256 * COLS; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const uint32_t colb = (threadIdx.x & (COLS - 1)) * 4u;
    constexpr uint32_t RMASK = COLS == 32 ? 0x7F80u : 0xFF00u;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x;
    uint32_t a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        x = x * 1664525u + 1013904223u;
        a[j] = ((x >> 16) & RMASK) | colb;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint32_t ad = a[j];
            if (VALU_PER_ADD >= 1) ad = (ad & 0xFFFFu) | colb;          // filler vector operations on the address
            if (VALU_PER_ADD >= 2) asm volatile("v_lshrrev_b32 %0, 0, %0" : "+v"(ad));
            __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)ad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" : "+v"(a[j]));
        }
    }
    __syncthreads();
    uint32_t s = 0;
    for (int i = threadIdx.x; i < 256 * COLS; i += blockDim.x) s += lds[i];
    if (s == 0xFFFFFFFFu) out[0] = s;
}

template <int V, int COLS>
static void run(const char *name, int cus, int wg_per_cu) {
    uint32_t *d;
    hipMalloc(&d, 4);
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)k_adds<V, COLS>, hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * COLS);
    k_adds<V, COLS><<<cus * wg_per_cu, 512, 1024 * COLS>>>(d, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_adds<V, COLS><<<cus * wg_per_cu, 512, 1024 * COLS>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double adds = (double)cus * wg_per_cu * 8 * iters * 16;   // wave instructions
    printf("%-28s %d WG/CU: %.3f ms, %.1f G wave-adds/s, %.2f ns per add per CU\n", name, wg_per_cu, ms, adds / ms * 1e-6,
           ms * 1e6 / (adds / cus));
    hipFree(d);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, %d MHz\n", p.name, cus, p.clockRate / 1000);
    for (int w = 1; w <= 3; ++w) run<0, 32>("32 columns, ds_add only", cus, w);
    for (int w = 1; w <= 3; ++w) run<1, 32>("32 columns, + 1 vector op", cus, w);
    for (int w = 1; w <= 3; ++w) run<2, 32>("32 columns, + 2 vector ops", cus, w);
    for (int w = 1; w <= 2; ++w) run<0, 64>("64 columns, ds_add only", cus, w);
    for (int w = 1; w <= 2; ++w) run<2, 64>("64 columns, + 2 vector ops", cus, w);
    return 0;
}
