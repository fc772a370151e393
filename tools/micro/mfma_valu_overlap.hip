// mfma_valu_overlap.hip -- do a wave's vector instructions hide under the matrix pipe?  Whole chip, one 8-wave workgroup per
// CU (two waves per SIMD, the sweeps' occupancy): per iteration 4 independent v_mfma_i32_32x32x32_i8 (or f16 32x32x16) on 4
// accumulators, each followed by V vector instructions that touch none of the MFMA registers (two independent chains of
// v_med3_f32 / v_fma_f32).  Prints ns per MFMA and SIMD for V = 0 .. 16.  If the two overlap the time stays at the bare rate
// until V instructions need longer than an MFMA; if they do not, it grows from V = 1.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o tools/micro/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int V, int DEP>   // KIND 0: f16, 1: i8; DEP 1: the V instructions form ONE dependent chain, 0: two chains
__global__ __launch_bounds__(512, 1) void loop(const uint4 *in, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    uint4 a[4], b;
    for (int i = 0; i < 4; ++i) a[i] = in[(lane * 5 + i * 131) & 1023];
    b = in[(lane * 3 + 7) & 1023];
    f32x16 accf[4]; i32x16 acci[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) { accf[t][r] = 0.f; acci[t][r] = 0; }
    float x0 = __uint_as_float(in[lane].x & 0x3FFFFFFFu), x1 = __uint_as_float(in[lane].y & 0x3FFFFFFFu), y0 = 1.0f, y1 = 2.0f, z = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (KIND == 0) accf[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8 *>(&a[t]), *reinterpret_cast<half8 *>(&b), accf[t], 0, 0, 0);
            else acci[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<i32x4 *>(&a[t]), *reinterpret_cast<i32x4 *>(&b), acci[t], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                if (DEP || (v & 1) == 0) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(y0), "v"(z));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(y1), "v"(z));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = x0 + x1;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += KIND == 0 ? accf[t][r] : (float)acci[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int V, int DEP>
static void run(const uint4 *d_in, float *d_out, int cus) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    loop<KIND, V, DEP><<<cus, 512>>>(d_in, d_out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    loop<KIND, V, DEP><<<cus, 512>>>(d_in, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2 waves x 4 MFMAs per iteration
    printf("%s V=%2d %s: %.2f ns per MFMA and SIMD, %.2f ns per (MFMA + V vector instructions) and wave\n", KIND ? "i8 32x32x32 " : "f16 32x32x16", V,
           DEP ? "one chain " : "two chains", ms * 1e6 / ((double)iters * 8), ms * 1e6 / ((double)iters * 4));
}


// The sweep epilogue's instruction sequence for one value behind every MFMA (MODE 0), with the wave-uniform branch around it
// that `if (do_epi)` compiles to (MODE 1), and two values behind every MFMA (MODE 2): integer combine, conversion, two
// multiply-adds, index bits, five in-place v_med3 on a list kept across the loop.
template <int MODE>
__global__ __launch_bounds__(512, 1) void loop_values(const uint4 *in, float *out, int iters, int flag) {
    const int lane = threadIdx.x & 63;
    uint4 a[4], b;
    for (int i = 0; i < 4; ++i) a[i] = in[(lane * 5 + i * 131) & 1023];
    b = in[(lane * 3 + 7) & 1023];
    i32x16 acci[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acci[t][r] = 0;
    float l0 = -1e30f, l1 = -2e30f, l2 = -3e30f, l3 = -4e30f, l4 = -5e30f;
    const float fbig = 3.3e38f, negT = -(float)(in[lane].z & 1023u), g = __uint_as_float(in[lane].x & 0x3FFFFFFFu), bb = __uint_as_float(in[lane].y & 0x3FFFFFFFu);
    int sh = (int)(in[lane].w & 255u), sm = (int)(in[lane].z & 255u);
    const bool do_epi = flag != 0;   // (wave-uniform, unknown at compile time)
    auto value = [&](int r) {
        int c;
        asm volatile("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(c) : "v"(sh), "v"(sm));
        float f, t, x;
        asm volatile("v_cvt_f32_i32_e32 %0, %1" : "=v"(f) : "v"(c));
        asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(t) : "v"(bb), "v"(negT));
        asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(t) : "v"(f), "v"(g));
        asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(x) : "v"(t), "s"(0xFFFFFFE0u), "v"(2 * r + 1));
        asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(l4) : "v"(l3), "v"(x));
        asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(l3) : "v"(l2), "v"(x));
        asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(l2) : "v"(l1), "v"(x));
        asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(l1) : "v"(l0), "v"(x));
        asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(l0) : "v"(x), "v"(fbig));
        sh += 3;
    };
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acci[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<i32x4 *>(&a[t]), *reinterpret_cast<i32x4 *>(&b), acci[t], 0, 0, 0);
            if (MODE == 0) value(t);
            if (MODE == 1) { if (do_epi) value(t); }
            if (MODE == 2) { value(t); value(t + 4); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = l0 + l1 + l2 + l3 + l4;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += (float)acci[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run_values(const uint4 *d_in, float *d_out, int cus) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    loop_values<MODE><<<cus, 512>>>(d_in, d_out, 200, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    loop_values<MODE><<<cus, 512>>>(d_in, d_out, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("i8 32x32x32  %s: %.2f ns per MFMA and SIMD\n", MODE == 0 ? "one list value (10 instructions) behind every MFMA" : MODE == 1 ? "the same inside a wave-uniform branch" : "two values behind every MFMA", ms * 1e6 / ((double)iters * 8));
}

// How many independent accumulators does a wave need?  NACC accumulators per wave, MFMAs issued round-robin (each depends on
// the one NACC back), no other instructions.  phk_knn_f16h_kernel has 2 per wave, the int8 sweep 12.
template <int KIND, int NACC>
__global__ __launch_bounds__(512, 1) void loop_chains(const uint4 *in, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    uint4 a[4], b;
    for (int i = 0; i < 4; ++i) a[i] = in[(lane * 5 + i * 131) & 1023];
    b = in[(lane * 3 + 7) & 1023];
    f32x16 accf[NACC]; i32x16 acci[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) { accf[t][r] = 0.f; acci[t][r] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (KIND == 0) accf[t % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8 *>(&a[t & 3]), *reinterpret_cast<half8 *>(&b), accf[t % NACC], 0, 0, 0);
            else acci[t % NACC] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<i32x4 *>(&a[t & 3]), *reinterpret_cast<i32x4 *>(&b), acci[t % NACC], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += KIND == 0 ? accf[t][r] : (float)acci[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NACC>
static void run_chains(const uint4 *d_in, float *d_out, int cus, int threads) {
    const int iters = 10000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    loop_chains<KIND, NACC><<<cus, threads>>>(d_in, d_out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    loop_chains<KIND, NACC><<<cus, threads>>>(d_in, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const int waves_per_simd = threads / 256;
    printf("%s %d accumulator(s) per wave, %d wave(s) per SIMD: %.2f ns per MFMA and SIMD\n", KIND ? "i8 32x32x32 " : "f16 32x32x16", NACC, waves_per_simd,
           ms * 1e6 / ((double)iters * 8 * waves_per_simd));
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    std::vector<uint4> h(1024);
    srand(5);
    for (auto &v : h) v = make_uint4(rand(), rand(), rand(), rand());
    uint4 *d_in; float *d_out;
    hipMalloc(&d_in, h.size() * sizeof(uint4));
    hipMalloc(&d_out, (size_t)cus * 512 * sizeof(float));
    hipMemcpy(d_in, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
    printf("%d CUs, one 512-thread workgroup each (2 waves per SIMD)\n", cus);
    run<1, 0, 0>(d_in, d_out, cus); run<1, 2, 0>(d_in, d_out, cus); run<1, 4, 0>(d_in, d_out, cus); run<1, 6, 0>(d_in, d_out, cus);
    run<1, 8, 0>(d_in, d_out, cus); run<1, 10, 0>(d_in, d_out, cus); run<1, 12, 0>(d_in, d_out, cus); run<1, 16, 0>(d_in, d_out, cus);
    run<1, 4, 1>(d_in, d_out, cus); run<1, 8, 1>(d_in, d_out, cus); run<1, 12, 1>(d_in, d_out, cus);
    run<0, 0, 0>(d_in, d_out, cus); run<0, 4, 0>(d_in, d_out, cus); run<0, 8, 0>(d_in, d_out, cus); run<0, 12, 0>(d_in, d_out, cus);
    run_chains<0, 1>(d_in, d_out, cus, 512); run_chains<0, 2>(d_in, d_out, cus, 512); run_chains<0, 4>(d_in, d_out, cus, 512); run_chains<0, 8>(d_in, d_out, cus, 512);
    run_chains<0, 2>(d_in, d_out, cus, 256); run_chains<0, 4>(d_in, d_out, cus, 256);
    run_chains<1, 2>(d_in, d_out, cus, 512); run_chains<1, 4>(d_in, d_out, cus, 512);
    run_values<0>(d_in, d_out, cus); run_values<1>(d_in, d_out, cus); run_values<2>(d_in, d_out, cus);
    return 0;
}
