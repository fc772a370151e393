// probe_loop.hip -- the count-exact proposal kernel's k-step loop in isolation (no DMA, no barrier, no
// list flush): NT=2 tiles, per k-step 4 MFMAs (2 fragments x 2 tiles) + 2 insertions (fma, and_or, 5 med3)
// + 2 ds_read_b128 of the next step's fragments.  Variants switch parts off to see what the gap costs.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_loop.hip -o tools/probe_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// HI = 1: the high-parts-only kernel's loop -- one MFMA per value (2 per k-step), one fragment read per step;
// INS = 2: insertions under the wave-uniform skip
template <int LDSR, int INS, int NW, int HI = 0>
__global__ __launch_bounds__(64 * NW, 1) void k(const float *in, float *out, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[33 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 33 * 256; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = in[i & 1023];
    __syncthreads();
    half8 bq[2][16];
    for (int t = 0; t < 2; ++t)
        for (int s = 0; s < 16; ++s)
            for (int e = 0; e < 8; ++e) bq[t][s][e] = (_Float16)(float)((lane * 7 + s * 3 + e + t) & 31);
    float lv[2][5];
    for (int t = 0; t < 2; ++t)
        for (int c = 0; c < 5; ++c) lv[t][c] = -3.0e38f;
    float negT[2] = {-5000.f, -4990.f};
    float bias[16];
    for (int r = 0; r < 16; ++r) bias[r] = in[r];
    f32x16 accA[2], accB[2];
    for (int t = 0; t < 2; ++t)
        for (int r = 0; r < 16; ++r) accB[t][r] = -3.3e38f;
    const float fbig = 3.3e38f;
    auto insert = [&](int t, float a, float b, int r) {
        const float w = fmaf(negT[t], b, a);
        const float x = __uint_as_float((__float_as_uint(w) & ~31u) | (uint32_t)(2 * r + 1));
        if (INS == 2 && __builtin_amdgcn_ballot_w64(x > lv[t][4]) == 0) return;
        const float n4 = __builtin_amdgcn_fmed3f(lv[t][3], lv[t][4], x);
        const float n3 = __builtin_amdgcn_fmed3f(lv[t][2], lv[t][3], x);
        const float n2 = __builtin_amdgcn_fmed3f(lv[t][1], lv[t][2], x);
        const float n1 = __builtin_amdgcn_fmed3f(lv[t][0], lv[t][1], x);
        lv[t][0] = __builtin_amdgcn_fmed3f(lv[t][0], x, fbig);
        lv[t][1] = n1; lv[t][2] = n2; lv[t][3] = n3; lv[t][4] = n4;
    };
    const half8 *fr = reinterpret_cast<const half8 *>(smem) + lane;
    auto block_iter = [&](f32x16 (&cur)[2], const f32x16 (&prev)[2]) {
        half8 ahn = fr[0], aln = fr[64];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const half8 ah = ahn, al = aln;
            if (LDSR && s < 15) {
                ahn = fr[(2 * s + 2) * 64];
                aln = fr[(2 * s + 3) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0) {
                f32x16 z;
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                for (int t = 0; t < 2; ++t) cur[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq[t][s], z, 0, 0, 0);
            } else {
                for (int t = 0; t < 2; ++t) cur[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq[t][s], cur[t], 0, 0, 0);
            }
            if (!HI)
                for (int t = 0; t < 2; ++t) cur[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bq[t][s], cur[t], 0, 0, 0);
            if (INS)
                for (int t = 0; t < 2; ++t) insert(t, prev[t][s], bias[s], s);
            if (INS != 2)
                for (int gi = 0; gi < (HI ? 2 : 4); ++gi) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, HI ? 8 : 4, 0);
                }
        }
    };
#pragma unroll 1
    for (int it = 0; it < iters; it += 2) {
        block_iter(accA, accB);
        block_iter(accB, accA);
    }
    float sum = 0.f;
    for (int t = 0; t < 2; ++t) {
        for (int c = 0; c < 5; ++c) sum += lv[t][c];
        for (int r = 0; r < 16; ++r) sum += accA[t][r] + accB[t][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}


// the same work on v_mfma_f32_16x16x32_f16: 4 query tiles of 16, column blocks of 16; per k-step of 32
// dimensions 8 MFMAs of 16 cycles + 2 insertions + 2 ds_read_b128
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int LDSR, int INS, int NW>
__global__ __launch_bounds__(64 * NW, 1) void k16(const float *in, float *out, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[33 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 33 * 256; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = in[i & 1023];
    __syncthreads();
    half8 bq[4][8];
    for (int t = 0; t < 4; ++t)
        for (int s = 0; s < 8; ++s)
            for (int e = 0; e < 8; ++e) bq[t][s][e] = (_Float16)(float)((lane * 7 + s * 3 + e + t) & 31);
    float lv[4][5];
    for (int t = 0; t < 4; ++t)
        for (int c = 0; c < 5; ++c) lv[t][c] = -3.0e38f;
    float negT[4] = {-5000.f, -4990.f, -4980.f, -4970.f};
    float bias[4];
    for (int r = 0; r < 4; ++r) bias[r] = in[r];
    f32x4 accA[4], accB[4];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 4; ++r) accB[t][r] = -3.3e38f;
    const float fbig = 3.3e38f;
    auto insert = [&](int t, float a, float b, int r) {
        const float w = fmaf(negT[t], b, a);
        const float x = __uint_as_float((__float_as_uint(w) & ~31u) | (uint32_t)(2 * r + 1));
        const float n4 = __builtin_amdgcn_fmed3f(lv[t][3], lv[t][4], x);
        const float n3 = __builtin_amdgcn_fmed3f(lv[t][2], lv[t][3], x);
        const float n2 = __builtin_amdgcn_fmed3f(lv[t][1], lv[t][2], x);
        const float n1 = __builtin_amdgcn_fmed3f(lv[t][0], lv[t][1], x);
        lv[t][0] = __builtin_amdgcn_fmed3f(lv[t][0], x, fbig);
        lv[t][1] = n1; lv[t][2] = n2; lv[t][3] = n3; lv[t][4] = n4;
    };
    const half8 *fr = reinterpret_cast<const half8 *>(smem) + lane;
    auto block_iter = [&](f32x4 (&cur)[4], const f32x4 (&prev)[4]) {   // 16 columns x 64 queries x 256 dims
        half8 ahn = fr[0], aln = fr[64];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const half8 ah = ahn, al = aln;
            if (LDSR && s < 7) {
                ahn = fr[(2 * s + 2) * 64];
                aln = fr[(2 * s + 3) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                for (int t = 0; t < 4; ++t) cur[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bq[t][s], z, 0, 0, 0);
            } else {
                for (int t = 0; t < 4; ++t) cur[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bq[t][s], cur[t], 0, 0, 0);
            }
            for (int t = 0; t < 4; ++t) cur[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bq[t][s], cur[t], 0, 0, 0);
            if (INS) {   // 16 values per lane and block: 2 per k-step
                insert((2 * s) >> 2, prev[(2 * s) >> 2][(2 * s) & 3], bias[(2 * s) & 3], (2 * s) & 15);
                insert((2 * s + 1) >> 2, prev[(2 * s + 1) >> 2][(2 * s + 1) & 3], bias[(2 * s + 1) & 3], (2 * s + 1) & 15);
            }
            for (int gi = 0; gi < 8; ++gi) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
        }
    };
#pragma unroll 1
    for (int it = 0; it < iters; it += 2) {
        block_iter(accA, accB);
        block_iter(accB, accA);
    }
    float sum = 0.f;
    for (int t = 0; t < 4; ++t) {
        for (int c = 0; c < 5; ++c) sum += lv[t][c];
        for (int r = 0; r < 4; ++r) sum += accA[t][r] + accB[t][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int LDSR, int INS, int NW>
static void run16(const float *din, float *d, const char *name) {
    const int iters = 1200, grid = 256;   // a 16-column block is half the work of a 32-column one
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k16<LDSR, INS, NW><<<grid, 64 * NW>>>(din, d, 20);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k16<LDSR, INS, NW><<<grid, 64 * NW>>>(din, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 64;            // 16x16x32 MFMAs per wave (16384 flop each)
    printf("16x16x32 %-20s waves/SIMD=%g : %7.3f ms  (%.0f TFLOP/s)\n", name, NW / 4.0, ms,
           16384.0 * mf * NW * grid / (ms * 1e-3) / 1e12);
}

template <int LDSR, int INS, int NW, int HI = 0>
static void run(const float *din, float *d, const char *name) {
    const int iters = 600, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<LDSR, INS, NW, HI><<<grid, 64 * NW>>>(din, d, 20);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<LDSR, INS, NW, HI><<<grid, 64 * NW>>>(din, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * (HI ? 32 : 64);            // MFMAs per wave
    const double wps = NW / 4.0;
    printf("%-28s waves/SIMD=%g : %7.3f ms  %6.2f ns per MFMA per SIMD  (%.0f TFLOP/s)\n", name, wps, ms,
           ms * 1e6 / (mf * wps), 32768.0 * mf * NW * grid / (ms * 1e-3) / 1e12);
}

int main() {
    float *din, *d;
    hipMalloc(&din, 4096 * 4); hipMalloc(&d, 256 * 512 * 4);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 0, 4>(din, d, "mfma only");
    run<1, 0, 4>(din, d, "mfma + lds reads");
    run<0, 1, 4>(din, d, "mfma + insertions");
    run<1, 1, 4>(din, d, "mfma + lds + insertions");
    run<0, 0, 8>(din, d, "mfma only");
    run<1, 0, 8>(din, d, "mfma + lds reads");
    run<0, 1, 8>(din, d, "mfma + insertions");
    run<1, 1, 8>(din, d, "mfma + lds + insertions");
    run<1, 0, 8, 1>(din, d, "HI: mfma + lds reads");
    run<1, 1, 8, 1>(din, d, "HI: + insertions");
    run<1, 2, 8, 1>(din, d, "HI: + insertions w/ skip");
    run16<0, 0, 4>(din, d, "mfma only");
    run16<1, 1, 4>(din, d, "mfma + lds + ins");
    run16<0, 0, 8>(din, d, "mfma only");
    run16<1, 1, 8>(din, d, "mfma + lds + ins");
    return 0;
}
