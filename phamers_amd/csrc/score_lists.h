// score_lists.h -- pieces shared by the proposal kernels (fp32 MFMA and split-f16 MFMA) and the
// decision stage.
#pragma once
#include "phk_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define FAST_D 256
#define CAND 4         // list depth per (query, segment, half)
#define NSEG 3         // train rows, positive centroids, negative centroids
#define PAD_V (-1.0e30f)

// Count rows enter the count-exact MFMA kernels CENTRED by an integer: c_i - c0 with c0 = the integer nearest to T / D
// (T = row sum).  The operand stays exact in fp16 (|c_i - c0| <= 2048), and the accumulator of the MFMA chain carries
// sum_i (c_i - c0) r_ji instead of sum_i c_i r_ji -- for a contig of unremarkable composition a quarter of the magnitude,
// which is what the chain's rounding and alignment errors scale with (see ErrBound in score_mfma.hip).  The column side
// absorbs the shift: its bias terms are built with mu - 1/D in place of mu (pack_segment_f16), which leaves
// (c0 - T/D) * sum_i r_ji, bounded by the model's hsum term.  Every kernel derives c0 from T with this one function.
// (phk_row_center: phk_common.h -- the count kernel's flush prepares the int8 operand with it too)

// Rounding model of v_mfma_f32_32x32x16_f16 (tools/diag/mfma_emulate.py, tests/test_gpu_score.py): the instruction works
// in two halves of 8 products; in a half every term -- the products and the running sum -- is cut (toward zero) to a
// multiple of 2^(E - 25), E = the exponent of the largest of |running sum| and 2 |product|, the cut terms are added
// exactly and the result is rounded to float32 (nearest even).  A term is aligned by its operands' exponent FIELDS: a
// non-zero float16 subnormal counts as 2^-14 whatever its value (round 4: tests/mfma_fuzz_worker.py found the charge
// stated on the products' values violated 150-fold by subnormal operands; tools/diag/mfma_emulate.py has the probe).
// With A >= every |running sum| of the chain and p >= every NOMINAL |product| (operands below 2^-14 taken as 2^-14),
// one instruction errs by at most
//     2 halves x [ 9 terms x 2^-25 max(A, 2p)  +  2^-24 A ]  <=  u (11 A + 18 p),   u = 2^-24.
#define PHK_MFMA_ACC 11.0    // per instruction, on the largest running sum
#define PHK_MFMA_PROD 18.0   // per instruction, on the largest single product

// A word every wave of a (large) grid reads -- a device-side list length written by the kernel before -- through the
// scalar cache.  As a vector load it is one and the same L2 line requested by every wave of the grid: with 250 k waves
// that line's channel became a queue every other load of the kernel stood in (40 k cycles for the first two loads of
// phk_rerank16_kernel, by its phase timers; profiles/r02/README.md).  The pointer must be wave-uniform.
__device__ __forceinline__ uint32_t phk_uniform_load(const uint32_t *p) {
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// Candidate lists of a batch of N queries, structure of arrays (a wave of the proposal kernels writes, and
// a lane-per-query reader reads, consecutive queries at consecutive addresses):
//   value / column index of slot c of half-list h of segment seg of query q : [((seg*2 + h)*CAND + c) * N + q]
//   best value the half-list ever dropped                                    : [(seg*2 + h) * N + q]
__device__ __forceinline__ uint64_t cand_at(int seg, int h, int c, uint64_t q, uint64_t N) {
    return (uint64_t)((seg * 2 + h) * CAND + c) * N + q;
}
__device__ __forceinline__ uint64_t candu_at(int seg, int h, uint64_t q, uint64_t N) { return (uint64_t)(seg * 2 + h) * N + q; }
__device__ __forceinline__ void cand_store(float *cv, uint32_t *ci, float *cu, int seg, int h, uint64_t q, uint64_t N,
                                           float v0, float v1, float v2, float v3, uint32_t i0, uint32_t i1, uint32_t i2,
                                           uint32_t i3, float u) {
    cv[cand_at(seg, h, 0, q, N)] = v0;
    cv[cand_at(seg, h, 1, q, N)] = v1;
    cv[cand_at(seg, h, 2, q, N)] = v2;
    cv[cand_at(seg, h, 3, q, N)] = v3;
    ci[cand_at(seg, h, 0, q, N)] = i0;
    ci[cand_at(seg, h, 1, q, N)] = i1;
    ci[cand_at(seg, h, 2, q, N)] = i2;
    ci[cand_at(seg, h, 3, q, N)] = i3;
    cu[candu_at(seg, h, q, N)] = u;
}
__device__ __forceinline__ void cand_store_empty(float *cv, uint32_t *ci, float *cu, int seg, int h, uint64_t q, uint64_t N) {
    cand_store(cv, ci, cu, seg, h, q, N, -3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu,
               0xFFFFFFFFu, -3.0e38f);
}

// v_bfi_b32 proper (the C form above is turned into compare + select pairs by hipcc)
__device__ __forceinline__ uint32_t phk_bfi_hw(uint32_t m, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}

// Sorted (descending) insert of (x, c) into a 4-deep list held in registers, tracking the largest
// value that ever fell off the list (`drop`): every column this list does not hold has a computed
// value <= drop.  Written without any i1 condition on purpose: hipcc turns a chain of `?:` selects
// into divergent branches (s_and_saveexec + register moves, ~10x the instructions).  Values move with
// v_max / v_med3 (the list is sorted, so the new k-th value is med3(v[k-1], v[k], x)); indices move with
// bitfield inserts under sign-replicated masks m_k = (x > v[k]) ? ~0 : 0 taken from the sign of v[k] - x.
__device__ __forceinline__ uint32_t phk_bfi(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }

__device__ __forceinline__ void list_insert(float (&v)[CAND], uint32_t (&ix)[CAND], float &drop, float x,
                                            uint32_t c) {
    const uint32_t m0 = (uint32_t)(__float_as_int(v[0] - x) >> 31);
    const uint32_t m1 = (uint32_t)(__float_as_int(v[1] - x) >> 31);
    const uint32_t m2 = (uint32_t)(__float_as_int(v[2] - x) >> 31);
    const uint32_t m3 = (uint32_t)(__float_as_int(v[3] - x) >> 31);
    drop = fmaxf(drop, fminf(v[3], x));
    ix[3] = phk_bfi(m2, ix[2], phk_bfi(m3, c, ix[3]));
    ix[2] = phk_bfi(m1, ix[1], phk_bfi(m2, c, ix[2]));
    ix[1] = phk_bfi(m0, ix[0], phk_bfi(m1, c, ix[1]));
    ix[0] = phk_bfi(m0, c, ix[0]);
    const float n3 = __builtin_amdgcn_fmed3f(v[2], v[3], x);
    const float n2 = __builtin_amdgcn_fmed3f(v[1], v[2], x);
    const float n1 = __builtin_amdgcn_fmed3f(v[0], v[1], x);
    v[0] = fmaxf(v[0], x);
    v[1] = n1;
    v[2] = n2;
    v[3] = n3;
}
// The same, skipped by the whole wave when no lane can place the value: x <= drop (<= v[3]) leaves list and drop as they
// are.  A list keeps the 4 best (+ the best dropped) of the n values it has seen, so late in a sweep almost every value
// is skipped; the branch is wave-uniform.
__device__ __forceinline__ void list_insert_needed(float (&v)[CAND], uint32_t (&ix)[CAND], float &drop, float x, uint32_t c) {
    if (__builtin_amdgcn_ballot_w64(x > drop) != 0) list_insert(v, ix, drop, x, c);
}
