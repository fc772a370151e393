// score_lists.h -- pieces shared by the proposal kernels (fp32 MFMA and split-f16 MFMA) and the
// decision stage.
#pragma once
#include "phk_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define FAST_D 256
#define CAND 4         // list depth per (query, segment, half)
#define NSEG 3         // train rows, positive centroids, negative centroids
#define PAD_V (-1.0e30f)

// sorted (descending) insert of (x, c) into a 4-deep list held in registers, tracking the largest
// value that ever fell off the list (`drop`): every column this list does not hold has a computed
// value <= drop.
__device__ __forceinline__ void list_insert(float (&v)[CAND], uint32_t (&ix)[CAND], float &drop, float x,
                                            uint32_t c) {
    const bool g0 = x > v[0], g1 = x > v[1], g2 = x > v[2], g3 = x > v[3];
    drop = fmaxf(drop, g3 ? v[3] : x);
    v[3] = g2 ? v[2] : (g3 ? x : v[3]);
    ix[3] = g2 ? ix[2] : (g3 ? c : ix[3]);
    v[2] = g1 ? v[1] : (g2 ? x : v[2]);
    ix[2] = g1 ? ix[1] : (g2 ? c : ix[2]);
    v[1] = g0 ? v[0] : (g1 ? x : v[1]);
    ix[1] = g0 ? ix[0] : (g1 ? c : ix[1]);
    v[0] = g0 ? x : v[0];
    ix[0] = g0 ? c : ix[0];
}
