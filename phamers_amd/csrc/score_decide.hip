// score_decide.hip -- the decision stage's lane-per-query kernels: what the margin test certifies is decided with one lane
// per query (three phases: lists -> row norms and exact centroid distances with 16 lanes per query -> tests and vote); the
// rest is handed, listed, to the wave-level kernels of score_rerank.hip.
//   phk_decide_kernel      D = 256, count-exact or split-query lists
//   phk_decide_gen_kernel  general D, the two-part int8 sweep's lists
//   phk_decide_h_kernel    D = 256, lists of HIGH-PART values (the default first pass at k = 4)
// Shared device code: score_decide.h.
#include "score_decide.h"

// ------------------------------------------------------------------------------------
// 2c. D = 256: the decision for the (large) majority of queries that the margin test certifies, with the
//     per-query scalar arithmetic -- ranking of the 8 candidates of a segment, error bounds, square roots,
//     tanh -- done by ONE lane per query instead of redundantly by the 16 lanes that share a query in
//     phk_rerank16_kernel (that kernel is VALU-bound on exactly this redundancy: ~345 instructions per query).
//     A 256-thread block handles 256 queries in three phases:
//       A  lane = query : read its six half-lists, merge them, fetch column norms / labels of the leaders
//       B  16 lanes = query, 16 queries at a time: the row itself -- |q|^2, |q'|^2 and the exact float64
//          distances to the leading positive / negative centroid (exact_d2_g16)
//       C  lane = query : margin tests (the conditions of resolve_segment_g16), vote, proximity metric
//     A query some segment of which is not certified by margin goes to `slow_list`; phk_rerank16_kernel
//     (LISTED) then treats it exactly as before (exact candidate distances, fallback queue).
// ------------------------------------------------------------------------------------
template <int SRC>
__global__ __launch_bounds__(256) void phk_decide_kernel(const void *__restrict__ src, RerankParams p) {
    __shared__ uint32_t s_ix[2][256];
    __shared__ double s_T[256], s_nq2[256], s_nqp2[256], s_dp2[256], s_dn2[256], s_opQ[256], s_opI[256], s_opH[256];
    const int tid = threadIdx.x, lane = tid & 63, t = lane & 15;
    const int QB = blockDim.x;   // queries per block (64: one wave per block, no cross-wave waiting at the phase changes)
    const uint64_t qb = (uint64_t)blockIdx.x * QB;
    const bool want_knn = (p.method & PHK_METHOD_KNN) != 0, want_cen = (p.method & PHK_METHOD_KMEANS) != 0;

    // the count rows of phase B are fetched two passes ahead (explicit register double buffer: hipcc does not
    // software-pipeline that loop by itself and each pass would otherwise start with a full memory round trip);
    // the first two are requested here, ahead of phase A's list reads
    auto rowptr = [&](int pass) {
        const int ql = pass * (QB / 16) + (tid >> 4);
        const uint64_t q = qb + ql < p.N ? qb + ql : p.N - 1;
        return reinterpret_cast<const uint2 *>(static_cast<const uint32_t *>(src) + q * FAST_D) + t;
    };
    uint2 pre[2][8];
    if (SRC == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            pre[0][i] = rowptr(0)[16 * i];
            pre[1][i] = rowptr(1)[16 * i];
        }
    }

    // ---- phase A: one lane per query ----
    const uint64_t qa = qb + tid;
    const bool in_a = qa < p.N;
    const uint64_t qc = in_a ? qa : p.N - 1;
    float gap_hi[NSEG], gap_lo[NSEG], Useg[NSEG];   // need-th and (need+1)-th computed value, best dropped value
    uint32_t lead[NSEG][3];
    bool filled[NSEG];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        const uint32_t ncols = sg == 0 ? (uint32_t)p.M : sg == 1 ? (uint32_t)p.n_cpos : (uint32_t)p.n_cneg;
        const int need = sg == 0 ? p.kn : 1;
        Useg[sg] = fmaxf(p.cand_u[candu_at(sg, 0, qc, p.N)], p.cand_u[candu_at(sg, 1, qc, p.N)]);
        // the 4 best of the 8 candidates by insertion (descending; an equal value stays behind: a tie at the
        // decisive position fails the margin test anyway); padding / empty slots never enter
        float v[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        uint32_t ix[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        float w[8];
        uint32_t wx[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {   // consecutive lanes = consecutive queries: coalesced
            w[c] = p.cand_v[cand_at(sg, c >> 2, c & 3, qc, p.N)];
            wx[c] = p.cand_i[cand_at(sg, c >> 2, c & 3, qc, p.N)];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float x = wx[c] >= ncols ? -3.0e38f : w[c];
            uint32_t xi = wx[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool up = x > v[k];
                const float tv = v[k];
                const uint32_t ti = ix[k];
                v[k] = up ? x : tv;
                ix[k] = up ? xi : ti;
                x = up ? tv : x;
                xi = up ? ti : xi;
            }
        }
        gap_hi[sg] = need == 1 ? v[0] : need == 2 ? v[1] : v[2];
        gap_lo[sg] = need == 1 ? v[1] : need == 2 ? v[2] : v[3];
        filled[sg] = (need == 1 ? ix[0] : need == 2 ? ix[1] : ix[2]) < ncols;
#pragma unroll
        for (int r = 0; r < 3; ++r) lead[sg][r] = ix[r];
    }
    s_ix[0][tid] = lead[1][0] < (uint32_t)p.n_cpos ? lead[1][0] : 0u;
    s_ix[1][tid] = lead[2][0] < (uint32_t)p.n_cneg ? lead[2][0] : 0u;
    // speculative gathers, consumed in phase C
    double cn0[3];
    uint8_t lab0[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint32_t c = lead[0][r] < (uint32_t)p.M ? lead[0][r] : 0u;
        cn0[r] = p.colnorm[c];
        lab0[r] = p.labels[c];
    }
    const double cnp = p.colnorm[p.M + s_ix[0][tid]], cnn = p.colnorm[p.M + p.n_cpos + s_ix[1][tid]];
    __syncthreads();

    // ---- phase B: 16 lanes per query, 16 queries per pass ----
    {
        double mu[16];
        const double2 *mp = reinterpret_cast<const double2 *>(p.mu64) + t;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double2 m2 = mp[16 * i];
            mu[2 * i] = m2.x;
            mu[2 * i + 1] = m2.y;
        }
#pragma unroll 2
        for (int pass = 0; pass < 16; ++pass) {
            const int ql = pass * (QB / 16) + (tid >> 4);
            const uint64_t q = qb + ql < p.N ? qb + ql : p.N - 1;
            double qd[16], Tq = 1.0, invT2 = 1.0;
            bool bad = false;
            uint32_t cmx = 0, cmn = 0xFFFFFFFFu;
            if (SRC == 0) {
                uint2 cur[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) cur[i] = pre[pass & 1][i];
                if (pass + 2 < 16) {
                    const uint2 *nrow = rowptr(pass + 2);
#pragma unroll
                    for (int i = 0; i < 8; ++i) pre[pass & 1][i] = nrow[16 * i];
                }
                uint32_t sum = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint2 c = cur[i];
                    sum += c.x + c.y;
                    qd[2 * i + 0] = (double)c.x;
                    qd[2 * i + 1] = (double)c.y;
                    cmx = max(cmx, max(c.x, c.y));
                    cmn = min(cmn, min(c.x, c.y));
                }
                sum = group16_sum(sum);
                cmx = group16_max(cmx);
                cmn = group16_min(cmn);
                bad = sum == 0;
                Tq = (double)sum;
                invT2 = 1.0 / (Tq * Tq);
            } else {
                const double2 *row = reinterpret_cast<const double2 *>(static_cast<const double *>(src) + q * FAST_D) + t;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double2 v2 = row[16 * i];
                    qd[2 * i] = v2.x;
                    qd[2 * i + 1] = v2.y;
                    bad |= v2.x != v2.x || v2.y != v2.y;
                }
                unsigned bb = bad ? 1u : 0u;
                bb |= __shfl_xor(bb, 8); bb |= __shfl_xor(bb, 4); bb |= __shfl_xor(bb, 2); bb |= __shfl_xor(bb, 1);
                bad = bb != 0;
            }
            double aq = 0.0, ap = 0.0, am = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const double c0 = fma(-Tq, mu[i], qd[i]);
                aq = fma(qd[i], qd[i], aq);
                ap = fma(c0, c0, ap);
                am = fmax(am, fabs(c0));
            }
            const double sumsq = group16_sum(aq);
            const double nq2 = sumsq * invT2, nqp2 = group16_sum(ap) * invT2;
            // the query operand of the proposal's MFMA chain (see ErrBound): c - c0 for count-exact lists, q' otherwise
            double opQ = 0.0, opI = group16_max(am) / Tq, opH = 0.0;
            if (SRC == 0 && p.per_row_scale && !bad) {
                const CenteredOperand cop = phk_centered_operand(sumsq, Tq, (double)cmx, (double)cmn, (double)FAST_D, p.eb_hsum);
                opQ = cop.Q; opI = cop.I; opH = cop.habs;
            }
            double dp2 = 0.0, dn2 = 0.0;
            if (want_cen) {
                exact_d2_pair_g16(qd, Tq, invT2, p.C64 + (uint64_t)s_ix[0][ql] * FAST_D,
                                  p.C64 + (p.n_cpos + (uint64_t)s_ix[1][ql]) * FAST_D, t, dp2, dn2);
            }
            if (t == 0) {
                s_T[ql] = bad ? 0.0 : Tq;   // 0 marks a NaN row
                s_opQ[ql] = opQ;
                s_opI[ql] = opI;
                s_opH[ql] = opH;
                s_nq2[ql] = nq2;
                s_nqp2[ql] = nqp2;
                s_dp2[ql] = dp2;
                s_dn2[ql] = dn2;
            }
        }
    }
    __syncthreads();

    // ---- phase C: one lane per query ----
    if (!in_a) return;
    const double Tq = s_T[tid];
    if (Tq == 0.0) {  // zero-count contig / NaN input: the reference's normalised row is NaN
        p.scores[p.q_base + qa] = __builtin_nan("");
        if (p.status) atomicAdd(p.status, 1u);
        return;
    }
    const double nq2 = s_nq2[tid], nqp2 = s_nqp2[tid];
    const double vs = p.per_row_scale ? p.vscale / Tq : p.vscale;
    ErrBound eb;
    eb.A = sqrt(nq2) + p.mu_norm;
    eb.P = sqrt(nqp2);
    eb.Q = s_opQ[tid]; eb.I = s_opI[tid]; eb.habs = s_opH[tid];
    eb.cA = p.eb_cA; eb.cP = p.eb_cP; eb.cR = p.eb_cR; eb.cabs = p.eb_abs; eb.cQ = p.eb_cQ; eb.cI = p.eb_cI; eb.cIf = p.eb_cIf;
    const double nqp = eb.P;
    const double eps_g = eb(p.rmax);
    auto certify = [&](int sg, int need, const double *cnorms) {   // resolve_segment_g16's margin test
        if (!filled[sg]) return false;
        const double d2up = fmax(nqp2 - 2.0 * ((double)gap_hi[sg] * vs - eps_g), 0.0);
        const double R0 = fmin(p.rmax, (nqp + sqrt(d2up)) * (1.0 + 1e-6));
        bool near = true;
        for (int r = 0; r < need; ++r) near = near && cnorms[r] <= R0;
        const double eps_m = near ? eb(R0) : eps_g;
        return ((double)gap_hi[sg] - (double)gap_lo[sg]) * vs > 2.0 * eps_m;
    };
    bool cert = true;
    double knn = 0.0, cen = 0.0;
    if (want_knn) {
        cert = certify(0, p.kn, cn0);
        int votes = 0;
        for (int r = 0; r < p.kn; ++r) votes += lab0[r] ? 1 : 0;
        knn = (2 * votes > p.kn) ? 1.0 : -1.0;
    }
    if (want_cen) {
        cert = cert && certify(1, 1, &cnp) && certify(2, 1, &cnn);
        const double ep = sqrt(s_dp2[tid]), en = sqrt(s_dn2[tid]);
        cen = tanh((en - ep) / (ep + en));  // scripts/phamer.py:206-209
    }
    if (cert) {
        p.scores[p.q_base + qa] = knn + cen;  // scripts/phamer.py:313
    } else {
        p.slow_list[atomicAdd(p.fb_count + 2, 1u)] = (uint32_t)qa;
    }
}

// ------------------------------------------------------------------------------------
// 2c'. The same idea at general D for the lists of the two-part int8 sweep (round 4).  phk_rerank_kernel spends one wave per
//     query, and most of that wave's ~650 instructions are per-query scalar work that all 64 lanes repeat (ranking, the
//     error bound, the window and margin tests); a third of the step at configs[2].  Here a 64-thread block takes 64
//     queries through the k = 4 kernel's three phases:
//       A  lane = query : its six half-lists, the 4 best of each segment's 8 candidates, labels of the leaders
//       B  16 lanes = query, 4 queries per pass: the count row in 1024-dimension chunks (G16 ownership: every load
//          covers one contiguous 256 B piece) -- sum of squares, |q'|^2 against the training mean in LDS, and the exact
//          distances to the leading positive / negative centroid in the canonical form of exact_d2
//       C  lane = query : e_l, eps, the two-part window; a query is decided HERE when, in every segment the method uses,
//          the window holds exactly `need` columns and ends above everything the half-lists dropped -- the case
//          rerank_one_query decides "as it stands" (96 % of configs[2]); the margin test is implied:
//          gap_lo vs < thr = gap_hi vs - 2 (e_l + eps_g)  =>  (gap_hi - gap_lo) vs > 2 eps.
//     Everything else -- wider windows (they need the L product), windows past the lists, rows beyond the int8 operand,
//     NaN rows excepted -- goes to slow_list, and phk_rerank_kernel (listed) treats those queries exactly as before.
//     The row sum comes from p.rowsum (the count kernel's / the launcher's), so a chunk's c - T mu needs no second pass.
// ------------------------------------------------------------------------------------
template <int DSUB>
__global__ __launch_bounds__(64, 3) void phk_decide_gen_kernel(const uint32_t *__restrict__ counts, RerankParams p) {
    constexpr int D = 256 * DSUB;
    constexpr int CHUNK = D < 1024 ? D : 1024;     // dimensions per chunk of phase B
    constexpr int NCH = D / CHUNK;
    constexpr int LPC = CHUNK / 64;                // uint4 loads per lane and chunk (4 dimensions each)
    static_assert(LPC % 4 == 0, "loads in groups of four");
    __shared__ double s_mu[D];
    __shared__ uint32_t s_ix[2][64];
    __shared__ double s_ssq[64], s_nqp2[64], s_dp2[64], s_dn2[64];
    __shared__ float s_gap[NSEG][3][64];   // phase A -> C: need-th / (need+1)-th list value, best dropped value (not kept in registers across phase B)
    __shared__ uint32_t s_flag[64];        // ... bits 0-2: segment filled, bits 4-6: labels of the three leading train columns
    const int tid = threadIdx.x, t = tid & 15, grp = tid >> 4;
    const uint64_t qb = (uint64_t)blockIdx.x * 64;
    const bool want_knn = (p.method & PHK_METHOD_KNN) != 0, want_cen = (p.method & PHK_METHOD_KMEANS) != 0;
    for (int i = tid; i < D / 2; i += 64) reinterpret_cast<double2 *>(s_mu)[i] = reinterpret_cast<const double2 *>(p.mu64)[i];

    // ---- phase A: one lane per query ----
    const uint64_t qa = qb + tid;
    const bool in_a = qa < p.N;
    const uint64_t qc = in_a ? qa : p.N - 1;
    float gap_hi[NSEG], gap_lo[NSEG], Useg[NSEG];   // need-th and (need+1)-th list value, best dropped value
    uint32_t lead[NSEG][3];
    bool filled[NSEG];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        const uint32_t ncols = sg == 0 ? (uint32_t)p.M : sg == 1 ? (uint32_t)p.n_cpos : (uint32_t)p.n_cneg;
        const int need = sg == 0 ? p.kn : 1;
        Useg[sg] = fmaxf(p.cand_u[candu_at(sg, 0, qc, p.N)], p.cand_u[candu_at(sg, 1, qc, p.N)]);
        float v[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        uint32_t ix[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        float w[8];
        uint32_t wx[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {   // consecutive lanes = consecutive queries: coalesced
            w[c] = p.cand_v[cand_at(sg, c >> 2, c & 3, qc, p.N)];
            wx[c] = p.cand_i[cand_at(sg, c >> 2, c & 3, qc, p.N)];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {   // the 4 best by insertion (descending; an equal value stays behind)
            float x = wx[c] >= ncols ? -3.0e38f : w[c];
            uint32_t xi = wx[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool up = x > v[k];
                const float tv = v[k];
                const uint32_t ti = ix[k];
                v[k] = up ? x : tv;
                ix[k] = up ? xi : ti;
                x = up ? tv : x;
                xi = up ? ti : xi;
            }
        }
        gap_hi[sg] = need == 1 ? v[0] : need == 2 ? v[1] : v[2];
        gap_lo[sg] = need == 1 ? v[1] : need == 2 ? v[2] : v[3];
        filled[sg] = (need == 1 ? ix[0] : need == 2 ? ix[1] : ix[2]) < ncols;
#pragma unroll
        for (int r = 0; r < 3; ++r) lead[sg][r] = ix[r];
    }
    s_ix[0][tid] = lead[1][0] < (uint32_t)p.n_cpos ? lead[1][0] : 0u;
    s_ix[1][tid] = lead[2][0] < (uint32_t)p.n_cneg ? lead[2][0] : 0u;
    {
        uint32_t fl = 0;
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
            s_gap[sg][0][tid] = gap_hi[sg];
            s_gap[sg][1][tid] = gap_lo[sg];
            s_gap[sg][2][tid] = Useg[sg];
            fl |= filled[sg] ? 1u << sg : 0u;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) fl |= p.labels[lead[0][r] < (uint32_t)p.M ? lead[0][r] : 0u] ? 16u << r : 0u;
        s_flag[tid] = fl;
    }
    __syncthreads();

    // ---- phase B: 16 lanes per query, 4 queries per pass ----
#pragma unroll 1
    for (int pass = 0; pass < 16; ++pass) {
        const int ql = pass * 4 + grp;
        const uint64_t q = qb + ql < p.N ? qb + ql : p.N - 1;
        const double Tq = (double)p.rowsum[q];
        const uint4 *row = reinterpret_cast<const uint4 *>(counts + q * D) + t;
        const double *cp = p.C64 + (uint64_t)s_ix[0][ql] * D, *cn = p.C64 + (p.n_cpos + (uint64_t)s_ix[1][ql]) * D;
        // The exact distances are formed EXACTLY as exact_d2 forms them in a 64-lane wave, so that a score does not depend on
        // which kernel decided it: that wave's lane 16 g + t owns dimensions 256 sub + 64 g + 4 t .. + 3 -- here load i of a
        // chunk, g = i & 3, sub = 4 ch + (i >> 2) -- and accumulates them over sub with the same nesting; its reduction is
        // the row sums of the four lane groups (the same DPP rotations over the same 16 positions), then (G0 + G1) + (G2 + G3).
        double ssq = 0.0, ap = 0.0, dp[4] = {0.0, 0.0, 0.0, 0.0}, dn[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll 1
            for (int io = 0; io < LPC / 4; ++io) {   // four loads at a time (one per lane group of the canonical form): ~80 registers in flight
            uint4 c[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = row[ch * (CHUNK / 4) + 16 * (4 * io + i)];   // dimensions CHUNK ch + 64 (4 io + i) + 4 t .. + 3
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d0 = ch * CHUNK + 64 * (4 * io + i) + 4 * t;
                const double x[4] = {(double)c[i].x, (double)c[i].y, (double)c[i].z, (double)c[i].w};
                const double2 m0 = *reinterpret_cast<const double2 *>(s_mu + d0), m1 = *reinterpret_cast<const double2 *>(s_mu + d0 + 2);
                const double mu4[4] = {m0.x, m0.y, m1.x, m1.y};
                double2 a0 = {0.0, 0.0}, a1 = {0.0, 0.0}, b0 = {0.0, 0.0}, b1 = {0.0, 0.0};
                if (want_cen) {
                    a0 = *reinterpret_cast<const double2 *>(cp + d0); a1 = *reinterpret_cast<const double2 *>(cp + d0 + 2);
                    b0 = *reinterpret_cast<const double2 *>(cn + d0); b1 = *reinterpret_cast<const double2 *>(cn + d0 + 2);
                }
                const double ca[4] = {a0.x, a0.y, a1.x, a1.y}, cb[4] = {b0.x, b0.y, b1.x, b1.y};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ssq = fma(x[e], x[e], ssq);
                    const double qm = fma(-Tq, mu4[e], x[e]);
                    ap = fma(qm, qm, ap);
                }
                if (want_cen) {
                    const double e0 = fma(-Tq, ca[0], x[0]), e1 = fma(-Tq, ca[1], x[1]), e2 = fma(-Tq, ca[2], x[2]), e3 = fma(-Tq, ca[3], x[3]);
                    dp[i & 3] = fma(e0, e0, fma(e1, e1, fma(e2, e2, fma(e3, e3, dp[i & 3]))));
                    const double f0 = fma(-Tq, cb[0], x[0]), f1 = fma(-Tq, cb[1], x[1]), f2 = fma(-Tq, cb[2], x[2]), f3 = fma(-Tq, cb[3], x[3]);
                    dn[i & 3] = fma(f0, f0, fma(f1, f1, fma(f2, f2, fma(f3, f3, dn[i & 3]))));
                }
            }
            }
        }
        ssq = group16_sum(ssq);
        ap = group16_sum(ap);
        double dps = 0.0, dns = 0.0;
        if (want_cen) {
            dps = (group16_sum(dp[0]) + group16_sum(dp[1])) + (group16_sum(dp[2]) + group16_sum(dp[3]));
            dns = (group16_sum(dn[0]) + group16_sum(dn[1])) + (group16_sum(dn[2]) + group16_sum(dn[3]));
        }
        if (t == 0) {
            s_ssq[ql] = ssq;
            s_nqp2[ql] = ap;      // (x T^2)
            s_dp2[ql] = dps;      // (x T^2)
            s_dn2[ql] = dns;
        }
    }
    __syncthreads();

    // ---- phase C: one lane per query ----
    bool slow = false;
    if (in_a) {
        const uint32_t Tu = p.rowsum[qa];
        const uint32_t fl = s_flag[tid];
        if (Tu == 0u) {  // zero-count contig: the reference's normalised row is NaN
            p.scores[p.q_base + qa] = __builtin_nan("");
            if (p.status) atomicAdd(p.status, 1u);
        } else {
            const double Tq = (double)Tu, ry = 1.0 / Tq, invT2 = 1.0 / (Tq * Tq);
            const double ssq = s_ssq[tid], nqp2 = s_nqp2[tid] * invT2;
            const double vs = p.vscale * ry;
            const double rcen = (double)phk_row_center(Tu, D);
            const CenteredOperand cop = phk_centered_operand_fast(ssq, Tq, ry, rcen, rcen, (double)D, p.eb_hsum);
            ErrBound eb;
            eb.A = phk_sqrt_up(ssq * (ry * ry) * (1.0 + 1e-12)) + p.mu_norm;
            eb.P = phk_sqrt_up(nqp2 + cop.shift2);
            eb.Q = cop.Q; eb.I = cop.I; eb.habs = cop.habs;
            eb.cA = p.eb_cA; eb.cP = p.eb_cP; eb.cR = p.eb_cR; eb.cabs = p.eb_abs; eb.cQ = p.eb_cQ; eb.cI = p.eb_cI; eb.cIf = p.eb_cIf;
            const double eps_g = eb(p.rmax);
            // a segment is decided as it stands: the window [h_need - 2 (e_l + eps), ..] holds exactly `need` list values
            // and ends above everything the half-lists dropped
            auto as_it_stands = [&](int sg) {
                if (!((fl >> sg) & 1u)) return false;
                const double el = cop.Q * p.lam8[sg] * (1.0 + 1.0e-6);
                const double thr = (double)s_gap[sg][0][tid] * vs - 2.0 * (el + eps_g);
                return (double)s_gap[sg][2][tid] * vs < thr && (double)s_gap[sg][1][tid] * vs < thr;
            };
            bool ok = true;
            double knn = 0.0;
            if (want_knn) {
                ok = as_it_stands(0);
                int votes = 0;
                for (int r = 0; r < p.kn; ++r) votes += (fl >> (4 + r)) & 1u;
                knn = (2 * votes > p.kn) ? 1.0 : -1.0;
            }
            if (want_cen) ok = ok && as_it_stands(1) && as_it_stands(2);
            if (ok) {
                if (want_cen) {
                    const uint64_t oq = p.out_map ? (uint64_t)p.out_map[qa] : qa;
                    if (p.pend) {   // (phk_finish_cen_kernel turns them into the proximity metric)
                        p.pend[2 * oq] = s_dp2[tid] * invT2;
                        p.pend[2 * oq + 1] = s_dn2[tid] * invT2;
                        p.scores[p.q_base + oq] = knn;
                    } else {
                        const double ep = sqrt(s_dp2[tid] * invT2), en = sqrt(s_dn2[tid] * invT2);
                        p.scores[p.q_base + oq] = knn + tanh((en - ep) / (ep + en));  // scripts/phamer.py:206-209, 313
                    }
                } else {
                    p.scores[p.q_base + (p.out_map ? (uint64_t)p.out_map[qa] : qa)] = knn;
                }
            } else {
                slow = true;
            }
        }
    }
    // hand-over: one atomic per wave, not per query (same-line atomics retire one after the other)
    const unsigned long long sm = __ballot(slow);
    if (sm) {
        uint32_t base = 0;
        if (tid == 0) base = atomicAdd(p.fb_count + 2, (uint32_t)__popcll(sm));
        base = __shfl(base, 0);
        if (slow) p.slow_list[base + (uint32_t)__popcll(sm & ((1ull << tid) - 1ull))] = (uint32_t)qa;
    }
}

// ------------------------------------------------------------------------------------
// 2d. decision stage of the high-parts-only proposal (phk_knn_f16h_kernel; D = 256, uint32 counts).
//     The lists hold HIGH-PART values  w^h_j  whose distance from the count-exact value  w_j  is the low product
//         w_j - w^h_j = sum_i (c_i - T mu_i) lo_ji,      |.| <= T S |q'| lam_j,   lam_j = |lo_j| / S   (Cauchy-Schwarz).
//     With  e_h = |q'| lam*(R0) + e22  (lam* = the largest lam_j among the columns within reach R0, HiParams.lam_tab;
//     e22 = the count-exact error model, which also covers this pass's fewer MFMA roundings):
//       window   every column whose true value can be among the `need` best has  w^h >= h_need - 2 e_h,  h_need = the
//                need-th best high-part value.  Columns in the window must all be list members: the best value either
//                half-list dropped has to be below the window, else the query takes the second chance.
//       refine   the window's members (3 to 8 columns, typically 3 or 4) get the low product -- float32 v_fma_mix on the
//                float64-centred counts, 16 lanes per query, from the low parts stored in G16 order (512 B per column);
//                its 20 roundings are bounded inside the test (2^-19 |q'| lam*).  They then carry count-exact
//                values and the count-exact margin test decides their order exactly as phk_decide_kernel does.
//       centroid segments (need = 1): the leader is certified by its high-part margin (h_1 - h_2 > 2 e_h; both are list
//                members: each half-list keeps its 4 best) and its exact float64 distance is computed as before; a
//                leader that is not certified sends the query to the second chance.
//     Phases as in phk_decide_kernel: A one lane per query (lists, window), B 16 lanes per query (row norms, low
//     products, exact centroid distances), C one lane per query (margin tests, vote, metric).
// ------------------------------------------------------------------------------------
template <bool KNN, bool CEN>
__global__ __launch_bounds__(64) void phk_decide_h_kernel(const uint32_t *__restrict__ counts, RerankParams p, HiParams hp) {
    __shared__ uint32_t s_c0[8][64];        // train-segment candidates by descending high-part value
    __shared__ uint32_t s_ix[2][64];        // centroid-segment leaders
    __shared__ float s_corr[PHK_HI_REFINE][64];
    // phase A's per-query results wait in LDS while phase B (the register-hungry part) runs
    __shared__ float s_v8[8][64], s_u0[64], s_ch[4][64];
    __shared__ double s_cn[5][64];
    __shared__ uint32_t s_flags[64];
    // phase B leaves the row's RAW sums here; everything one lane per query can finish -- the division by T^2, the centred
    // operand's norms (a float64 square root and three divisions) -- is phase C's: in phase B the 16 lanes of a query, 64 lanes of
    // a wave, each repeated it in every pass (a fifth of that phase's instructions, and the kernel is bound by its own
    // instruction stream at two waves per SIMD)
    __shared__ double s_T[64], s_sumsq[64], s_apsum[64], s_dp2[64], s_dn2[64];
    __shared__ uint32_t s_cmx[64], s_cmn[64];
    __shared__ double s_mu[FAST_D];         // the training mean (LDS reads keep vmcnt for the row / column loads)
    const int tid = threadIdx.x, t = tid & 15;
    const uint64_t qb = (uint64_t)blockIdx.x * 64;
    constexpr bool want_knn = KNN, want_cen = CEN;
    // Every load of this kernel is issued in batches that do not depend on each other, with a scheduling barrier between
    // a batch and its first use: left alone, the compiler trades loads in flight for registers and emits load, wait,
    // use, load, wait, ... (24 exposed round trips in phase A and 16 per pass in phase B, by the ISA), and a load under
    // a run-time condition makes every later wait conservative (vmcnt completes in order) -- hence the template
    // parameters instead of `method` tests.
    auto rowptr = [&](int pass) {
        const int ql = pass * 4 + (tid >> 4);
        const uint64_t q = qb + ql < p.N ? qb + ql : p.N - 1;
        return reinterpret_cast<const uint2 *>(counts + q * FAST_D) + t;
    };
    uint2 cur[8];   // the count row of the pass at hand; the next one is requested a whole pass ahead
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = rowptr(0)[16 * i];
    {
        const double2 m2a = reinterpret_cast<const double2 *>(p.mu64)[2 * tid], m2b = reinterpret_cast<const double2 *>(p.mu64)[2 * tid + 1];
        reinterpret_cast<double2 *>(s_mu)[2 * tid] = m2a;
        reinterpret_cast<double2 *>(s_mu)[2 * tid + 1] = m2b;
    }

    // ---- phase A: one lane per query ----
    const uint64_t qa = qb + tid;
    const bool in_a = qa < p.N;
    const uint64_t qc = in_a ? qa : p.N - 1;
    float v8[8];
    uint32_t i8[8];
    float U0 = 0.f;
    bool ok0 = true;          // the need-th list position holds a real column
    float ch1[2] = {0.f, 0.f}, ch2[2] = {0.f, 0.f};
    uint32_t cl[2] = {0u, 0u};
    bool cfill[2] = {true, true};
    {
        float lv[NSEG][8], lu[2] = {0.f, 0.f};
        uint32_t li[NSEG][8];
        if (KNN) {
            lu[0] = p.cand_u[candu_at(0, 0, qc, p.N)];
            lu[1] = p.cand_u[candu_at(0, 1, qc, p.N)];
        }
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg)
            if (sg == 0 ? KNN : CEN) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    li[sg][c] = p.cand_i[cand_at(sg, c >> 2, c & 3, qc, p.N)];
                    lv[sg][c] = p.cand_v[cand_at(sg, c >> 2, c & 3, qc, p.N)];
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        // train segment: the 8 candidates sorted by high-part value (descending; empty / padding slots last)
        if (KNN) {
            U0 = fmaxf(lu[0], lu[1]);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                v8[c] = li[0][c] >= (uint32_t)p.M ? -3.0e38f : lv[0][c];
                i8[c] = li[0][c];
            }
#pragma unroll
            for (int a = 1; a < 8; ++a)          // insertion sort network, fully unrolled (descending)
#pragma unroll
                for (int b = a; b > 0; --b) {
                    const bool sw = v8[b] > v8[b - 1];
                    const float tv = v8[b]; const uint32_t ti = i8[b];
                    v8[b] = sw ? v8[b - 1] : v8[b]; i8[b] = sw ? i8[b - 1] : i8[b];
                    v8[b - 1] = sw ? tv : v8[b - 1]; i8[b - 1] = sw ? ti : i8[b - 1];
                }
        }
        // centroid segments: leader and runner-up by high-part value
        if (CEN) {
#pragma unroll
            for (int sg = 1; sg <= 2; ++sg) {
                const uint32_t ncols = sg == 1 ? (uint32_t)p.n_cpos : (uint32_t)p.n_cneg;
                float b1 = -3.0e38f, b2 = -3.0e38f;
                uint32_t bi = 0xFFFFFFFFu;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t ix = li[sg][c];
                    const float w = ix >= ncols ? -3.0e38f : lv[sg][c];
                    const bool up = w > b1;
                    b2 = up ? b1 : fmaxf(b2, w);
                    bi = up ? ix : bi;
                    b1 = up ? w : b1;
                }
                ch1[sg - 1] = b1; ch2[sg - 1] = b2; cl[sg - 1] = bi;
                cfill[sg - 1] = bi < ncols;
            }
        }
    }
    const uint32_t ixp = cl[0] < (uint32_t)p.n_cpos ? cl[0] : 0u, ixn = cl[1] < (uint32_t)p.n_cneg ? cl[1] : 0u;
    s_ix[0][tid] = ixp;
    s_ix[1][tid] = ixn;
    {
        // gathers consumed in phase C: one batch
        double cn0[3];
        uint8_t lab[8];
        uint32_t c8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) c8[r] = (KNN && i8[r] < (uint32_t)p.M) ? i8[r] : 0u;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < 3) cn0[r] = p.colnorm[c8[r]];
            lab[r] = p.labels[c8[r]];
        }
        const double cnp = p.colnorm[p.M + ixp], cnn = p.colnorm[p.M + p.n_cpos + ixn];
        __builtin_amdgcn_sched_barrier(0);
        uint32_t labbits = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            labbits |= (uint32_t)(lab[r] ? 1u : 0u) << r;
            s_c0[r][tid] = c8[r];
            s_v8[r][tid] = (KNN && i8[r] < (uint32_t)p.M) ? v8[r] : -3.0e38f;
        }
        if (KNN) ok0 = i8[p.kn - 1] < (uint32_t)p.M;
        s_u0[tid] = U0;
        s_ch[0][tid] = ch1[0]; s_ch[1][tid] = ch2[0]; s_ch[2][tid] = ch1[1]; s_ch[3][tid] = ch2[1];
        s_cn[0][tid] = cn0[0]; s_cn[1][tid] = cn0[1]; s_cn[2][tid] = cn0[2]; s_cn[3][tid] = cnp; s_cn[4][tid] = cnn;
        s_flags[tid] = labbits | (ok0 ? 0x100u : 0u) | (cfill[0] ? 0x200u : 0u) | (cfill[1] ? 0x400u : 0u);
    }
    __syncthreads();

    // ---- phase B: 16 lanes per query, 4 queries per pass ----
    // A pass issues all its column loads -- the two centroid rows (float64) and the low parts of the first
    // PHK_HI_REFINE train candidates, whose addresses depend on the lists only -- and, youngest, the count row of the
    // NEXT pass; then it reduces its own row (in registers since the pass before).  One exposed L2 round trip per pass,
    // and no wait ever covers the count row's HBM miss.  The low products are fetched for all PHK_HI_REFINE candidates
    // (a window-sized fetch, tried, has to wait for the row first and cost 60 % more time); phase C, which knows the
    // window, only uses the members.
#pragma unroll 2
    for (int pass = 0; pass < 16; ++pass) {
        const int ql = pass * 4 + (tid >> 4);
        double2 ca[8], cb[8];
        uint4 l0[PHK_HI_REFINE], l1[PHK_HI_REFINE];
        uint2 nxt[8];
        if (CEN) {
            const double2 *ra = reinterpret_cast<const double2 *>(p.C64 + (uint64_t)s_ix[0][ql] * FAST_D) + t;
            const double2 *rb = reinterpret_cast<const double2 *>(p.C64 + (p.n_cpos + (uint64_t)s_ix[1][ql]) * FAST_D) + t;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                ca[i] = ra[16 * i];
                cb[i] = rb[16 * i];
            }
        }
        if (KNN) {
#pragma unroll
            for (int r = 0; r < PHK_HI_REFINE; ++r) {
                const uint4 *lp = reinterpret_cast<const uint4 *>(hp.lo16 + (uint64_t)s_c0[r][ql] * FAST_D) + t;   // G16-ordered rows
                l0[r] = lp[0];
                l1[r] = lp[16];
            }
        }
        {
            const uint2 *nrow = rowptr((pass + 1) & 15);   // (the last pass re-reads row 0: a load under a condition would
#pragma unroll                                             //  make the waits below conservative)
            for (int i = 0; i < 8; ++i) nxt[i] = nrow[16 * i];
        }
        __builtin_amdgcn_sched_barrier(0);
        double qd[16];
        uint32_t sum = 0, cmx = 0, cmn = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint2 c = cur[i];
            sum += c.x + c.y;
            cmx = max(cmx, max(c.x, c.y));
            cmn = min(cmn, min(c.x, c.y));
            qd[2 * i + 0] = (double)c.x;
            qd[2 * i + 1] = (double)c.y;
        }
        sum = group16_sum(sum);
        cmx = group16_max(cmx);
        cmn = group16_min(cmn);
        const bool bad = sum == 0;
        const double Tq = (double)sum;
        double dp2 = 0.0, dn2 = 0.0;   // (the sums; phase C multiplies by 1 / T^2)
        if (CEN) {   // exact float64 distances to the two leading centroids (as exact_d2_g16)
            double acca = 0.0, accb = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double a0 = fma(-Tq, ca[i].x, qd[2 * i]), a1 = fma(-Tq, ca[i].y, qd[2 * i + 1]);
                const double b0 = fma(-Tq, cb[i].x, qd[2 * i]), b1 = fma(-Tq, cb[i].y, qd[2 * i + 1]);
                acca = fma(a0, a0, fma(a1, a1, acca));
                accb = fma(b0, b0, fma(b1, b1, accb));
            }
            dp2 = group16_sum(acca);
            dn2 = group16_sum(accb);
        }
        double aq = 0.0, ap = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {   // qd becomes c - T mu (the centred counts) in place
            const double2 m2 = reinterpret_cast<const double2 *>(s_mu)[16 * i + t];
            aq = fma(qd[2 * i], qd[2 * i], fma(qd[2 * i + 1], qd[2 * i + 1], aq));
            qd[2 * i] = fma(-Tq, m2.x, qd[2 * i]);
            qd[2 * i + 1] = fma(-Tq, m2.y, qd[2 * i + 1]);
            ap = fma(qd[2 * i], qd[2 * i], fma(qd[2 * i + 1], qd[2 * i + 1], ap));
        }
        const double sumsq = group16_sum(aq), apsum = group16_sum(ap);
        // The proposal kernel's query operand is c - c0 (phk_row_center), so its value is the high product of the
        // UNcentred counts minus (c0 - T/D) sum_i hi_ji; the low product that completes it is therefore taken with
        // c - T mu - (c0 - T/D): sum_i (c_i - T mu_i - dlt) lo_ji = sum_i (c_i - T mu_i) lo_ji + dlt sum_i hi_ji - dlt sum_i r~'_ji,
        // the last term being the model's hsum residue (see ErrBound).
        if (KNN) {
            // float32 products (v_fma_mix takes the half operand as it is): 16 + 4 roundings per sum, bounded in
            // phase C by 2^-19 |x| lam* -- 1e-6 of the low product's own bound
            const double dlt = (double)phk_row_center(sum, FAST_D) - Tq * (1.0 / FAST_D);
            float qf[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) qf[i] = (float)(qd[i] - dlt);
#pragma unroll
            for (int r = 0; r < PHK_HI_REFINE; ++r) {
                const _Float16 *lh0 = reinterpret_cast<const _Float16 *>(&l0[r]), *lh1 = reinterpret_cast<const _Float16 *>(&l1[r]);
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc = fmaf(qf[i], (float)lh0[i], acc);
                    acc = fmaf(qf[8 + i], (float)lh1[i], acc);
                }
                acc = group16_sum(acc);
                if (t == 0) s_corr[r][ql] = acc;
            }
        }
        if (t == 0) {
            s_T[ql] = bad ? 0.0 : Tq;   // 0 marks a NaN row
            s_sumsq[ql] = sumsq;
            s_apsum[ql] = apsum;
            s_dp2[ql] = dp2;
            s_dn2[ql] = dn2;
            s_cmx[ql] = cmx;
            s_cmn[ql] = cmn;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
    }
    __syncthreads();

    // ---- phase C: one lane per query ----
    if (!in_a) return;
    const double Tq = s_T[tid];
    if (Tq == 0.0) {  // zero-count contig: the reference's normalised row is NaN
        p.scores[p.q_base + qa] = __builtin_nan("");
        if (p.status) atomicAdd(p.status, 1u);
        return;
    }
    const double invT2 = 1.0 / (Tq * Tq);
    const double nq2 = s_sumsq[tid] * invT2, nqp2 = s_apsum[tid] * invT2;
    const CenteredOperand cop = phk_centered_operand(s_sumsq[tid], Tq, (double)s_cmx[tid], (double)s_cmn[tid], (double)FAST_D, p.eb_hsum);
    const double vs = p.vscale / Tq;
    double cn0[3];
    uint32_t labbits;
    {   // phase A's results back from LDS
#pragma unroll
        for (int r = 0; r < 8; ++r) v8[r] = s_v8[r][tid];
        U0 = s_u0[tid];
        ch1[0] = s_ch[0][tid]; ch2[0] = s_ch[1][tid]; ch1[1] = s_ch[2][tid]; ch2[1] = s_ch[3][tid];
        cn0[0] = s_cn[0][tid]; cn0[1] = s_cn[1][tid]; cn0[2] = s_cn[2][tid];
        const uint32_t fl = s_flags[tid];
        labbits = fl & 0xFFu;
        ok0 = (fl & 0x100u) != 0;
        cfill[0] = (fl & 0x200u) != 0;
        cfill[1] = (fl & 0x400u) != 0;
    }
    const double cnp2 = s_cn[3][tid], cnn2 = s_cn[4][tid];
    ErrBound eb;
    eb.A = sqrt(nq2) + p.mu_norm;
    eb.P = sqrt(nqp2);
    eb.Q = cop.Q; eb.I = cop.I; eb.habs = cop.habs + p.eb_babs;
    eb.cA = p.eb_cA; eb.cP = p.eb_cP; eb.cR = p.eb_cR; eb.cabs = p.eb_abs; eb.cQ = p.eb_cQ; eb.cI = p.eb_cI; eb.cIf = p.eb_cIf;
    eb.cM = p.eb_cM; eb.M = p.eb_M;
    const double nqp = eb.P;
    const double nqx = sqrt(nqp2 + cop.shift2);   // |q' - (c0/T - 1/D) 1|: what the low parts multiply (see phase B)
    auto e_hi = [&](int sg, double R) { return nqx * phk_lam_of(hp, sg, R) + eb(R); };
    bool cert = true;      // the k-NN part
    bool cert_c = true;    // the centroid part
    double knn = 0.0, cen = 0.0;
    if (want_knn) {
        const int need = p.kn;
        cert = ok0;
        if (cert) {
            // reach of the need nearest columns from the need-th high-part value; error bounds at that reach when the
            // leaders lie within it
            const double eg = e_hi(0, p.rmax);
            const double d2up = fmax(nqp2 - 2.0 * ((double)v8[need - 1] * vs - eg), 0.0);
            const double R0 = fmin(p.rmax, (nqp + sqrt(d2up)) * (1.0 + 1e-6));
            bool near = true;
            for (int r = 0; r < need; ++r) near = near && cn0[r] <= R0;
            const double Rw = near ? R0 : p.rmax;
            // count-exact error model + the float32 rounding of the low products (22 roundings x 2^-24 < 2^-19)
            const double eh = near ? e_hi(0, R0) : eg, e22 = eb(Rw) + 0x1p-19 * nqx * phk_lam_of(hp, 0, Rw);
            // window members: list positions 0 .. nw-1 (sorted by high-part value)
            const double thr = (double)v8[need - 1] * vs - 2.0 * eh;
            int nw = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) nw += ((double)v8[r] * vs >= thr && v8[r] > -1.0e38f) ? 1 : 0;
            // every column of the window has to be a list member with a refined value
            cert = nw <= PHK_HI_REFINE && (double)U0 * vs < thr;
            // diagnostics: window too wide / reaches past the lists (one uniform address per statement, so that the
            // compiler folds a wave's increments into one atomic; a per-lane address costs ~10 ns per lane)
            if (!cert && nw > PHK_HI_REFINE) atomicAdd(phk_stat_word(p, p.counters + 8, 8), 1u);
            if (!cert && nw <= PHK_HI_REFINE) atomicAdd(phk_stat_word(p, p.counters + 9, 9), 1u);
            if (cert) {
                // refined values of the window's members, descending
                double rv[PHK_HI_REFINE];
                uint32_t rl[PHK_HI_REFINE];
#pragma unroll
                for (int r = 0; r < PHK_HI_REFINE; ++r) {
                    const bool in = r < nw;
                    rv[r] = in ? ((double)v8[r] + (double)s_corr[r][tid]) * vs : -1.0e300;
                    rl[r] = (labbits >> r) & 1u;
                }
#pragma unroll
                for (int a = 1; a < PHK_HI_REFINE; ++a)
#pragma unroll
                    for (int b = a; b > 0; --b) {
                        const bool sw = rv[b] > rv[b - 1];
                        const double tv = rv[b]; const uint32_t tl = rl[b];
                        rv[b] = sw ? rv[b - 1] : rv[b]; rl[b] = sw ? rl[b - 1] : rl[b];
                        rv[b - 1] = sw ? tv : rv[b - 1]; rl[b - 1] = sw ? tl : rl[b - 1];
                    }
                // the need-th and (need+1)-th refined values decide (a window of exactly `need` members is decided)
                const double hi_v = need == 1 ? rv[0] : need == 2 ? rv[1] : rv[2];
                const double lo_v = need == 1 ? rv[1] : need == 2 ? rv[2] : rv[3];
                cert = nw == need || hi_v - lo_v > 2.0 * e22;
                if (!cert) atomicAdd(phk_stat_word(p, p.counters + 10, 10), 1u);   // diagnostics: refined values too close
                int votes = 0;
                for (int r = 0; r < need; ++r) votes += (int)rl[r];
                knn = (2 * votes > need) ? 1.0 : -1.0;
            }
        }
    }
    if (want_cen) {
        auto leader_ok = [&](int k2, double cnorm) {
            if (!cfill[k2]) return false;
            const double eg = e_hi(1 + k2, p.rmax);
            const double d2up = fmax(nqp2 - 2.0 * ((double)ch1[k2] * vs - eg), 0.0);
            const double R0 = fmin(p.rmax, (nqp + sqrt(d2up)) * (1.0 + 1e-6));
            const double eh = cnorm <= R0 ? e_hi(1 + k2, R0) : eg;
            return ((double)ch1[k2] - (double)ch2[k2]) * vs > 2.0 * eh;
        };
        cert_c = leader_ok(0, cnp2) && leader_ok(1, cnn2);
        if (!cert_c) atomicAdd(phk_stat_word(p, p.counters + 11, 11), 1u);   // diagnostics: centroid leader not certified
        const double ep = sqrt(s_dp2[tid] * invT2), en = sqrt(s_dn2[tid] * invT2);   // (the squared distances as exact_d2_g16 forms them)
        cen = tanh((en - ep) / (ep + en));  // scripts/phamer.py:206-209
    }
    if (cert && cert_c) {
        p.scores[p.q_base + qa] = knn + cen;  // scripts/phamer.py:313
    } else {
        // the open part(s) go to phk_rerank16_kernel (MODE 1), which takes the query's lists as they are -- high-part
        // values under the high-part error model -- and decides by exact float64 candidate distances where that
        // suffices; the decided part waits in scores[q]  (entry = query | open parts << 30)
        p.scores[p.q_base + qa] = (cert ? knn : 0.0) + (cert_c ? cen : 0.0);
        const uint32_t open_parts = ((want_knn && !cert) ? 1u : 0u) | ((want_cen && !cert_c) ? 2u : 0u);
        // two lists in one array, so that a wave of the next kernel works on one kind of segment: queries with the k-NN
        // part open from the front, those with only the centroid part open from the back
        if (p.sub_lists) {   // this workgroup's pair of lists (see RerankParams::sub_lists)
            const uint32_t sl = blockIdx.x % p.sub_lists;
            uint32_t *cw = p.stripes + sl * 32u;
            const uint64_t base = (uint64_t)sl * p.sub_cap;
            if (open_parts == 2u) p.slow_list[base + p.sub_cap - 1 - atomicAdd(cw + 12, 1u)] = (uint32_t)qa | (open_parts << 30);
            else p.slow_list[base + atomicAdd(cw + 2, 1u)] = (uint32_t)qa | (open_parts << 30);
        } else if (open_parts == 2u) p.slow_list[p.slow_cap - 1 - atomicAdd(p.counters + 12, 1u)] = (uint32_t)qa | (open_parts << 30);
        else p.slow_list[atomicAdd(p.fb_count + 2, 1u)] = (uint32_t)qa | (open_parts << 30);
    }
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
int phk_launch_decide(phk_ctx *ctx, int src_kind, const void *src, const RerankParams &p) {
    const dim3 grid((unsigned)phk_div_up(p.N, 64));
    if (src_kind == 0) { PHK_LAUNCH(ctx, "phk_decide_kernel", phk_decide_kernel<0><<<grid, dim3(64), 0, ctx->stream>>>(src, p)); }
    else { PHK_LAUNCH(ctx, "phk_decide_kernel", phk_decide_kernel<1><<<grid, dim3(64), 0, ctx->stream>>>(src, p)); }
    return PHK_OK;
}

int phk_launch_decide_gen(phk_ctx *ctx, int dsub, const uint32_t *counts, const RerankParams &p) {
    const dim3 grid((unsigned)phk_div_up(p.N, 64));
#define PHK_DG(DS) PHK_LAUNCH(ctx, "phk_decide_gen_kernel", (phk_decide_gen_kernel<DS><<<grid, dim3(64), 0, ctx->stream>>>(counts, p)))
    switch (dsub) {
        case 2: PHK_DG(2); break;
        case 4: PHK_DG(4); break;
        case 8: PHK_DG(8); break;
        case 16: PHK_DG(16); break;
        default: phk_set_error("phk_launch_decide_gen: D = %d", 256 * dsub); return PHK_ERR_UNSUPPORTED;
    }
#undef PHK_DG
    return PHK_OK;
}

int phk_launch_decide_h(phk_ctx *ctx, bool knn, bool cen, dim3 grid, dim3 block, const uint32_t *counts, const RerankParams &p, const HiParams &hp) {
    PHK_REQUIRE(knn || cen, "phk_launch_decide_h: nothing to decide");
    if (knn && cen) { PHK_LAUNCH(ctx, "phk_decide_h_kernel", (phk_decide_h_kernel<true, true><<<grid, block, 0, ctx->stream>>>(counts, p, hp))); }
    else if (knn) { PHK_LAUNCH(ctx, "phk_decide_h_kernel", (phk_decide_h_kernel<true, false><<<grid, block, 0, ctx->stream>>>(counts, p, hp))); }
    else { PHK_LAUNCH(ctx, "phk_decide_h_kernel", (phk_decide_h_kernel<false, true><<<grid, block, 0, ctx->stream>>>(counts, p, hp))); }
    return PHK_OK;
}
