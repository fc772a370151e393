// score_rerank.hip -- the decision stage's wave-level kernels: every emitted number is decided here (or in score_decide.hip)
// from the proposal pass's candidate lists by float64 arithmetic of the reference's form, or by a certified ordering.
//   phk_rerank_kernel     one wavefront per query, any supported D (lane = 4 DSUB dimensions)
//   phk_rerank16_kernel   D = 256: four queries per wavefront (16 lanes each)
//   phk_rerank_h_kernel   general D, lists of high-part values: window, low products, count-exact margin test
// Shared device code (parameters, error model, exact distances): score_decide.h.
#include "score_decide.h"

// Resolve one segment for the wave's query: find the `need` best columns.
//   returns false if the candidate set cannot be certified (-> fallback queue);
//   out_idx[0..need) = column indices of the best; out_d2 = exact d^2 of the best (computed when
//   want_d2, or when the order had to be decided by exact distances).
// Columns with |r'| > |q'| + d_need cannot be among the `need` nearest (triangle inequality), so the
// error bound only has to hold for columns with |r'| <= R0 = |q'| + (upper bound of d_need).
template <int DSUB>
__device__ bool resolve_segment(const RerankParams &p, uint64_t q, int seg, uint32_t ncols, int need,
                                const double (&qd)[4 * DSUB], double Tq, double invT2, double nqp2, const ErrBound &eb, const double vs,
                                const double *rows, const double *colnorm, bool want_d2, int lane,
                                uint32_t (&out_idx)[3], double &out_d2, float pre_v, uint32_t pre_i, float pre_u,
                                bool allow_margin = true, double u_extra = 0.0) {
    // lanes 0..7 hold the 8 candidates (half = lane>>2, slot = lane&3), out of the lists the caller loaded up front
    float v = __shfl(pre_v, seg * 8 + (lane & 7));
    uint32_t ix = __shfl(pre_i, seg * 8 + (lane & 7));
    if (lane >= 8 || ix >= ncols) v = -3.0e38f;  // padding / empty slot
    if (lane >= 8) ix = 0xFFFFFFFFu;
    // vs: computed values -> v units (a power of two, divided by the row sum for the count-exact proposal)
    // every column the two half-lists dropped has a computed value <= the larger of their
    // best-dropped values (-3e38 when nothing real was dropped)
    // (u_extra: lists whose values are short of the true ones by up to that much -- the two-part int8 sweep)
    const double U = fmax((double)__shfl(pre_u, seg * 2), (double)__shfl(pre_u, seg * 2 + 1)) * vs + u_extra;
    // rank of each candidate among the 8 (descending v, ties by lane)
    int rank = 0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const float ov = __shfl(v, m);
        rank += (lane < 8 && (ov > v || (ov == v && m < lane))) ? 1 : 0;
    }
    float rv[4];
    uint32_t ri[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned long long bal = __ballot(lane < 8 && rank == r);
        const int srcl = __ffsll((long long)bal) - 1;
        rv[r] = __shfl(v, srcl);
        ri[r] = __shfl(ix, srcl);
    }
    const double nqp = sqrt(nqp2);
    const double eps_g = eb(p.rmax);  // holds for every column
    if (allow_margin && ri[need - 1] < ncols) {
        // upper bound of the true need-th nearest distance from the computed candidates
        const double d2up = fmax(nqp2 - 2.0 * ((double)rv[need - 1] * vs - eps_g), 0.0);
        const double R0 = fmin(p.rmax, (nqp + sqrt(d2up)) * (1.0 + 1e-6));
        bool near = true;  // the top `need` computed candidates all lie within R0
        for (int r = 0; r < need; ++r) near = near && colnorm[ri[r]] <= R0;
        const double eps_m = near ? eb(R0) : eps_g;
        // certified by margin: the need-th and (need+1)-th computed values are > 2 eps apart
        if (((double)rv[need - 1] - (double)rv[need]) * vs > 2.0 * eps_m) {
#pragma unroll
            for (int r = 0; r < 3; ++r) out_idx[r] = ri[r];
            if (want_d2) out_d2 = exact_d2<DSUB>(qd, Tq, invT2, rows + (uint64_t)ri[0] * (256 * DSUB), lane);
            return true;
        }
    }
    if (lane == 0) atomicAdd(phk_stat_word(p, p.fb_count + 1, 1), 1u);  // statistics: resolved by exact distances
    // not certified: exact float64 distances for all (valid) candidates
    double best[3] = {INFINITY, INFINITY, INFINITY};
    uint32_t bidx[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (int m = 0; m < 8; ++m) {
        const uint32_t c = __shfl(ix, m);
        if (c >= ncols) continue;
        const double d2 = exact_d2<DSUB>(qd, Tq, invT2, rows + (uint64_t)c * (256 * DSUB), lane);
        // insert (d2, c) ascending; ties to the lower column index
        if (d2 < best[2] || (d2 == best[2] && c < bidx[2])) {
            best[2] = d2; bidx[2] = c;
            if (best[2] < best[1] || (best[2] == best[1] && bidx[2] < bidx[1])) {
                double t = best[1]; best[1] = best[2]; best[2] = t;
                uint32_t ti = bidx[1]; bidx[1] = bidx[2]; bidx[2] = ti;
                if (best[1] < best[0] || (best[1] == best[0] && bidx[1] < bidx[0])) {
                    t = best[0]; best[0] = best[1]; best[1] = t;
                    ti = bidx[0]; bidx[0] = bidx[1]; bidx[1] = ti;
                }
            }
        }
    }
    if (bidx[need - 1] == 0xFFFFFFFFu) return false;
    // the need-th best (exact) must beat what any dropped column within reach could be
    const double R0x = fmin(p.rmax, (nqp + sqrt(best[need - 1])) * (1.0 + 1e-6));
    const double tv = 0.5 * (nqp2 - best[need - 1]);
    if (!(tv > U + eb(R0x))) return false;
#pragma unroll
    for (int r = 0; r < 3; ++r) out_idx[r] = bidx[r];
    out_d2 = best[0];
    return true;
}

// one query, one wave (the body of phk_rerank_kernel)
// MU_LDS: the training mean is read from LDS at byte offset mu_lds (address space 3: a generic pointer would turn every
// read into a flat load, which also counts on the vector-memory counter and serialises the kernel's other loads)
// I8H: the lists come from the two-part int8 sweep (see RerankParams::L8 and the refinement step below)
template <int SRC, int DSUB, bool MU_LDS, bool I8H>
__device__ __forceinline__ void rerank_one_query(const void *__restrict__ src, const RerankParams &p, uint64_t q, uint32_t mu_lds,
                                                 int lane) {
    constexpr int D = 256 * DSUB;
    typedef __attribute__((address_space(3))) const double lds_cdouble;
    // this lane's query elements, dims 256*sub + 4*lane .. +3: the integer counts with their sum Tq (count rows) or the
    // float64 row with Tq = 1 -- see exact_d2
    double qd[4 * DSUB];
    double Tq = 1.0, invT2 = 1.0;
    double vs = p.vscale;
    bool nan_row = false;
    double ssq = 0.0, rtq = 1.0;   // count rows: sum of squares, reciprocal of the row sum
    CenteredOperand cop = {0.0, 0.0, 0.0, 0.0};
    uint32_t xq[I8H ? DSUB : 1];   // I8H: this lane's centred counts as four int8 per 256-dimension chunk (the sweep's operand)
    // The candidate lists of all three segments are requested first, beside the query row: lane l < 24 holds candidate l & 7
    // of segment l >> 3, lane l < 6 the best dropped value of half-list l.  Loaded where they are used, each segment's
    // lists were one more dependent round trip in a kernel that is a chain of them (10 M queries at configs[2]).
    float pre_v = -3.0e38f, pre_u = -3.0e38f;
    uint32_t pre_i = 0xFFFFFFFFu;
    if (lane < 24) {
        const uint64_t o = cand_at(lane >> 3, (lane & 7) >> 2, lane & 3, q, p.N);
        pre_v = p.cand_v[o];
        pre_i = p.cand_i[o];
    }
    if (lane < 6) pre_u = p.cand_u[candu_at(lane >> 1, lane & 1, q, p.N)];
    if (SRC == 0) {
        const uint32_t *row = static_cast<const uint32_t *>(src) + q * D;
        uint4 c[DSUB];
        uint32_t s = 0;
#pragma unroll
        for (int sub = 0; sub < DSUB; ++sub) {
            c[sub] = reinterpret_cast<const uint4 *>(row + 256 * sub)[lane];
            s += c[sub].x + c[sub].y + c[sub].z + c[sub].w;
        }
        s = wave_sum(s);
        nan_row = s == 0;
        const double ds = (double)s, ry = 1.0 / ds;
        if (p.per_row_scale) vs = p.vscale * ry;   // (= vscale / ds up to the reciprocal's rounding; vs scales margins and bounds, never a score)
        uint32_t cmx = 0, cmn = 0xFFFFFFFFu;
        double sq = 0.0, qc2 = 0.0;
        const double rcen = (double)phk_row_center(s, D);
        rtq = ry;
        Tq = ds;
        invT2 = 1.0 / (ds * ds);
        if (I8H) {
            const uint32_t cen = phk_row_center(s, D);
#pragma unroll
            for (int sub = 0; sub < DSUB; ++sub)   // (rows beyond the int8 range have empty lists: their bytes are never used)
                xq[sub] = ((c[sub].x - cen) & 0xFFu) | (((c[sub].y - cen) & 0xFFu) << 8) | (((c[sub].z - cen) & 0xFFu) << 16) |
                          ((c[sub].w - cen) << 24);
        }
#pragma unroll
        for (int sub = 0; sub < DSUB; ++sub) {
            const double x0 = (double)c[sub].x, x1 = (double)c[sub].y, x2 = (double)c[sub].z, x3 = (double)c[sub].w;
            qd[4 * sub + 0] = x0; qd[4 * sub + 1] = x1; qd[4 * sub + 2] = x2; qd[4 * sub + 3] = x3;
            sq = fma(x0, x0, fma(x1, x1, fma(x2, x2, fma(x3, x3, sq))));
            if (!I8H) {   // (the largest / smallest count: for the maximum-norm term of the f16 chains' bound only)
                cmx = max(max(cmx, max(c[sub].x, c[sub].y)), max(c[sub].z, c[sub].w));
                cmn = min(min(cmn, min(c[sub].x, c[sub].y)), min(c[sub].z, c[sub].w));
            }
            if (!I8H && DSUB > 1 && p.per_row_scale && p.eb_cAmax > 0.0) {   // norm of this 256-dimension chunk of c - c0 (f16 chains)
                const double y0 = x0 - rcen, y1 = x1 - rcen, y2 = x2 - rcen, y3 = x3 - rcen;
                qc2 = fmax(qc2, wave_sum(fma(y0, y0, fma(y1, y1, fma(y2, y2, y3 * y3)))));
            }
        }
        ssq = wave_sum(sq);   // sum of squared counts, exact: |q|^2 = ssq / T^2
        if (p.per_row_scale && !nan_row) {  // the operand of the count-exact chain: the counts minus their centre
            // (largest / smallest count: only where the bound has a maximum-norm term -- the f16 chains)
            const double cmax = (!I8H && p.eb_cI > 0.0) ? wave_max((double)cmx) : rcen, cmin = (!I8H && p.eb_cI > 0.0) ? -wave_max(-(double)cmn) : rcen;
            cop = phk_centered_operand_fast(ssq, ds, ry, cmax, cmin, (double)D, p.eb_hsum);
            if (!I8H && DSUB > 1 && p.eb_cAmax > 0.0) cop.Q = phk_sqrt_up(qc2) * ry * (1.0 + 1e-9);   // (with the observed running sums)
        }
    } else {
        const double *row = static_cast<const double *>(src) + q * D;
        bool bad = false;
#pragma unroll
        for (int sub = 0; sub < DSUB; ++sub) {
            const double2 a = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane];
            const double2 b = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane + 1];
            qd[4 * sub + 0] = a.x; qd[4 * sub + 1] = a.y; qd[4 * sub + 2] = b.x; qd[4 * sub + 3] = b.y;
            bad |= a.x != a.x || a.y != a.y || b.x != b.x || b.y != b.y;
        }
        nan_row = __any(bad);
    }
    if (nan_row) {  // zero-count contig: the reference's normalised row is NaN
        if (lane == 0) {
            p.scores[p.q_base + q] = __builtin_nan("");
            if (p.status) atomicAdd(p.status, 1u);
        }
        return;
    }
    double aq = 0.0, ap = 0.0, am = 0.0, pc2 = 0.0;
#pragma unroll
    for (int sub = 0; sub < DSUB; ++sub) {
        double2 m0, m1;
        if (MU_LDS) {
            lds_cdouble *lm = (lds_cdouble *)(uintptr_t)mu_lds + 256 * sub + 4 * lane;
            m0.x = lm[0]; m0.y = lm[1]; m1.x = lm[2]; m1.y = lm[3];
        } else {
            m0 = reinterpret_cast<const double2 *>(p.mu64 + 256 * sub)[2 * lane];
            m1 = reinterpret_cast<const double2 *>(p.mu64 + 256 * sub)[2 * lane + 1];
        }
        // q' in units of 1 / Tq (count rows: c_i - Tq mu_i; float64 rows: Tq = 1, q_i - mu_i exactly)
        const double c0 = fma(-Tq, m0.x, qd[4 * sub + 0]), c1 = fma(-Tq, m0.y, qd[4 * sub + 1]);
        const double c2 = fma(-Tq, m1.x, qd[4 * sub + 2]), c3 = fma(-Tq, m1.y, qd[4 * sub + 3]);
        if (SRC != 0)
            aq = fma(qd[4 * sub + 0], qd[4 * sub + 0], fma(qd[4 * sub + 1], qd[4 * sub + 1],
                     fma(qd[4 * sub + 2], qd[4 * sub + 2], fma(qd[4 * sub + 3], qd[4 * sub + 3], aq))));
        const double apc = fma(c0, c0, fma(c1, c1, fma(c2, c2, c3 * c3)));
        ap += apc;
        if (!(SRC == 0 && p.per_row_scale)) {   // split-f16 lists only: maximum norm and chunk norms of q'
            am = fmax(fmax(am, fmax(fabs(c0), fabs(c1))), fmax(fabs(c2), fabs(c3)));
            if (DSUB > 1) pc2 = fmax(pc2, wave_sum(apc));
        }
    }
    // |q|^2 enters the error bound only: for count rows from the exact sum of squares (+ slack for the two roundings)
    const double nq2 = SRC == 0 ? ssq * (rtq * rtq) * (1.0 + 1e-12) : wave_sum(aq);
    const double nqp2 = wave_sum(ap) * invT2;
    if (SRC == 0) {   // (split-f16 lists of count rows: the maximum norm and the chunk norms in units of q)
        am *= rtq;
        pc2 *= invT2;
    }
    ErrBound eb;   // (its norms: upper bounds, phk_sqrt_up)
    eb.A = phk_sqrt_up(nq2) + p.mu_norm;
    const double nqp_up = phk_sqrt_up(nqp2);
    eb.P = nqp_up;
    eb.cA = p.eb_cA; eb.cP = p.eb_cP; eb.cR = p.eb_cR; eb.cabs = p.eb_abs; eb.cQ = p.eb_cQ; eb.cI = p.eb_cI; eb.cIf = p.eb_cIf;
    if (SRC == 0 && p.per_row_scale) {   // count-exact lists: the chain's query operand is c - c0
        eb.Q = cop.Q; eb.I = cop.I; eb.habs = cop.habs;
        eb.P = phk_sqrt_up(nqp2 + cop.shift2);   // (high-parts-only lists: what the low parts multiply)
    } else {                             // split-f16 lists: the chain's query operand is q' (largest chunk norm)
        eb.Q = DSUB > 1 ? sqrt(pc2) * (1.0 + 1e-12) : eb.P;
        eb.I = wave_max(am);
    }
    if (p.cand_a)   // the running sums this query's sweep met at its chunk boundaries, in v units
        eb.habs += p.eb_cAmax * 5.9604644775390625e-08 * (double)fmaxf(p.cand_a[q], p.cand_a[p.N + q]) * vs * 1.001;

    bool ok = true;
    double knn = 0.0, cen = 0.0;
    uint32_t idx[3];
    double d2;
    // Lists of the two-part int8 sweep: a value w^h lacks g_j S_L, at most el = |c - c0| / T * lam8[segment] in v units.  With
    // h_need the need-th best list value of a segment and e_h = el + eps: every column that can be among the `need` nearest
    // has w^h >= h_need - 2 e_h (the leaders' true values are >= h_need - e_h; below the window a true value is < that).  The
    // window's members that are list members get g_j S_L added -- an exact integer dot product with the column's L digits,
    // 4 DSUB bytes per lane -- and then carry the three-part sweep's value; the window must end above everything the
    // half-lists dropped (uok), else the segment is left to the exact candidate distances.  Lane l < 24 works for candidate
    // l & 7 of segment l >> 3, as in certify_segments.
    uint64_t uok = ~0ull;
    // (a row beyond the int8 operand: the sweep stored its sentinel, 3e38 as the best dropped value)
    const bool big_row = p.q2_count ? __any(lane < 6 && pre_u > 1.0e38f) != 0 : false;
    if (I8H) {
        const int g3 = lane >> 3, gb = lane & 56;
        const uint32_t ncols = g3 == 0 ? (uint32_t)p.M : g3 == 1 ? (uint32_t)p.n_cpos : g3 == 2 ? (uint32_t)p.n_cneg : 0u;
        const int need = g3 == 0 ? p.kn : 1;
        const bool valid = pre_i < ncols;
        const float gq = valid ? p.T8[(uint64_t)(p.t8_blk[g3 < 3 ? g3 : 0] + (pre_i >> 5)) * 64 + (pre_i & 31u)] : 0.0f;
        const uint32_t gcol = pre_i + (g3 == 0 ? 0u : g3 == 1 ? (uint32_t)p.M : (uint32_t)(p.M + p.n_cpos));
        const float v = valid ? pre_v : -3.0e38f;
        int rank = 0;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float ov = __shfl(v, gb | m);
            rank += (ov > v || (ov == v && m < (lane & 7))) ? 1 : 0;
        }
        const unsigned long long bal = __ballot(valid && rank == need - 1);
        const unsigned long long mine = (bal >> gb) & 0xFFull;
        const float hneed = __shfl(v, gb + (mine ? __ffsll((long long)mine) - 1 : 0));
        const double el = cop.Q * p.lam8[g3 < 3 ? g3 : 0] * (1.0 + 1.0e-6);
        const double thr = (double)hneed * vs - 2.0 * (el + eb(p.rmax));
        const bool inwin = valid && mine != 0 && (double)v * vs >= thr;
        const double Ug = (double)fmaxf(__shfl(pre_u, 2 * (g3 < 3 ? g3 : 0)), __shfl(pre_u, 2 * (g3 < 3 ? g3 : 0) + 1)) * vs;
        uok = __ballot(mine != 0 && Ug < thr && (lane & 7) == 0 && lane < 24);
        // a window of exactly `need` members is decided as it stands (the members ARE the nearest; the margin test below
        // passes on the list values, the next one lying 2 e_h lower): only wider windows are refined
        unsigned long long wm = __ballot(inwin && lane < 24);
        const bool wide = __popcll((wm >> gb) & 0xFFull) > need;
        wm = __ballot(inwin && wide && lane < 24);
        while (wm) {   // wave-uniform; two members per round trip
            const int m0 = __ffsll((long long)wm) - 1;
            wm &= wm - 1;
            const int m1 = wm ? __ffsll((long long)wm) - 1 : m0;
            wm &= wm - 1;
            const uint32_t *r0 = reinterpret_cast<const uint32_t *>(p.L8 + (uint64_t)__shfl(gcol, m0) * D) + lane;
            const uint32_t *r1 = reinterpret_cast<const uint32_t *>(p.L8 + (uint64_t)__shfl(gcol, m1) * D) + lane;
            uint32_t w0[DSUB], w1[DSUB];
#pragma unroll
            for (int sub = 0; sub < DSUB; ++sub) {
                w0[sub] = r0[64 * sub];
                w1[sub] = r1[64 * sub];
            }
            int a0 = 0, a1 = 0;
#pragma unroll
            for (int sub = 0; sub < DSUB; ++sub) {
                a0 = __builtin_amdgcn_sdot4((int)xq[sub], (int)w0[sub], a0, false);
                a1 = __builtin_amdgcn_sdot4((int)xq[sub], (int)w1[sub], a1, false);
            }
            a0 = wave_sum_i32(a0);
            a1 = wave_sum_i32(a1);
            if (lane == m0) pre_v = fmaf((float)a0, gq, pre_v);
            if (lane == m1 && m1 != m0) pre_v = fmaf((float)a1, gq, pre_v);
        }
    }
    // all three margin tests side by side; what they certify needs no further list work
    uint32_t cri[3];
    uint64_t certified = certify_segments(p, nqp2, nqp_up, eb, vs, lane, pre_v, pre_i, cri);
    if (I8H) {
        certified &= uok;
        // statistics: a window of a segment the method uses reached past the lists (rare: one atomic per such query)
        const uint64_t used = ((p.method & PHK_METHOD_KNN) ? 0x01ull : 0ull) | ((p.method & PHK_METHOD_KMEANS) ? 0x010100ull : 0ull);
        if (lane == 0 && (uok & used) != used && p.counters && !big_row) atomicAdd(p.counters + 9, 1u);
    }
    const double uex0 = I8H ? cop.Q * p.lam8[0] * (1.0 + 1.0e-6) : 0.0, uex1 = I8H ? cop.Q * p.lam8[1] * (1.0 + 1.0e-6) : 0.0,
                 uex2 = I8H ? cop.Q * p.lam8[2] * (1.0 + 1.0e-6) : 0.0;
    if (p.method & PHK_METHOD_KNN) {
        if (certified & 1ull) {
#pragma unroll
            for (int r = 0; r < 3; ++r) idx[r] = __shfl(cri[r], 0);
        } else {
            ok = resolve_segment<DSUB>(p, q, 0, (uint32_t)p.M, p.kn, qd, Tq, invT2, nqp2, eb, vs, p.R64, p.colnorm, false, lane, idx, d2, pre_v, pre_i, pre_u, !I8H, uex0);
        }
        if (ok) {
            int votes = 0;
            for (int r = 0; r < p.kn; ++r) votes += p.labels[idx[r]] ? 1 : 0;
            knn = (2 * votes > p.kn) ? 1.0 : -1.0;
        }
    }
    if (ok && (p.method & PHK_METHOD_KMEANS)) {
        double dp2 = 0.0, dn2 = 0.0;
        if (certified & (1ull << 8))
            dp2 = exact_d2<DSUB>(qd, Tq, invT2, p.C64 + (uint64_t)__shfl(cri[0], 8) * D, lane);
        else
            ok = resolve_segment<DSUB>(p, q, 1, (uint32_t)p.n_cpos, 1, qd, Tq, invT2, nqp2, eb, vs, p.C64, p.colnorm + p.M, true, lane, idx, dp2, pre_v, pre_i, pre_u, !I8H, uex1);
        if (ok && (certified & (1ull << 16)))
            dn2 = exact_d2<DSUB>(qd, Tq, invT2, p.C64 + (p.n_cpos + (uint64_t)__shfl(cri[0], 16)) * D, lane);
        else if (ok)
            ok = resolve_segment<DSUB>(p, q, 2, (uint32_t)p.n_cneg, 1, qd, Tq, invT2, nqp2, eb, vs, p.C64 + p.n_cpos * D,
                                 p.colnorm + p.M + p.n_cpos, true, lane, idx, dn2, pre_v, pre_i, pre_u, !I8H, uex2);
        if (ok) {
            if (p.pend) {   // two square roots, a division and a tanh in float64 are ~180 instructions of this wave, for one
                            // number: left to a lane-per-query kernel (phk_finish_cen_kernel, the same expressions)
                if (lane == 0) {
                    const uint64_t oq = p.out_map ? (uint64_t)p.out_map[q] : q;
                    p.pend[2 * oq] = dp2;
                    p.pend[2 * oq + 1] = dn2;
                }
            } else {
                const double ep = sqrt(dp2), en = sqrt(dn2);
                cen = tanh((en - ep) / (ep + en));  // scripts/phamer.py:206-209
            }
        }
    }
    if (lane == 0) {
        const uint64_t oq = p.out_map ? (uint64_t)p.out_map[q] : q;
        if (ok) {
            p.scores[p.q_base + oq] = knn + cen;  // scripts/phamer.py:313 (cen: see pend)
        } else if (p.q2_count && big_row) {
            p.q2_big[atomicAdd(p.q2_count + 1, 1u)] = (uint32_t)q;
        } else if (p.q2_count && p.q2_wide) {
            p.q2_wide[atomicAdd(p.q2_count, 1u)] = (uint32_t)q;
        } else {
            const uint32_t slot = atomicAdd(p.fb_count, 1u);
            p.fb_list[slot] = (uint32_t)oq;
        }
    }
}

// One wave per query.  D >= 2048: a workgroup walks its share of the queries (grid-stride) with the training mean in LDS,
// loaded once -- read from memory per query it was 8 D bytes through the vector cache, a quarter of the kernel's traffic
// (configs[4]: 11.4 -> 9.6 ms).  Smaller D: one query per wave and launch slot, the mean from the cache (the 8 KB of
// D = 1024 stay resident there, and the walk was measured slower: configs[2] 37.5 -> 44.7 ms).
#ifndef PHK_RERANK_WAVES
#define PHK_RERANK_WAVES 4
#endif
template <int SRC, int DSUB, bool I8H = false>
__global__ __launch_bounds__(256, (DSUB <= 4 ? PHK_RERANK_WAVES : 1)) void phk_rerank_kernel(const void *__restrict__ src, RerankParams p) {
    constexpr int D = 256 * DSUB;
    constexpr bool WALK = DSUB >= 8;
    const int lane = threadIdx.x & 63;
    __shared__ double s_mu[WALK ? D : 2];
    if (WALK) {
        for (int i = threadIdx.x; i < D / 2; i += 256)
            reinterpret_cast<double2 *>(s_mu)[i] = reinterpret_cast<const double2 *>(p.mu64)[i];
        __syncthreads();
    }
    // listed (slow_back == 2): only the queries phk_rerank_h_kernel passed on (slow_list, fb_count[2] of them)
    const uint64_t nq = p.slow_back == 2 ? (uint64_t)phk_uniform_load(p.fb_count + 2) : p.N;
    const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t mu_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)s_mu;
    uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (WALK) {
        for (; w < nq; w += stride)
            rerank_one_query<SRC, DSUB, true, I8H>(src, p, p.slow_back == 2 ? (uint64_t)(p.slow_list[w] & 0x3FFFFFFFu) : w, mu_lds, lane);
    } else if (w < nq) {   // (no loop: its live state costs the registers that keep four waves per SIMD)
        rerank_one_query<SRC, DSUB, false, I8H>(src, p, p.slow_back == 2 ? (uint64_t)(p.slow_list[w] & 0x3FFFFFFFu) : w, 0u, lane);
    }
}

// ------------------------------------------------------------------------------------
// 2b. the same decision stage for D = 256 with FOUR queries per wavefront (16 lanes each, 16
//     dimensions per lane).  The one-wave-per-query kernel above is latency bound (a chain of ~6
//     dependent memory round trips per query); packing 4 queries into a wave quarters the number of
//     such chains per SIMD.  Control flow is uniform per wave: a group that does not need a step
//     runs it predicated on safe addresses.
// ------------------------------------------------------------------------------------
// per-group version of resolve_segment; `live` = this group still needs an answer.  Returns ok.
__device__ static bool resolve_segment_g16(const RerankParams &p, uint64_t q, int seg, uint32_t ncols, int need,
                                    const double (&qd)[16], double Tq, double invT2, double nqp2, const ErrBound &eb,
                                    const double vs, const double *rows, const double *colnorm, bool want_d2, bool live,
                                    int lane, float v_in, uint32_t ix_in, float u_in, uint32_t (&out_idx)[3],
                                    double &out_d2) {
    // v_in / ix_in: this lane's candidate of the segment (lanes t < 8: half t >> 2, slot t & 3), u_in: the
    // best-dropped value of half-list t & 1 -- loaded by the caller together with the query row, so that the
    // three segments' lists cost one memory round trip, not three
    const int t = lane & 15, base = lane & 48;
    float v = -3.0e38f;
    uint32_t ix = 0xFFFFFFFFu;
    if (t < 8) {
        v = ix_in >= ncols ? -3.0e38f : v_in;
        ix = ix_in;
    }
    const double U = fmax((double)u_in, (double)__shfl_xor(u_in, 1)) * vs;
    // rank among the group's 8 candidates, then values / indices by rank
    int rank = 0;
    float cvv[8];
    uint32_t cix[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        cvv[m] = __shfl(v, base + m);
        cix[m] = __shfl(ix, base + m);
        rank += (t < 8 && (cvv[m] > v || (cvv[m] == v && m < t))) ? 1 : 0;
    }
    float rv[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    uint32_t ri[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int rm = __shfl(rank, base + m);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rv[r] = rm == r ? cvv[m] : rv[r];
            ri[r] = rm == r ? cix[m] : ri[r];
        }
    }
    const double nqp = sqrt(nqp2);
    const double eps_g = eb(p.rmax);
    double eps_w = eps_g;   // the bound the window of the exact route uses (the margin test's)
    bool certified = false;
    if (pick4(ri, need - 1) < ncols) {
        const double d2up = fmax(nqp2 - 2.0 * ((double)pick4(rv, need - 1) * vs - eps_g), 0.0);
        const double R0 = fmin(p.rmax, (nqp + sqrt(d2up)) * (1.0 + 1e-6));
        double cnr[3] = {0.0, 0.0, 0.0};  // independent loads (no short-circuit chain of round trips)
#pragma unroll
        for (int r = 0; r < 3; ++r)
            if (r < need) cnr[r] = colnorm[ri[r]];
        bool near = true;
#pragma unroll
        for (int r = 0; r < 3; ++r) near = near && (r >= need || cnr[r] <= R0);
        const double eps_m = near ? eb(R0) : eps_g;
        eps_w = eps_m;
        certified = ((double)pick4(rv, need - 1) - (double)pick4(rv, need)) * vs > 2.0 * eps_m;
    }
    bool ok = certified;
    double best[3] = {INFINITY, INFINITY, INFINITY};
    uint32_t bidx[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const bool need_exact = live && !certified;
    if (__any(need_exact)) {  // wave-uniform: some group must decide by exact float64 distances
        if (need_exact && t == 0) atomicAdd(phk_stat_word(p, p.fb_count + 1, 1), 1u);
        // only candidates inside the window can be among the `need` nearest: one whose computed value lies more than
        // 2 eps below the need-th best has a true value below the true need-th best (same argument as the margin test).
        // The lists are sorted within each half, so the late slots are skipped by whole waves most of the time.
        const double wthr = (double)pick4(rv, need - 1) * vs - 2.0 * eps_w;
#pragma unroll   // (fully: cix[m] / cvv[m] under a run-time m would live in scratch memory)
        for (int m = 0; m < 8; ++m) {
            const uint32_t c = cix[m];
            const bool valid = need_exact && c < ncols && (double)cvv[m] * vs >= wthr;
            if (!__any(valid)) continue;
            const double d2 = exact_d2_g16(qd, Tq, invT2, rows + (uint64_t)(valid ? c : 0u) * FAST_D, t);
            if (valid && (d2 < best[2] || (d2 == best[2] && c < bidx[2]))) {
                best[2] = d2; bidx[2] = c;
                if (best[2] < best[1] || (best[2] == best[1] && bidx[2] < bidx[1])) {
                    double td = best[1]; best[1] = best[2]; best[2] = td;
                    uint32_t ti = bidx[1]; bidx[1] = bidx[2]; bidx[2] = ti;
                    if (best[1] < best[0] || (best[1] == best[0] && bidx[1] < bidx[0])) {
                        td = best[0]; best[0] = best[1]; best[1] = td;
                        ti = bidx[0]; bidx[0] = bidx[1]; bidx[1] = ti;
                    }
                }
            }
        }
        if (need_exact) {
            ok = false;
            if (pick3(bidx, need - 1) != 0xFFFFFFFFu) {
                const double bneed = pick3(best, need - 1);
                const double R0x = fmin(p.rmax, (nqp + sqrt(bneed)) * (1.0 + 1e-6));
                const double tv = 0.5 * (nqp2 - bneed);
                ok = tv > U + eb(R0x);
            }
        }
    }
    if (certified) {
#pragma unroll
        for (int r = 0; r < 3; ++r) out_idx[r] = ri[r];
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) out_idx[r] = bidx[r];
    }
    if (want_d2) {  // distance to the best column (centroid segments): one exact evaluation when certified
        const uint32_t c0 = ri[0];  // speculative: the row is fetched alongside the column norms, not after the verdict
        const double d2c = exact_d2_g16(qd, Tq, invT2, rows + (uint64_t)(c0 < ncols ? c0 : 0u) * FAST_D, t);
        out_d2 = certified ? d2c : best[0];
    }
    return ok;
}

// MODE 0: every query of the batch; 1: the queries phk_decide_kernel handed over (slow_list); 2: second chance -- rows
// map[0 .. *map_count) with their lists at dense positions (see RerankParams)
template <int SRC, int MODE>
__global__ __launch_bounds__(256, 3) void phk_rerank16_kernel(const void *__restrict__ src, RerankParams p) {
    const int lane = threadIdx.x & 63, t = lane & 15;
    uint64_t qraw = (((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 4 + (lane >> 4);
    bool inrange;
    uint64_t q, ql;   // the query's row in src / scores; the position of its candidate lists
    uint32_t todo = 3u;
    if (MODE == 1) {
        // slow_back == 3: both lists in one launch -- waves from the front of the grid take the front list (counted by
        // fb_count[2]), waves from its end the back list (counters[12]); front + back <= N, so the two never meet, and a
        // wave serves one kind of list.  (Two launches, each over the whole grid, spent 0.09 ms apiece on waves that
        // read a count and left.)
        uint64_t cnt, base = 0, cap = p.slow_cap;
        bool back = p.slow_back == 1;
        if (p.slow_back == 3 && p.sub_lists) {
            // the grid's waves in sub_lists runs of nlw: run s serves pair s, wave lw of the run from the front of its numbering
            // the pair's front list, from its end the back list (front + back <= sub_cap <= 4 (nlw - 2): the two never meet);
            // neighbouring waves work on neighbouring queries of one list
            const uint64_t wave_id = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
            const uint64_t nlw = ((((uint64_t)gridDim.x * blockDim.x) >> 6)) / p.sub_lists;
            if (wave_id >= nlw * p.sub_lists) return;   // (the waves beyond a whole number of runs)
            const uint32_t sl = __builtin_amdgcn_readfirstlane((uint32_t)(wave_id / nlw));   // (wave-uniform: scalar loads below)
            const uint64_t lw = wave_id % nlw;
            base = (uint64_t)sl * p.sub_cap;
            cap = p.sub_cap;
            uint64_t q4 = lw * 4;
            cnt = phk_uniform_load(p.stripes + sl * 32u + 2u);
            if (q4 >= cnt) {
                q4 = (nlw - 1 - lw) * 4;
                cnt = phk_uniform_load(p.stripes + sl * 32u + 12u);
                back = true;
            }
            qraw = q4 + (uint64_t)(lane >> 4);
        } else {
            cnt = phk_uniform_load(p.slow_back == 1 ? p.counters + 12 : p.fb_count + 2);
            if (p.slow_back == 3 && (qraw & ~3ull) >= cnt) {
                const uint64_t nwave4 = (((uint64_t)gridDim.x * blockDim.x) >> 6) * 4;
                qraw = nwave4 - 4 - (qraw & ~3ull) + (qraw & 3ull);   // wave k from the end, same lane group
                cnt = phk_uniform_load(p.counters + 12);
                back = true;
            }
        }
        if ((qraw & ~3ull) >= cnt) return;
        inrange = qraw < cnt;
        // entry = query | todo << 30: which parts are still open (bit 0 the k-NN vote, bit 1 the centroid metric; 0 = both).
        // A part the sender has decided already sits in scores[q] and is only added to.
        const uint64_t pos = inrange ? qraw : cnt - 1;
        const uint32_t entry = p.slow_list[base + (back ? cap - 1 - pos : pos)];
        q = ql = entry & 0x3FFFFFFFu;
        todo = entry >> 30 ? entry >> 30 : 3u;
    } else if (MODE == 2) {
        const uint64_t cnt_all = phk_uniform_load(p.map_count);
        // a handful of rows is cheaper to brute-force than to sweep (one workgroup's sweep is ~0.2 ms of latency):
        // the second proposal pass stands down below PHK_SECOND_MIN rows (score_model.h) and so does this kernel
        const uint64_t cnt = cnt_all < PHK_SECOND_MIN ? 0 : (cnt_all < p.N ? cnt_all : p.N);
        // rows without a second list set (beyond its capacity, or all of them when the pass stood down) go straight to
        // the brute-force queue
        for (uint64_t i = cnt + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt_all; i += (uint64_t)gridDim.x * blockDim.x)
            p.fb_list[atomicAdd(p.fb_count, 1u)] = p.map[i];
        if ((qraw & ~3ull) >= cnt) return;
        inrange = qraw < cnt;
        ql = inrange ? qraw : cnt - 1;
        q = p.map[ql];
    } else {
        if ((qraw & ~3ull) >= p.N) return;  // whole wave past the end
        inrange = qraw < p.N;
        q = ql = inrange ? qraw : p.N - 1;
    }
    double qd[16];
    double vs = p.vscale;
    bool nan_row = false;
    double Tq = 1.0, invT2 = 1.0;  // row sum and 1 / Tq^2 (counts); 1 for float64 rows
    uint32_t cmx = 0, cmn = 0xFFFFFFFFu;   // largest / smallest count of the row
    // the query's six half-lists, fetched with the row
    float lv[NSEG], lu[NSEG];
    uint32_t lix[NSEG];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        const uint64_t e = cand_at(sg, (t >> 2) & 1, t & 3, ql, p.N);
        lv[sg] = p.cand_v[e];
        lix[sg] = p.cand_i[e];
        lu[sg] = p.cand_u[candu_at(sg, t & 1, ql, p.N)];
    }
    if (SRC == 0) {
        const uint2 *row = reinterpret_cast<const uint2 *>(static_cast<const uint32_t *>(src) + q * FAST_D) + t;
        uint2 c[8];
        uint32_t sum = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c[i] = row[16 * i];
            sum += c[i].x + c[i].y;
        }
        sum = group16_sum(sum);
        nan_row = sum == 0;
        const double ds = (double)sum;
        if (p.per_row_scale) vs = p.vscale / ds;
        Tq = ds;
        invT2 = 1.0 / (ds * ds);
#pragma unroll
        for (int i = 0; i < 8; ++i) {   // the counts themselves: see exact_d2_g16
            qd[2 * i + 0] = (double)c[i].x;
            qd[2 * i + 1] = (double)c[i].y;
            cmx = max(cmx, max(c[i].x, c[i].y));
            cmn = min(cmn, min(c[i].x, c[i].y));
        }
        cmx = group16_max(cmx);
        cmn = group16_min(cmn);
    } else {
        const double2 *row = reinterpret_cast<const double2 *>(static_cast<const double *>(src) + q * FAST_D) + t;
        bool bad = false;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double2 v = row[16 * i];
            qd[2 * i] = v.x;
            qd[2 * i + 1] = v.y;
            bad |= v.x != v.x || v.y != v.y;
        }
        // any NaN in the group's row
        unsigned b = bad ? 1u : 0u;
        b |= __shfl_xor(b, 8); b |= __shfl_xor(b, 4); b |= __shfl_xor(b, 2); b |= __shfl_xor(b, 1);
        nan_row = b != 0;
    }
    double aq = 0.0, ap = 0.0, am = 0.0;
    {
        const double2 *mp = reinterpret_cast<const double2 *>(p.mu64) + t;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double2 m = mp[16 * i];
            const double c0 = fma(-Tq, m.x, qd[2 * i]), c1 = fma(-Tq, m.y, qd[2 * i + 1]);
            aq = fma(qd[2 * i], qd[2 * i], fma(qd[2 * i + 1], qd[2 * i + 1], aq));
            ap = fma(c0, c0, fma(c1, c1, ap));
            am = fmax(am, fmax(fabs(c0), fabs(c1)));
        }
    }
    const double sumsq = group16_sum(aq);
    const double nq2 = sumsq * invT2, nqp2 = group16_sum(ap) * invT2;
    ErrBound eb;
    eb.A = sqrt(nq2) + p.mu_norm;
    eb.P = sqrt(nqp2);
    eb.cA = p.eb_cA; eb.cP = p.eb_cP; eb.cR = p.eb_cR; eb.cabs = p.eb_abs; eb.cQ = p.eb_cQ; eb.cI = p.eb_cI; eb.cIf = p.eb_cIf;
    eb.cM = p.eb_cM; eb.M = p.eb_M;
    if (SRC == 0 && p.per_row_scale) {   // count-exact lists: the chain's query operand is c - c0 (see ErrBound)
        const CenteredOperand cop = phk_centered_operand(sumsq, nan_row ? 1.0 : Tq, (double)cmx, (double)cmn, (double)FAST_D, p.eb_hsum);
        eb.Q = cop.Q; eb.I = cop.I; eb.habs = cop.habs + p.eb_babs;
        eb.P = sqrt(nqp2 + cop.shift2);   // (high-parts-only lists: what the low parts multiply)
    } else {                             // split-f16 lists: the chain's query operand is q' (Tq = 1 for float64 rows)
        eb.Q = 0.0; eb.I = group16_max(am) / Tq;
    }

    bool live = inrange && !nan_row;   // NaN rows: every comparison below is false; they are answered separately
    bool ok = true;
    double knn = 0.0, cen = 0.0;
    uint32_t idx[3];
    double d2 = 0.0;
    const bool do_knn = (p.method & PHK_METHOD_KNN) && (todo & 1u), do_cen = (p.method & PHK_METHOD_KMEANS) && (todo & 2u);
    // wave-uniform skips: a segment nobody in the wave needs is not touched
    if (__any(do_knn)) {
        const bool okk = resolve_segment_g16(p, q, 0, (uint32_t)p.M, p.kn, qd, Tq, invT2, nqp2, eb, vs, p.R64, p.colnorm, false,
                                             live && do_knn, lane, lv[0], lix[0], lu[0], idx, d2);
        int votes = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) votes += (r < p.kn && idx[r] < p.M && p.labels[idx[r] < p.M ? idx[r] : 0u]) ? 1 : 0;
        knn = do_knn ? ((2 * votes > p.kn) ? 1.0 : -1.0) : 0.0;
        ok = okk || !do_knn;
    }
    if (__any(do_cen)) {
        double dp2 = 0.0, dn2 = 0.0;
        const bool ok1 = resolve_segment_g16(p, q, 1, (uint32_t)p.n_cpos, 1, qd, Tq, invT2, nqp2, eb, vs, p.C64, p.colnorm + p.M, true,
                                             live && ok && do_cen, lane, lv[1], lix[1], lu[1], idx, dp2);
        const bool ok2 = resolve_segment_g16(p, q, 2, (uint32_t)p.n_cneg, 1, qd, Tq, invT2, nqp2, eb, vs, p.C64 + p.n_cpos * FAST_D,
                                             p.colnorm + p.M + p.n_cpos, true, live && ok && ok1 && do_cen, lane, lv[2], lix[2],
                                             lu[2], idx, dn2);
        ok = ok && ((ok1 && ok2) || !do_cen);
        const double ep = sqrt(dp2), en = sqrt(dn2);
        cen = do_cen ? tanh((en - ep) / (ep + en)) : 0.0;  // scripts/phamer.py:206-209
    }
    // parts the sender had decided already are in scores[q]
    const double prev = (MODE == 1 && todo != 3u && inrange && !nan_row) ? p.scores[p.q_base + q] : 0.0;
    if (t == 0 && inrange) {
        if (nan_row) {  // zero-count contig: the reference's normalised row is NaN
            p.scores[p.q_base + q] = __builtin_nan("");
            if (p.status) atomicAdd(p.status, 1u);
        } else if (ok) {
            p.scores[p.q_base + q] = prev + knn + cen;  // scripts/phamer.py:313
        } else {
            const uint32_t slot = atomicAdd(p.fb_count, 1u);
            p.fb_list[slot] = (uint32_t)q;
        }
    }
}

// ------------------------------------------------------------------------------------
// 2e. the same decision for general D (k = 5, 6; phk_knn_f16_general_kernel with HI): one wave per query, as
//     phk_rerank_kernel.  Window, low products (float64, 4 DSUB dimensions per lane, 2 D bytes of lo16 per member),
//     count-exact margin test among the refined members; centroid leaders certified by their high-part margin and
//     given their exact distance.  A query it cannot decide goes to phk_rerank_kernel (listed), which works from the
//     same lists under the high-part error model with exact candidate distances.
// ------------------------------------------------------------------------------------
template <int DSUB>
__global__ __launch_bounds__(256) void phk_rerank_h_kernel(const uint32_t *__restrict__ counts, RerankParams p, HiParams hp) {
    constexpr int D = 256 * DSUB;
    const int lane = threadIdx.x & 63;
    const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (q >= p.N) return;
    const uint32_t *row = counts + q * D;
    uint4 c[DSUB];
    uint32_t s = 0;
#pragma unroll
    for (int sub = 0; sub < DSUB; ++sub) {
        c[sub] = reinterpret_cast<const uint4 *>(row + 256 * sub)[lane];
        s += c[sub].x + c[sub].y + c[sub].z + c[sub].w;
    }
    s = wave_sum(s);
    if (s == 0) {  // zero-count contig: the reference's normalised row is NaN
        if (lane == 0) {
            p.scores[p.q_base + q] = __builtin_nan("");
            if (p.status) atomicAdd(p.status, 1u);
        }
        return;
    }
    const double Tq = (double)s;
    double qd[4 * DSUB];   // normalised row (exact distances), then reused
    double aq = 0.0, ap = 0.0, sq = 0.0, qc2 = 0.0;
    uint32_t cmx = 0, cmn = 0xFFFFFFFFu;
    const double rcen = (double)phk_row_center(s, D);
#pragma unroll
    for (int sub = 0; sub < DSUB; ++sub) {
        const double x0 = (double)c[sub].x, x1 = (double)c[sub].y, x2 = (double)c[sub].z, x3 = (double)c[sub].w;
        sq = fma(x0, x0, fma(x1, x1, fma(x2, x2, fma(x3, x3, sq))));
        {   // norm of this 256-dimension chunk of c - c0 (see ErrBound, D > 256)
            const double y0 = x0 - rcen, y1 = x1 - rcen, y2 = x2 - rcen, y3 = x3 - rcen;
            qc2 = fmax(qc2, wave_sum(fma(y0, y0, fma(y1, y1, fma(y2, y2, y3 * y3)))));
        }
        cmx = max(max(cmx, max(c[sub].x, c[sub].y)), max(c[sub].z, c[sub].w));
        cmn = min(min(cmn, min(c[sub].x, c[sub].y)), min(c[sub].z, c[sub].w));
        qd[4 * sub + 0] = (double)c[sub].x / Tq;
        qd[4 * sub + 1] = (double)c[sub].y / Tq;
        qd[4 * sub + 2] = (double)c[sub].z / Tq;
        qd[4 * sub + 3] = (double)c[sub].w / Tq;
        const double2 m0 = reinterpret_cast<const double2 *>(p.mu64 + 256 * sub)[2 * lane];
        const double2 m1 = reinterpret_cast<const double2 *>(p.mu64 + 256 * sub)[2 * lane + 1];
        const double c0 = qd[4 * sub + 0] - m0.x, c1 = qd[4 * sub + 1] - m0.y, c2 = qd[4 * sub + 2] - m1.x, c3 = qd[4 * sub + 3] - m1.y;
        aq = fma(qd[4 * sub + 0], qd[4 * sub + 0], fma(qd[4 * sub + 1], qd[4 * sub + 1],
                 fma(qd[4 * sub + 2], qd[4 * sub + 2], fma(qd[4 * sub + 3], qd[4 * sub + 3], aq))));
        ap = fma(c0, c0, fma(c1, c1, fma(c2, c2, fma(c3, c3, ap))));
    }
    const double nq2 = wave_sum(aq), nqp2 = wave_sum(ap);
    const double vs = p.vscale / Tq;
    // the proposal's query operand is c - c0 (see phk_decide_h_kernel): Q, I of ErrBound, and the low product is taken
    // with c - T mu - (c0 - T/D)
    const CenteredOperand cop = phk_centered_operand(wave_sum(sq), Tq, wave_max((double)cmx), -wave_max(-(double)cmn), (double)D, p.eb_hsum);
    const double dlt = (double)phk_row_center(s, D) - Tq / (double)D;
    ErrBound eb;
    eb.A = sqrt(nq2) + p.mu_norm;
    eb.P = sqrt(nqp2);
    eb.Q = sqrt(qc2) / Tq * (1.0 + 1e-12); eb.I = cop.I; eb.habs = cop.habs;
    if (p.cand_a) eb.habs += p.eb_cAmax * 5.9604644775390625e-08 * (double)fmaxf(p.cand_a[q], p.cand_a[p.N + q]) * vs * 1.001;
    eb.cA = p.eb_cA; eb.cP = p.eb_cP; eb.cR = p.eb_cR; eb.cabs = p.eb_abs; eb.cQ = p.eb_cQ; eb.cI = p.eb_cI; eb.cIf = p.eb_cIf;
    const double nqp = eb.P;
    const double nqx = sqrt(nqp2 + cop.shift2);
    auto e_hi = [&](int sg, double R) { return nqx * phk_lam_of(hp, sg, R) + eb(R); };
    // low product of column `col` (global column index): sum_i (c_i - T mu_i - dlt) lo_i, the whole wave
    auto low_product = [&](uint64_t col) {
        const _Float16 *lr = hp.lo16 + col * D;
        double acc = 0.0;
#pragma unroll
        for (int sub = 0; sub < DSUB; ++sub) {
            const uint2 l = reinterpret_cast<const uint2 *>(lr + 256 * sub)[lane];
            const _Float16 *lh = reinterpret_cast<const _Float16 *>(&l);
            const double2 m0 = reinterpret_cast<const double2 *>(p.mu64 + 256 * sub)[2 * lane];
            const double2 m1 = reinterpret_cast<const double2 *>(p.mu64 + 256 * sub)[2 * lane + 1];
            acc = fma(fma(-Tq, m0.x, (double)c[sub].x) - dlt, (double)lh[0], acc);
            acc = fma(fma(-Tq, m0.y, (double)c[sub].y) - dlt, (double)lh[1], acc);
            acc = fma(fma(-Tq, m1.x, (double)c[sub].z) - dlt, (double)lh[2], acc);
            acc = fma(fma(-Tq, m1.y, (double)c[sub].w) - dlt, (double)lh[3], acc);
        }
        return wave_sum(acc);
    };
    // the 8 candidates of a segment sorted by high-part value (descending), on every lane
    auto sorted8 = [&](int seg, uint32_t ncols, float (&rv)[8], uint32_t (&ri)[8], double &U) {
        float v = -3.0e38f;
        uint32_t ix = 0xFFFFFFFFu;
        if (lane < 8) {
            const uint64_t o = cand_at(seg, lane >> 2, lane & 3, q, p.N);
            v = p.cand_v[o];
            ix = p.cand_i[o];
            if (ix >= ncols) v = -3.0e38f;
        }
        U = fmax((double)p.cand_u[candu_at(seg, 0, q, p.N)], (double)p.cand_u[candu_at(seg, 1, q, p.N)]) * vs;
        int rank = 0;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float ov = __shfl(v, m);
            rank += (lane < 8 && (ov > v || (ov == v && m < lane))) ? 1 : 0;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const unsigned long long bal = __ballot(lane < 8 && rank == r);
            const int srcl = __ffsll((long long)bal) - 1;
            rv[r] = __shfl(v, srcl);
            ri[r] = __shfl(ix, srcl);
        }
    };
    bool ok = true;
    double knn = 0.0, cen = 0.0;
    if (p.method & PHK_METHOD_KNN) {
        const int need = p.kn;
        float rv[8];
        uint32_t ri[8];
        double U;
        sorted8(0, (uint32_t)p.M, rv, ri, U);
        ok = ri[need - 1] < (uint32_t)p.M;
        if (ok) {
            const double eg = e_hi(0, p.rmax);
            const double d2up = fmax(nqp2 - 2.0 * ((double)rv[need - 1] * vs - eg), 0.0);
            const double R0 = fmin(p.rmax, (nqp + sqrt(d2up)) * (1.0 + 1e-6));
            bool near = true;
            for (int r = 0; r < need; ++r) near = near && p.colnorm[ri[r]] <= R0;
            const double eh = near ? e_hi(0, R0) : eg, e22 = near ? eb(R0) : eb(p.rmax);
            const double thr = (double)rv[need - 1] * vs - 2.0 * eh;
            int nw = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) nw += ((double)rv[r] * vs >= thr && ri[r] < (uint32_t)p.M) ? 1 : 0;
            ok = nw <= PHK_HI_REFINE && U < thr;
            if (ok) {
                double fv[PHK_HI_REFINE];
                uint32_t fl[PHK_HI_REFINE];
#pragma unroll
                for (int r = 0; r < PHK_HI_REFINE; ++r) {
                    fv[r] = -1.0e300;
                    fl[r] = 0;
                    if (r < nw) {   // wave-uniform
                        fv[r] = ((double)rv[r] + (nw > need ? low_product(ri[r]) : 0.0)) * vs;
                        fl[r] = p.labels[ri[r]] ? 1u : 0u;
                    }
                }
#pragma unroll
                for (int a = 1; a < PHK_HI_REFINE; ++a)
#pragma unroll
                    for (int b = a; b > 0; --b) {
                        const bool sw = fv[b] > fv[b - 1];
                        const double tv = fv[b]; const uint32_t tl = fl[b];
                        fv[b] = sw ? fv[b - 1] : fv[b]; fl[b] = sw ? fl[b - 1] : fl[b];
                        fv[b - 1] = sw ? tv : fv[b - 1]; fl[b - 1] = sw ? tl : fl[b - 1];
                    }
                const double hi_v = need == 1 ? fv[0] : need == 2 ? fv[1] : fv[2];
                const double lo_v = need == 1 ? fv[1] : need == 2 ? fv[2] : fv[3];
                ok = nw == need || hi_v - lo_v > 2.0 * e22;
                int votes = 0;
                for (int r = 0; r < need; ++r) votes += (int)fl[r];
                knn = (2 * votes > need) ? 1.0 : -1.0;
            }
        }
    }
    if (ok && (p.method & PHK_METHOD_KMEANS)) {
        double d2[2] = {0.0, 0.0};
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const uint32_t ncols = k2 == 0 ? (uint32_t)p.n_cpos : (uint32_t)p.n_cneg;
            float rv[8];
            uint32_t ri[8];
            double U;
            sorted8(1 + k2, ncols, rv, ri, U);
            bool good = ok && ri[0] < ncols;
            if (good) {
                const double eg = e_hi(1 + k2, p.rmax);
                const double d2up = fmax(nqp2 - 2.0 * ((double)rv[0] * vs - eg), 0.0);
                const double R0 = fmin(p.rmax, (nqp + sqrt(d2up)) * (1.0 + 1e-6));
                const double cn = p.colnorm[p.M + (k2 ? p.n_cpos : 0) + ri[0]];
                const double eh = cn <= R0 ? e_hi(1 + k2, R0) : eg;
                good = ((double)rv[0] - (double)rv[1]) * vs > 2.0 * eh;
                if (good) {   // (the canonical form of exact_d2: raw counts and their sum)
                    double qraw[4 * DSUB];
#pragma unroll
                    for (int sub = 0; sub < DSUB; ++sub) {
                        qraw[4 * sub + 0] = (double)c[sub].x; qraw[4 * sub + 1] = (double)c[sub].y;
                        qraw[4 * sub + 2] = (double)c[sub].z; qraw[4 * sub + 3] = (double)c[sub].w;
                    }
                    d2[k2] = exact_d2<DSUB>(qraw, Tq, 1.0 / (Tq * Tq), p.C64 + ((k2 ? p.n_cpos : 0) + (uint64_t)ri[0]) * D, lane);
                }
            }
            ok = good;
        }
        const double ep = sqrt(d2[0]), en = sqrt(d2[1]);
        cen = tanh((en - ep) / (ep + en));  // scripts/phamer.py:206-209
    }
    if (lane == 0) {
        if (ok) p.scores[p.q_base + q] = knn + cen;  // scripts/phamer.py:313
        else p.slow_list[atomicAdd(p.fb_count + 2, 1u)] = (uint32_t)q;
    }
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
int phk_launch_rerank_wave(phk_ctx *ctx, int src_kind, int dsub, bool i8h, unsigned blocks, const void *src, const RerankParams &p) {
    PHK_REQUIRE(!i8h || (src_kind == 0 && dsub > 1), "phk_launch_rerank_wave: the int8 lists belong to count rows at D >= 512");
#define PHK_RW(S, DS, I8) PHK_LAUNCH(ctx, "phk_rerank_kernel", (phk_rerank_kernel<S, DS, I8><<<dim3(blocks), dim3(256), 0, ctx->stream>>>(src, p)))
#define PHK_RW_DS(DS)                                      \
    do {                                                   \
        if (i8h) { PHK_RW(0, DS, true); }                  \
        else if (src_kind == 0) { PHK_RW(0, DS, false); }  \
        else { PHK_RW(1, DS, false); }                     \
    } while (0)
    switch (dsub) {
        case 1:
            if (src_kind == 0) { PHK_RW(0, 1, false); } else { PHK_RW(1, 1, false); }
            break;
        case 2: PHK_RW_DS(2); break;
        case 4: PHK_RW_DS(4); break;
        case 8: PHK_RW_DS(8); break;
        case 16: PHK_RW_DS(16); break;
        default: phk_set_error("phk_launch_rerank_wave: D = %d", 256 * dsub); return PHK_ERR_UNSUPPORTED;
    }
#undef PHK_RW_DS
#undef PHK_RW
    return PHK_OK;
}

// mode: the kernel's MODE (0 every query, 1 the listed ones, 2 the second chance's dense rows -- count rows only)
int phk_launch_rerank16(phk_ctx *ctx, int src_kind, int mode, unsigned blocks, const void *src, const RerankParams &p) {
    PHK_REQUIRE(mode >= 0 && mode <= 2 && (mode != 2 || src_kind == 0), "phk_launch_rerank16: mode");
#define PHK_R16(S, MODE) PHK_LAUNCH(ctx, "phk_rerank16_kernel", (phk_rerank16_kernel<S, MODE><<<dim3(blocks), dim3(256), 0, ctx->stream>>>(src, p)))
    if (mode == 2) { PHK_R16(0, 2); }
    else if (mode == 1) {
        if (src_kind == 0) { PHK_R16(0, 1); } else { PHK_R16(1, 1); }
    } else {
        if (src_kind == 0) { PHK_R16(0, 0); } else { PHK_R16(1, 0); }
    }
#undef PHK_R16
    return PHK_OK;
}

int phk_launch_rerank_h(phk_ctx *ctx, int dsub, unsigned blocks, const uint32_t *counts, const RerankParams &p, const HiParams &hp) {
#define PHK_RH(DS) PHK_LAUNCH(ctx, "phk_rerank_h_kernel", (phk_rerank_h_kernel<DS><<<dim3(blocks), dim3(256), 0, ctx->stream>>>(counts, p, hp)))
    switch (dsub) {
        case 2: PHK_RH(2); break;
        case 4: PHK_RH(4); break;
        case 8: PHK_RH(8); break;
        case 16: PHK_RH(16); break;
        default: phk_set_error("phk_launch_rerank_h: D = %d", 256 * dsub); return PHK_ERR_UNSUPPORTED;
    }
#undef PHK_RH
    return PHK_OK;
}
