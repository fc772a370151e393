// score_fallback.hip -- the exact float64 brute force for queued queries (what no certificate covers ends here), and the
// small kernels around the hand-over queues.  Shared device code: score_decide.h.
#include "score_decide.h"

// ------------------------------------------------------------------------------------
// 3. exact brute force for queued queries.  Work item = (queued query, column chunk): a block
//    computes the direct-difference float64 distances of its chunk (one thread per column), then
//    reduces them to a partial record (3 nearest train columns of the chunk + nearest positive /
//    negative centroid of the chunk).  A second kernel merges the FB_CHUNKS records of a query.
// ------------------------------------------------------------------------------------
template <int SRC>
__global__ __launch_bounds__(256) void phk_fallback_partial_kernel(const void *__restrict__ src, RerankParams p) {
    extern __shared__ double fb_lds[];  // [0, 256): the query; then one chunk of distances
    const uint64_t D = p.D;
    double *fb_q = fb_lds, *fb_dist = fb_lds + D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t count = phk_uniform_load(p.fb_count);
    const uint64_t ncols = p.M + p.n_cpos + p.n_cneg;
    const uint64_t nch = fb_group_chunks(count, p.fb_rec_cap);
    const uint64_t cw = (ncols + nch - 1) / nch;
    FbRecord *rec = static_cast<FbRecord *>(p.fb_rec);
    const uint64_t items = (uint64_t)count * nch;
    for (uint64_t it = blockIdx.x; it < items; it += gridDim.x) {
        const uint64_t qi = it / nch, ch = it % nch;
        const uint64_t q = p.fb_list[qi];
        const uint64_t c0 = ch * cw < ncols ? ch * cw : ncols, c1 = (c0 + cw < ncols) ? c0 + cw : ncols;
        if (c0 >= c1) {   // (more chunks than columns: an empty record)
            if (threadIdx.x == 0) {
                FbRecord e;
                for (int k = 0; k < 3; ++k) { e.d[k] = INFINITY; e.i[k] = 0xFFFFFFFFu; }
                e.minpos = e.minneg = INFINITY; e.pad = 0;
                rec[it] = e;
            }
            continue;
        }
        // The distances are evaluated in the SAME float64 form, element ownership and summation order as every other
        // exact evaluation of this model shape, so that a query's score does not depend on the route that decided it
        // (which depends on how many rows its batch queued): D = 256 -- exact_d2_g16 (raw counts c and the row sum T,
        // sum (c_i - T r_i)^2 / T^2, 16 lanes per column in G16 ownership).  (Other D: phk_fallback_group_kernel; the branch
        // below for them is not reached by phk_score_fast.)
        const bool g16 = D == FAST_D;
        double Tq = 1.0, invT2 = 1.0;
        if (SRC == 0) {
            const uint32_t *row = static_cast<const uint32_t *>(src) + q * D;
            uint32_t s = 0;
            for (uint64_t d = lane; d < D; d += 64) s += row[d];  // every wave sums the whole row
            s = wave_sum(s);
            if (g16) {
                Tq = (double)s;
                invT2 = 1.0 / (Tq * Tq);
                for (uint64_t d = threadIdx.x; d < D; d += 256) fb_q[d] = (double)row[d];
            } else {
                for (uint64_t d = threadIdx.x; d < D; d += 256) fb_q[d] = (double)row[d] / (double)s;
            }
        } else {
            for (uint64_t d = threadIdx.x; d < D; d += 256) fb_q[d] = static_cast<const double *>(src)[q * D + d];
        }
        __syncthreads();
        if (g16) {
            // 16 lanes per column (contiguous 256-byte pieces of its row per load), 16 columns per pass
            for (uint64_t cb = c0; cb < c1; cb += 16) {
                const uint64_t c = cb + (threadIdx.x >> 4);
                const int t16 = threadIdx.x & 15;
                const uint64_t cc = c < c1 ? c : c1 - 1;
                const double2 *row = reinterpret_cast<const double2 *>(cc < p.M ? p.R64 + cc * D : p.C64 + (cc - p.M) * D) + t16;
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double2 r = row[16 * i];
                    const double d0 = fma(-Tq, r.x, fb_q[32 * i + 2 * t16]), d1 = fma(-Tq, r.y, fb_q[32 * i + 2 * t16 + 1]);
                    acc = fma(d0, d0, fma(d1, d1, acc));
                }
                acc = group16_sum(acc) * invT2;
                if (t16 == 0 && c < c1) fb_dist[c - c0] = (p.col_mask && c < p.M && p.col_mask[c]) ? INFINITY : acc;
            }
        } else {
            // one wave per column, 4 columns per pass
            for (uint64_t cb = c0; cb < c1; cb += 4) {
                const uint64_t c = cb + wave;
                const uint64_t cc = c < c1 ? c : c1 - 1;
                const double *row = cc < p.M ? p.R64 + cc * D : p.C64 + (cc - p.M) * D;
                double acc = 0.0;
                for (uint64_t sub = 0; sub < D / 256; ++sub) {
                    const double2 a = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane];
                    const double2 b = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane + 1];
                    const double *qv = fb_q + 256 * sub + 4 * lane;
                    const double d0 = qv[0] - a.x, d1 = qv[1] - a.y, d2 = qv[2] - b.x, d3 = qv[3] - b.y;
                    acc = fma(d0, d0, fma(d1, d1, fma(d2, d2, fma(d3, d3, acc))));
                }
                acc = wave_sum(acc);
                if (lane == 0 && c < c1) fb_dist[c - c0] = (p.col_mask && c < p.M && p.col_mask[c]) ? INFINITY : acc;
            }
        }
        __syncthreads();
        if (wave == 0) {
            FbRecord r;
            double last_d = -1.0;
            uint64_t last_i = 0;
            bool first = true;
            for (int k = 0; k < 3; ++k) {  // (distance, index)-ordered selection among train columns
                double bd = INFINITY;
                uint64_t bi = ~0ull;
                for (uint64_t c = c0 + lane; c < c1 && c < p.M; c += 64) {
                    const double d = fb_dist[c - c0];
                    const bool after = first || fb_less(last_d, last_i, d, c);
                    if (after && fb_less(d, c, bd, bi)) { bd = d; bi = c; }
                }
#pragma unroll
                for (int sft = 32; sft > 0; sft >>= 1) {
                    const double od = __shfl_xor(bd, sft);
                    const uint64_t oi = __shfl_xor(bi, sft);
                    if (fb_less(od, oi, bd, bi)) { bd = od; bi = oi; }
                }
                r.d[k] = bd;
                r.i[k] = (uint32_t)bi;  // 0xFFFFFFFF when the chunk has fewer train columns
                last_d = bd; last_i = bi; first = false;
            }
            double bp = INFINITY, bn = INFINITY;
            for (uint64_t c = c0 + lane; c < c1; c += 64) {
                if (c >= p.M && c < p.M + p.n_cpos) bp = fmin(bp, fb_dist[c - c0]);
                if (c >= p.M + p.n_cpos) bn = fmin(bn, fb_dist[c - c0]);
            }
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) {
                bp = fmin(bp, __shfl_xor(bp, sft));
                bn = fmin(bn, __shfl_xor(bn, sft));
            }
            r.minpos = bp; r.minneg = bn; r.pad = 0;
            if (lane == 0) rec[it] = r;
        }
        __syncthreads();
    }
}

// General D: the same brute force with the queued queries taken EIGHT at a time.  A queued query of the one-query kernel
// above streams the whole float64 reference through its CU (50 000 x 32 KiB at configs[4]: 1.6 GB per query, 22 ms for
// a hundred queries); here a workgroup of 8 waves holds 8 queries -- one per wave, the normalised row in registers -- and
// all of them meet every column of the item's chunk while it passes through the caches once.  Each wave evaluates its
// query exactly as phk_rerank_kernel does (exact_d2<DSUB>: same operands, element ownership and summation order), so
// a score does not depend on the route that produced it.  Items are numbered chunk-major: the workgroups that run
// together share a chunk of the reference.
template <int SRC, int DSUB>
__global__ __launch_bounds__(512) void phk_fallback_group_kernel(const void *__restrict__ src, RerankParams p) {
    constexpr int D = 256 * DSUB;
    // two reference rows (float64) in LDS: the row every wave works on and the next one on its way in.  Read straight from
    // memory by eight waves a row crossed the CU's 32 KiB L1 eight times (184 GB of L2 -> L1 traffic for a hundred queries
    // at configs[4]); through LDS it crosses once.
    extern __shared__ __attribute__((aligned(16))) uint8_t fbg_lds[];   // 2 x 8 D bytes
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t count = phk_uniform_load(p.fb_count);
    const uint64_t ncols = p.M + p.n_cpos + p.n_cneg;
    const uint64_t nch = fb_group_chunks(count, p.fb_rec_cap);
    const uint64_t cw = (ncols + nch - 1) / nch;
    const uint64_t ngroups = ((uint64_t)count + 7) / 8;
    FbRecord *rec = static_cast<FbRecord *>(p.fb_rec);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)fbg_lds;
    // row `c` -> buffer `buf`: wave w moves the 1 KiB pieces w, w + 8, ..  (LDS-DMA: lane l's 16 bytes land at piece + 16 l)
    auto stage = [&](uint64_t c, int buf) {
        const uint8_t *row = reinterpret_cast<const uint8_t *>(c < p.M ? p.R64 + c * D : p.C64 + (c - p.M) * D);
#pragma unroll
        for (int pc = 0; pc < (D * 8) / 8192 + 1; ++pc) {
            const uint32_t piece = (uint32_t)wave + 8u * (uint32_t)pc;
            if (piece * 1024u < (uint32_t)(D * 8)) {
                const uint8_t *gp = row + piece * 1024u + (uint32_t)lane * 16u;
                const uint32_t lp = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)buf * (uint32_t)(D * 8) + piece * 1024u);
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(lp) : "memory");
            }
        }
    };
    for (uint64_t it = blockIdx.x; it < ngroups * nch; it += gridDim.x) {   // (uniform over the workgroup)
        const uint64_t ch = it / ngroups, g8 = (it % ngroups) * 8;
        const uint64_t qi = g8 + (uint64_t)wave;
        const bool active = qi < count;                 // a group's last waves may have no query: they keep the barriers
        const uint64_t q = p.fb_list[active ? qi : g8];
        const uint64_t c0 = ch * cw < ncols ? ch * cw : ncols, c1 = (c0 + cw < ncols) ? c0 + cw : ncols;
        if (c0 >= c1) {   // (more chunks than columns)
            if (active && lane == 0) {
                FbRecord e;
                for (int k = 0; k < 3; ++k) { e.d[k] = INFINITY; e.i[k] = 0xFFFFFFFFu; }
                e.minpos = e.minneg = INFINITY; e.pad = 0;
                rec[qi * nch + ch] = e;
            }
            continue;
        }
        stage(c0, 0);
        double qd[4 * DSUB];
        double Tq = 1.0, invT2 = 1.0;
        if (SRC == 0) {
            const uint32_t *row = static_cast<const uint32_t *>(src) + q * D;
            uint4 c[DSUB];
            uint32_t sm = 0;
#pragma unroll
            for (int sub = 0; sub < DSUB; ++sub) {
                c[sub] = reinterpret_cast<const uint4 *>(row + 256 * sub)[lane];
                sm += c[sub].x + c[sub].y + c[sub].z + c[sub].w;
            }
            const double ds = (double)wave_sum(sm);
            Tq = ds;
            invT2 = 1.0 / (ds * ds);
#pragma unroll
            for (int sub = 0; sub < DSUB; ++sub) {   // the counts themselves: see exact_d2
                qd[4 * sub + 0] = (double)c[sub].x; qd[4 * sub + 1] = (double)c[sub].y;
                qd[4 * sub + 2] = (double)c[sub].z; qd[4 * sub + 3] = (double)c[sub].w;
            }
        } else {
            const double *row = static_cast<const double *>(src) + q * D;
#pragma unroll
            for (int sub = 0; sub < DSUB; ++sub) {
                const double2 a = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane];
                const double2 b = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane + 1];
                qd[4 * sub + 0] = a.x; qd[4 * sub + 1] = a.y; qd[4 * sub + 2] = b.x; qd[4 * sub + 3] = b.y;
            }
        }
        FbRecord r;
#pragma unroll
        for (int k = 0; k < 3; ++k) { r.d[k] = INFINITY; r.i[k] = 0xFFFFFFFFu; }
        r.minpos = r.minneg = INFINITY;
        r.pad = 0;
        for (uint64_t c = c0; c < c1; ++c) {
            const int buf = (int)((c - c0) & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of row c have landed ...
            __syncthreads();                                   // ... and everybody's; everybody is done with the other buffer
            if (c + 1 < c1) stage(c + 1, buf ^ 1);
            // exact_d2<DSUB> with the row read from LDS: same operands, element ownership and summation order
            const double *row = reinterpret_cast<const double *>(fbg_lds + (size_t)buf * (D * 8));
            double acc = 0.0;
#pragma unroll
            for (int sub = 0; sub < DSUB; ++sub) {
                const double2 a = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane];
                const double2 b = reinterpret_cast<const double2 *>(row + 256 * sub)[2 * lane + 1];
                const double d0 = fma(-Tq, a.x, qd[4 * sub + 0]), d1 = fma(-Tq, a.y, qd[4 * sub + 1]);
                const double d2 = fma(-Tq, b.x, qd[4 * sub + 2]), d3 = fma(-Tq, b.y, qd[4 * sub + 3]);
                acc = fma(d0, d0, fma(d1, d1, fma(d2, d2, fma(d3, d3, acc))));
            }
            double dist = wave_sum(acc) * invT2;   // the same value on every lane
            if (c < p.M) {
                if (p.col_mask && p.col_mask[c]) dist = INFINITY;
                double d = dist;
                uint64_t ix = c;
#pragma unroll
                for (int k = 0; k < 3; ++k) {   // (distance, index)-ordered
                    const uint64_t cur = r.i[k] == 0xFFFFFFFFu ? ~0ull : (uint64_t)r.i[k];
                    if (fb_less(d, ix, r.d[k], cur)) {
                        const double td = r.d[k];
                        r.d[k] = d; r.i[k] = (uint32_t)ix; d = td; ix = cur;
                    }
                }
            } else if (c < p.M + p.n_cpos) {
                r.minpos = fmin(r.minpos, dist);
            } else {
                r.minneg = fmin(r.minneg, dist);
            }
        }
        __syncthreads();   // the last row is read: the next item's first row may overwrite buffer 0
        if (active && lane == 0) rec[qi * nch + ch] = r;
    }
}

// one thread per queued query: merge its FB_CHUNKS partial records and emit the score
__global__ __launch_bounds__(256) void phk_fallback_merge_kernel(RerankParams p) {
    const uint32_t count = *p.fb_count;
    if (blockIdx.x == 0 && p.stat_total) {   // statistics: this batch's counters into the call's totals, one word per thread
        // (atomic: the striped copies are added by another workgroup of this launch)
        const uint32_t t = threadIdx.x;
        uint32_t word = 0xFFFFFFFFu, add = 0;
        if (t == 0) { word = 0; add = p.fb_count[0]; }
        else if (t == 1) { word = 1; add = p.fb_count[1] + (p.exact_extra ? *p.exact_extra : 0u); }
        else if (t == 2 && p.map_count) { word = 2; add = *p.map_count; }          // queries that took the second chance
        else if (t == 3 && p.q2_count) { word = 2; add = p.q2_count[0] + p.q2_count[1]; }   // general D: rows re-swept with three digits / swept
        else if (t == 4 && p.q2_count) { word = 7; add = p.q2_count[0]; }          // by the f16 kernel (both are second chances)
        else if (t == 5 && p.q2_count) { word = 8; add = p.q2_count[1]; }
        else if (t >= 8 && t < 12 && p.counters) { word = 3 + (t - 8); add = p.counters[t]; }   // why the high-parts-only decision stage passed them on
        if (word != 0xFFFFFFFFu && add) atomicAdd(p.stat_total + word, add);
    }
    if (blockIdx.x == 1 && p.stat_total && p.stripes) {          // ... and the striped copies of the same words: summed in the
        // workgroup, one atomic per word and wave (one per stripe and word queued 1 280 atomics on five addresses)
        uint32_t acc[5] = {0, 0, 0, 0, 0};
        for (uint32_t sidx = threadIdx.x; sidx < PHK_STRIPES; sidx += blockDim.x) {
            const uint32_t *w = p.stripes + sidx * 32u;
            acc[0] += w[1];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[1 + i] += w[8 + i];
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const uint32_t tot = wave_sum(acc[i]);
            if ((threadIdx.x & 63) == 0 && tot) atomicAdd(p.stat_total + (i == 0 ? 1 : 2 + i), tot);
        }
    }
    // One WAVE per queued query, lane = chunk record (round 5: one thread walked the query's 64 records in turn -- a chain of
    // 64 dependent-looking loads, 40 of the kernel's 45 us for the usual two rows of a batch).  The three nearest train
    // columns under the total order (distance, column index) -- column indices are unique, so the result does not depend on
    // the order of the merge -- by three rounds of a wave-wide minimum whose owner pops its head.
    const FbRecord *rec = static_cast<const FbRecord *>(p.fb_rec);
    const uint32_t nch = fb_group_chunks(count, p.fb_rec_cap);   // 64 or FB_CHUNKS: at most one record per lane
    const int lane = threadIdx.x & 63;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t qi = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; qi < count; qi += nwaves) {
        const uint64_t q = p.fb_list[qi];
        double d[3] = {INFINITY, INFINITY, INFINITY};
        uint64_t ix[3] = {~0ull, ~0ull, ~0ull};
        double bp = INFINITY, bn = INFINITY;
        if ((uint32_t)lane < nch) {
            const FbRecord r = rec[qi * nch + (uint32_t)lane];
#pragma unroll
            for (int k = 0; k < 3; ++k) {   // the record's entries into the lane's sorted triple (absent entries: index 2^32 - 1)
                if (r.i[k] == 0xFFFFFFFFu) continue;
                double dk = r.d[k];
                uint64_t ck = r.i[k];
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3)
                    if (fb_less(dk, ck, d[s3], ix[s3])) {
                        const double td = d[s3]; const uint64_t ti = ix[s3];
                        d[s3] = dk; ix[s3] = ck; dk = td; ck = ti;
                    }
            }
            bp = r.minpos;
            bn = r.minneg;
        }
        double bd[3];
        uint64_t bi[3];
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            double md = d[0];
            uint64_t mi = ix[0];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double od = __shfl_xor(md, off);
                const uint64_t oi = __shfl_xor(mi, off);
                if (fb_less(od, oi, md, mi)) { md = od; mi = oi; }
            }
            bd[s3] = md;
            bi[s3] = mi;
            if (mi != ~0ull && ix[0] == mi) {   // this lane's head won: pop it
                d[0] = d[1]; ix[0] = ix[1];
                d[1] = d[2]; ix[1] = ix[2];
                d[2] = INFINITY; ix[2] = ~0ull;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            bp = fmin(bp, __shfl_xor(bp, off));
            bn = fmin(bn, __shfl_xor(bn, off));
        }
        if (lane == 0) {
            double knn = 0.0, cen = 0.0;
            if (p.method & PHK_METHOD_KNN) {
                int votes = 0;
                for (int k = 0; k < p.kn; ++k) votes += p.labels[bi[k]] ? 1 : 0;
                knn = (2 * votes > p.kn) ? 1.0 : -1.0;
            }
            if (p.method & PHK_METHOD_KMEANS) {
                const double ep = sqrt(bp), en = sqrt(bn);
                cen = tanh((en - ep) / (ep + en));
            }
            p.scores[p.q_base + q] = knn + cen;
        }
        (void)bd;
    }
    // the last workgroup out zeroes the set's control words: every workgroup's reads of them precede its ticket
    if (p.clean_counters) {
        __shared__ uint32_t s_last;
        // (this workgroup's loads of the words have returned -- their values are in registers -- once vmcnt is 0: no fence, which
        // would also wait for the atomics above to be performed, 3.5 us)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(p.clean_counters + 15, 1u) == gridDim.x - 1 ? 1u : 0u;
        __syncthreads();
        if (s_last) {
            for (uint32_t i = threadIdx.x; i < 16u; i += blockDim.x) p.clean_counters[i] = 0;
            if (p.clean_stripes)
                for (uint32_t i = threadIdx.x; i < PHK_STRIPES * 32u; i += blockDim.x) p.clean_stripes[i] = 0;
        }
    }
}

// one thread per query: score += tanh((en - ep) / (ep + en)) for the queries whose centroid distances the general-D
// decision kernel left in `pend` (the rest holds the NaN fill); scripts/phamer.py:206-209, 313
__global__ __launch_bounds__(256) void phk_finish_cen_kernel(uint64_t N, const double *__restrict__ pend, double *__restrict__ scores) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N) return;
    const double dp2 = pend[2 * q], dn2 = pend[2 * q + 1];
    if (!(dp2 >= 0.0)) return;
    const double ep = sqrt(dp2), en = sqrt(dn2);
    scores[q] += tanh((en - ep) / (ep + en));
}

// rows list[0 .. n) of a count matrix -> a dense matrix (+ their row sums): the sub-batch of a second pass.  One wave per row.
__global__ __launch_bounds__(256) void phk_gather_list_rows_kernel(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ rowsum,
                                                              const uint32_t *__restrict__ list, uint64_t n, uint64_t D,
                                                              uint32_t *__restrict__ out, uint32_t *__restrict__ out_sum) {
    const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n) return;
    const uint64_t r = list[w];
    const uint4 *src = reinterpret_cast<const uint4 *>(counts + r * D);
    uint4 *dst = reinterpret_cast<uint4 *>(out + w * D);
    for (uint64_t i = lane; i < D / 4; i += 64) dst[i] = src[i];
    if (lane == 0 && rowsum) out_sum[w] = rowsum[r];
}

// a short hand-over queue goes straight to the brute force: its rows are appended to that queue
__global__ __launch_bounds__(256) void phk_append_queue_kernel(const uint32_t *__restrict__ list, uint32_t n, uint32_t *__restrict__ fb_list,
                                                               uint32_t *__restrict__ fb_count, uint32_t *__restrict__ q_count) {
    __shared__ uint32_t base;
    if (threadIdx.x == 0) {
        base = atomicAdd(fb_count, n);
        *q_count = 0;   // (statistics: these rows are brute-forced, not swept again)
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) fb_list[base + i] = list[i];
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
int phk_launch_fallback_partial(phk_ctx *ctx, int src_kind, size_t lds, const void *src, const RerankParams &p) {
    const dim3 grid((unsigned)ctx->num_cus * 2);
    PHK_REQUIRE(lds <= FB_LDS_MAX, "phk_launch_fallback_partial: LDS request beyond the kernel's attribute");
    if (src_kind == 0) { PHK_LAUNCH(ctx, "phk_fallback_partial_kernel", phk_fallback_partial_kernel<0><<<grid, dim3(256), lds, ctx->stream>>>(src, p)); }
    else { PHK_LAUNCH(ctx, "phk_fallback_partial_kernel", phk_fallback_partial_kernel<1><<<grid, dim3(256), lds, ctx->stream>>>(src, p)); }
    return PHK_OK;
}

// general D: the queued queries eight at a time against a chunk of the reference (see the kernel)
int phk_launch_fallback_group(phk_ctx *ctx, int src_kind, int dsub, const void *src, const RerankParams &p) {
    const dim3 fg((unsigned)ctx->num_cus * 2), fbk(512);
    const size_t fgl = 2 * (size_t)256 * dsub * sizeof(double);   // <= 64 KiB (D <= 4096)
#define PHK_FBG(DS)                                                                                                                                     \
    do {                                                                                                                                                \
        if (src_kind == 0) { PHK_LAUNCH(ctx, "phk_fallback_group_kernel", (phk_fallback_group_kernel<0, DS><<<fg, fbk, fgl, ctx->stream>>>(src, p))); } \
        else { PHK_LAUNCH(ctx, "phk_fallback_group_kernel", (phk_fallback_group_kernel<1, DS><<<fg, fbk, fgl, ctx->stream>>>(src, p))); }               \
    } while (0)
    switch (dsub) {
        case 2: PHK_FBG(2); break;
        case 4: PHK_FBG(4); break;
        case 8: PHK_FBG(8); break;
        case 16: PHK_FBG(16); break;
        default: phk_set_error("phk_launch_fallback_group: D = %d", 256 * dsub); return PHK_ERR_UNSUPPORTED;
    }
#undef PHK_FBG
    return PHK_OK;
}

int phk_launch_fallback_merge(phk_ctx *ctx, const RerankParams &p) {
    PHK_LAUNCH(ctx, "phk_fallback_merge_kernel", phk_fallback_merge_kernel<<<dim3(64), dim3(256), 0, ctx->stream>>>(p));
    return PHK_OK;
}

int phk_launch_finish_cen(phk_ctx *ctx, uint64_t nb, const double *pend, double *scores) {
    PHK_LAUNCH(ctx, "phk_finish_cen_kernel",
               phk_finish_cen_kernel<<<dim3((unsigned)phk_div_up(nb, 256)), dim3(256), 0, ctx->stream>>>(nb, pend, scores));
    return PHK_OK;
}

int phk_launch_gather_list_rows(phk_ctx *ctx, const uint32_t *counts, const uint32_t *rowsum, const uint32_t *list, uint64_t n, uint64_t D,
                                uint32_t *out, uint32_t *out_sum) {
    PHK_LAUNCH(ctx, "phk_gather_list_rows_kernel",
               phk_gather_list_rows_kernel<<<dim3((unsigned)phk_div_up(n, 4)), dim3(256), 0, ctx->stream>>>(counts, rowsum, list, n, D, out, out_sum));
    return PHK_OK;
}

int phk_launch_append_queue(phk_ctx *ctx, const uint32_t *list, uint32_t n, uint32_t *fb_list, uint32_t *fb_count, uint32_t *q_count) {
    PHK_LAUNCH(ctx, "phk_append_queue_kernel", phk_append_queue_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(list, n, fb_list, fb_count, q_count));
    return PHK_OK;
}

// per-device kernel attributes (phk_score_mfma_init_device, called from phk_create)
int phk_score_fallback_init_device(phk_ctx *ctx) {
    (void)ctx;
    PHK_HIP(hipFuncSetAttribute((const void *)phk_fallback_partial_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB_LDS_MAX));
    PHK_HIP(hipFuncSetAttribute((const void *)phk_fallback_partial_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB_LDS_MAX));
    return PHK_OK;
}
