// count.hip -- ASCII -> 2-bit packer, per-contig 4^k histogram, widening and row
// normalisation kernels (gfx950).
//
// Replaces the arithmetic of kmer.sequence_to_integers (scripts/kmer.py:183-196),
// kmer.count_string's window loop (scripts/kmer.py:47-50) and kmer.normalize_counts
// (scripts/kmer.py:209-221).  Packed-stream layout: include/phamers_hip.h.
#include <stdlib.h>

#include "phk_common.h"

// ------------------------------------------------------------------------------------
// pack: one thread per 32 bases -> two packed words + one mask word
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int phk_code_of(uint32_t ch, uint32_t sym) {
    // sym = symbols4 packed little-endian: code i <-> byte i.  Case-sensitive exact match
    // (scripts/kmer.py:190-191: every character outside `symbols` is a no-read).
    int code = -1;
    code = (ch == (sym & 0xFF)) ? 0 : code;
    code = (ch == ((sym >> 8) & 0xFF)) ? 1 : code;
    code = (ch == ((sym >> 16) & 0xFF)) ? 2 : code;
    code = (ch == (sym >> 24)) ? 3 : code;
    return code;
}

__global__ __launch_bounds__(256) void phk_pack_kernel(const uint8_t *__restrict__ bases, uint64_t T,
                                                       uint32_t sym, uint32_t *__restrict__ packed,
                                                       uint32_t *__restrict__ mask,
                                                       uint64_t packed_words, uint64_t mask_words,
                                                       uint32_t *any_invalid, uint64_t block0) {
    uint64_t t = (block0 + blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= mask_words) return;
    uint64_t g0 = t * 32;
    uint32_t w[2] = {0u, 0u};
    uint32_t m = 0u;
    bool bad = false;
    if (g0 + 32 <= T) {
        const uint4 *p = reinterpret_cast<const uint4 *>(bases + g0);  // 32-byte aligned
        uint4 v[2] = {p[0], p[1]};
        const uint32_t *d = reinterpret_cast<const uint32_t *>(v);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            uint32_t ch = (d[i >> 2] >> (8 * (i & 3))) & 0xFF;
            int code = phk_code_of(ch, sym);
            bool ok = code >= 0;
            bad |= !ok;
            w[i >> 4] |= (uint32_t)(ok ? code : 0) << (30 - 2 * (i & 15));
            m |= (uint32_t)ok << (31 - i);
        }
    } else {
        for (int i = 0; i < 32; ++i) {
            uint64_t g = g0 + i;
            if (g < T) {
                int code = phk_code_of(bases[g], sym);
                bool ok = code >= 0;
                bad |= !ok;
                w[i >> 4] |= (uint32_t)(ok ? code : 0) << (30 - 2 * (i & 15));
                m |= (uint32_t)ok << (31 - i);
            }
        }
    }
    mask[t] = m;
    if (2 * t < packed_words) packed[2 * t] = w[0];
    if (2 * t + 1 < packed_words) packed[2 * t + 1] = w[1];
    if (bad && any_invalid) atomicOr(any_invalid, 1u);
}

int phk_launch_pack(phk_ctx *ctx, const char *d_bases, uint64_t T, const char *symbols4,
                    uint32_t *d_packed, uint32_t *d_mask, uint32_t *d_any_invalid) {
    PHK_REQUIRE(d_packed && d_mask, "phk_pack: packed and mask outputs are required");
    PHK_REQUIRE(T == 0 || d_bases, "phk_pack: bases is NULL");
    PHK_REQUIRE(((uintptr_t)d_bases & 15) == 0, "phk_pack: bases must be 16-byte aligned");
    uint32_t sym = (uint32_t)(uint8_t)symbols4[0] | ((uint32_t)(uint8_t)symbols4[1] << 8) |
                   ((uint32_t)(uint8_t)symbols4[2] << 16) | ((uint32_t)(uint8_t)symbols4[3] << 24);
    uint64_t packed_words = phk_div_up(T, 16) + 1, mask_words = phk_div_up(T, 32) + 1;
    if (d_any_invalid) PHK_HIP(hipMemsetAsync(d_any_invalid, 0, sizeof(uint32_t), ctx->stream));
    PHK_LAUNCH_SLICED(ctx, "phk_pack_kernel", phk_div_up(mask_words, 256), b0, nblk,
                      phk_pack_kernel<<<dim3(nblk), dim3(256), 0, ctx->stream>>>(
                          (const uint8_t *)d_bases, T, sym, d_packed, d_mask, packed_words, mask_words, d_any_invalid, b0));
    return PHK_OK;
}

// ------------------------------------------------------------------------------------
// de-line + pack (round 5): the packer fed by the RAW bytes of a FASTA file.  The host no longer copies every sequence
// line into the upload's staging buffers (0.25 s per 5 GB on 16 cores, the largest piece of the command line's load):
// the file's bytes go up as they are, and this kernel finds base i of record r at
//     raw[ begin[r] + (i / lw[r]) * (lw[r] + tl[r]) + i % lw[r] ]
// -- lw = bases per line, tl = bytes between a line's last base and the next line's first (trailing white space + '\n'),
// both fixed within a record: what the host's index scan checks line by line anyway.  A record it finds irregular
// (lines of different widths, a blank or a '\r' inside a line, a record cut by the scan's thread slices) is written
// de-lined by the host into a side buffer and described as one line of it (lw = 2^32 - 1: the kernel then reads `side`).
// One thread per 32 output bases: its record by binary search in the offsets, then a walk (position, column).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void phk_deline_pack_kernel(const uint8_t *__restrict__ raw, const uint8_t *__restrict__ side,
                                                              const uint64_t *__restrict__ offsets,
                                                              uint64_t n, const uint64_t *__restrict__ rbegin,
                                                              const uint32_t *__restrict__ rlw, const uint32_t *__restrict__ rtl,
                                                              uint64_t T, uint32_t sym, uint32_t *__restrict__ packed,
                                                              uint32_t *__restrict__ mask, uint64_t packed_words,
                                                              uint64_t mask_words, uint32_t *any_invalid, uint64_t block0) {
    const uint64_t t = (block0 + blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= mask_words) return;
    const uint64_t g0 = t * 32;
    uint32_t w[2] = {0u, 0u};
    uint32_t m = 0u;
    bool bad = false;
    if (g0 < T) {
        // record of base g0: the last r with offsets[r] <= g0 (empty records before it are passed)
        uint64_t lo = 0, hi = n;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (offsets[mid] <= g0) lo = mid; else hi = mid;
        }
        uint64_t r = lo, end = offsets[r + 1];
        uint32_t lw = rlw[r], tl = rtl[r];
        const uint64_t i0 = g0 - offsets[r];
        uint64_t line = i0 / lw;
        uint32_t col = (uint32_t)(i0 - line * lw);
        uint64_t pos = rbegin[r] + line * ((uint64_t)lw + tl) + col;
        const uint8_t *src = lw == 0xFFFFFFFFu ? side : raw;   // (a record the host de-lined: one line in the side buffer)
        for (int b = 0; b < 32; ++b) {
            const uint64_t g = g0 + b;
            if (g >= T) break;
            while (g >= end) {   // next record with bases
                ++r;
                end = offsets[r + 1];
                if (g < end) {
                    lw = rlw[r];
                    tl = rtl[r];
                    pos = rbegin[r];
                    col = 0;
                    src = lw == 0xFFFFFFFFu ? side : raw;
                }
            }
            const int code = phk_code_of(src[pos], sym);
            const bool ok = code >= 0;
            bad |= !ok;
            w[b >> 4] |= (uint32_t)(ok ? code : 0) << (30 - 2 * (b & 15));
            m |= (uint32_t)ok << (31 - b);
            ++pos;
            if (++col == lw) {
                col = 0;
                pos += tl;
            }
        }
    }
    mask[t] = m;
    if (2 * t < packed_words) packed[2 * t] = w[0];
    if (2 * t + 1 < packed_words) packed[2 * t + 1] = w[1];
    if (bad && any_invalid) atomicOr(any_invalid, 1u);
}

int phk_launch_deline_pack(phk_ctx *ctx, const uint8_t *d_raw, const uint8_t *d_side, const uint64_t *d_offsets, uint64_t n, const uint64_t *d_rbegin,
                           const uint32_t *d_rlw, const uint32_t *d_rtl, uint64_t T, const char *symbols4, uint32_t *d_packed,
                           uint32_t *d_mask, uint32_t *d_any_invalid) {
    PHK_REQUIRE(d_packed && d_mask && d_offsets, "phk_deline_pack: NULL output / offsets");
    PHK_REQUIRE(T == 0 || (d_raw && d_rbegin && d_rlw && d_rtl && n > 0), "phk_deline_pack: NULL input");
    const uint32_t sym = (uint32_t)(uint8_t)symbols4[0] | ((uint32_t)(uint8_t)symbols4[1] << 8) |
                         ((uint32_t)(uint8_t)symbols4[2] << 16) | ((uint32_t)(uint8_t)symbols4[3] << 24);
    const uint64_t packed_words = phk_div_up(T, 16) + 1, mask_words = phk_div_up(T, 32) + 1;
    if (d_any_invalid) PHK_HIP(hipMemsetAsync(d_any_invalid, 0, sizeof(uint32_t), ctx->stream));
    PHK_LAUNCH_SLICED(ctx, "phk_deline_pack_kernel", phk_div_up(mask_words, 256), b0, nblk,
                      phk_deline_pack_kernel<<<dim3(nblk), dim3(256), 0, ctx->stream>>>(
                          d_raw, d_side, d_offsets, n, d_rbegin, d_rlw, d_rtl, T, sym, d_packed, d_mask, packed_words, mask_words, d_any_invalid, b0));
    return PHK_OK;
}

// ------------------------------------------------------------------------------------
// count: one wavefront per contig; per-wave LDS histogram replicated over COPIES lanes
// ------------------------------------------------------------------------------------
// Lane l of a wave-iteration owns packed word w = w0 + l (16 window starts) and reads word
// w+1 for the K-1 bases a window may reach into; the 64-bit funnel X = w:w+1 makes window i
// the bit field X[63-2i .. 64-2K-2i], which IS the reference's bin index
// int(window, 4) (first base most significant, scripts/kmer.py:50).
//
// LDS layout per wave: lane l adds into copy l % COPIES of each bin, so the 32 lanes of an LDS
// group spread over COPIES banks per bin (random bins then collide ~2-way, which the 4-cycle
// ds_add data path hides).
//   PACK16 = false (k <= 4): bins[4^K][COPIES] uint32.
//   PACK16 = true  (k >= 5): rows[4^K/2][COPIES] uint32, row r = bins 2r (low half) and 2r+1 (high
//     half); at most 63 wave iterations (64512 windows) go between flushes, so no half can carry.
// A flush sums the copies with 16-byte LDS reads, clears them, and stores / accumulates uint32.
// atomic: the row receives the partial histograms of several pieces of one long contig (global atomic adds onto a
// zeroed row) instead of being stored
template <int K, int COPIES, bool PACK16>
__device__ __forceinline__ void phk_flush_bins(uint32_t *bins, uint32_t *row_out, bool first, int lane, bool atomic = false) {
    constexpr int D = 1 << (2 * K);
    constexpr int ROWS = PACK16 ? D / 2 : D;
    constexpr int W = ROWS * COPIES;        // dwords per wave
    constexpr int PER = PACK16 ? 2 : 1;     // output bins per row
    for (int base = 0; base < W; base += 256) {
        const int i4 = base + lane * 4;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i4 < W) {
            v = *reinterpret_cast<uint4 *>(bins + i4);
            *reinterpret_cast<uint4 *>(bins + i4) = make_uint4(0, 0, 0, 0);
        }
        if (COPIES >= 4) {  // 4 copies of one row per lane; COPIES/4 neighbouring lanes share the row
            uint32_t sum = v.x + v.y + v.z + v.w;
#pragma unroll
            for (int m = 1; m < COPIES / 4; m <<= 1) sum += __shfl_xor(sum, m);
            const int row = i4 / COPIES;
            if (i4 < W && (lane % (COPIES >= 4 ? COPIES / 4 : 1)) == 0) {
                uint32_t *dst = row_out + PER * row;
                if (atomic) {   // consecutive lanes hold consecutive rows: one wave instruction adds a contiguous run
                    if (PACK16) {
                        if (sum & 0xFFFFu) atomicAdd(dst, sum & 0xFFFFu);
                        if (sum >> 16) atomicAdd(dst + 1, sum >> 16);
                    } else if (sum) {
                        atomicAdd(dst, sum);
                    }
                } else if (PACK16) {
                    uint2 o = make_uint2(sum & 0xFFFFu, sum >> 16);
                    if (!first) {
                        const uint2 old = *reinterpret_cast<uint2 *>(dst);
                        o.x += old.x;
                        o.y += old.y;
                    }
                    *reinterpret_cast<uint2 *>(dst) = o;
                } else {
                    *dst = first ? sum : *dst + sum;
                }
            }
        } else if (!PACK16) {  // COPIES 1 or 2, plain uint32 bins: 4 or 2 bins per lane
            if (i4 < W && atomic) {
                if (COPIES == 1) {
                    const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (vv[e]) atomicAdd(row_out + i4 + e, vv[e]);
                } else {
                    if (v.x + v.y) atomicAdd(row_out + i4 / 2, v.x + v.y);
                    if (v.z + v.w) atomicAdd(row_out + i4 / 2 + 1, v.z + v.w);
                }
            } else if (i4 < W) {
                if (COPIES == 1) {
                    uint4 *dst = reinterpret_cast<uint4 *>(row_out + i4);
                    if (!first) {
                        const uint4 o = *dst;
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    *dst = v;
                } else {
                    uint2 o = make_uint2(v.x + v.y, v.z + v.w);
                    uint2 *dst = reinterpret_cast<uint2 *>(row_out + i4 / 2);
                    if (!first) {
                        const uint2 old = *dst;
                        o.x += old.x; o.y += old.y;
                    }
                    *dst = o;
                }
            }
        } else {  // COPIES == 1 (k = 7, PACK16): four rows per lane -> bins 2*i4 .. 2*i4+7
            if (i4 < W && atomic) {
                const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (vv[e] & 0xFFFFu) atomicAdd(row_out + 2 * (i4 + e), vv[e] & 0xFFFFu);
                    if (vv[e] >> 16) atomicAdd(row_out + 2 * (i4 + e) + 1, vv[e] >> 16);
                }
            } else if (i4 < W) {
                uint4 o0 = make_uint4(v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16);
                uint4 o1 = make_uint4(v.z & 0xFFFFu, v.z >> 16, v.w & 0xFFFFu, v.w >> 16);
                uint4 *dst = reinterpret_cast<uint4 *>(row_out + 2 * i4);
                if (!first) {
                    const uint4 a0 = dst[0], a1 = dst[1];
                    o0.x += a0.x; o0.y += a0.y; o0.z += a0.z; o0.w += a0.w;
                    o1.x += a1.x; o1.y += a1.y; o1.z += a1.z; o1.w += a1.w;
                }
                dst[0] = o0;
                dst[1] = o1;
            }
        }
    }
}

// one wave-iteration: 16 window starts per lane from the funnel a:b of packed word wc
template <int K, int COPIES, bool PACK16, bool MASK>
__device__ __forceinline__ uint32_t phk_count_word(uint32_t *mybins, uint32_t a, uint32_t b, bool active, uint64_t wc,
                                                   uint64_t start, uint64_t last,
                                                   const uint32_t *__restrict__ mask) {
    constexpr uint32_t D = 1u << (2 * K);
    constexpr int LOGC = COPIES == 16 ? 4 : COPIES == 8 ? 3 : COPIES == 4 ? 2 : COPIES == 2 ? 1 : 0;
    const uint64_t X = ((uint64_t)a << 32) | b;
    const uint64_t g0 = wc << 4;
    // window starts i in [lo, hi] of this word belong to the contig
    const int lo = start > g0 ? (int)(start - g0) : 0;
    const int hi = last - g0 < 15 ? (int)(last - g0) : 15;
    uint32_t okbits = active ? ((2u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;  // bit i = window i counted
    if (MASK) {
        const uint64_t mi = wc >> 1;
        const uint64_t V = ((uint64_t)mask[mi] << 32) | mask[mi + 1];
        const uint32_t vb = (uint32_t)(V >> (32 - 16 * (int)(wc & 1)));  // bit 31-i = base g0+i valid
        // window i is valid iff bases i .. i+K-1 are: AND the K shifted copies
        uint32_t wv = vb;
#pragma unroll
        for (int j = 1; j < K; ++j) wv &= vb << j;  // bit 31-i = window i valid
        okbits &= __brev(wv);
    }
    if (!MASK && __all(okbits == 0xFFFFu)) {
        // every lane holds 16 counted windows: no predicates at all
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t idx = (uint32_t)(X >> (64 - 2 * K - 2 * i)) & (D - 1);
            if (PACK16)
                atomicAdd(mybins + ((idx >> 1) << LOGC), 1u + __umul24(idx & 1u, 0xFFFFu));
            else
                atomicAdd(mybins + (idx << LOGC), 1u);
        }
        return 16;
    }
    // edge / masked words: uncounted windows add 0 (no exec-mask churn)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t idx = (uint32_t)(X >> (64 - 2 * K - 2 * i)) & (D - 1);
        const uint32_t ok = (okbits >> i) & 1u;
        if (PACK16)
            atomicAdd(mybins + ((idx >> 1) << LOGC), ok + __umul24(ok & idx, 0xFFFFu));
        else
            atomicAdd(mybins + (idx << LOGC), ok);
    }
    return __popc(okbits);
}

// Batch statistics for the slot kernel's stand-down rule.  A slot workgroup runs as many 1024-window stages
// as the longest of its 32 contigs needs; stats[0] = sum over groups of 32 x that padded maximum, stats[1] =
// windows actually counted.  Below 60 % the wave-per-contig kernel is the faster one and takes the whole batch.
__device__ __forceinline__ bool phk_slots_apply(const unsigned long long *stats) { return stats[1] * 10ull >= stats[0] * 6ull; }

// The kernel is latency-bound unless loads run ahead of the LDS work, so it is software pipelined
// at the contig level: offsets are fetched two contigs ahead, the first PF wave-iterations of
// packed words one contig ahead (a register ring; longer contigs refill the ring as they go).
template <int K, int COPIES, bool PACK16, bool MASK>
__global__ __launch_bounds__(256) void phk_count_kernel(const uint32_t *__restrict__ packed,
                                                        const uint32_t *__restrict__ mask,
                                                        const uint64_t *__restrict__ offsets,
                                                        uint64_t n, uint64_t max_word,
                                                        uint32_t *__restrict__ counts,
                                                        uint32_t *__restrict__ nwin,
                                                        const uint2 *__restrict__ list,
                                                        const uint32_t *__restrict__ list_count,
                                                        uint32_t piece_w) {
    static_assert(COPIES >= 4 || !PACK16 || COPIES == 1, "unsupported replication");
    // with `list`: count the work items list[0 .. *list_count) the slot kernel handed over, item = (contig, piece):
    // the windows [piece * piece_w, (piece + 1) * piece_w) of the contig, added atomically onto its zeroed row
    // (piece_w == 0: whole contigs, stored).  Legacy stand-down (count_sort off): when the batch statistics behind
    // list_count say the slot kernel stood down (ragged batch), everything is counted here instead.
    const bool pieces = list && piece_w != 0;
    if (list && (pieces || phk_slots_apply(reinterpret_cast<const unsigned long long *>(list_count + 2)))) n = *list_count;
    else list = nullptr;
    auto cid = [&](uint64_t i) { return list ? (uint64_t)list[i].x : i; };
    // stream range [s, e) whose windows item i counts (a window is identified by its first base)
    auto bounds = [&](uint64_t i, uint64_t &s, uint64_t &e) {
        const uint64_t c = cid(i);
        s = offsets[c];
        e = offsets[c + 1];
        if (pieces) {
            s += (uint64_t)list[i].y * piece_w;
            const uint64_t pe = s + piece_w + (K - 1);
            e = pe < e ? pe : e;
        }
    };
    constexpr uint32_t D = 1u << (2 * K);
    constexpr int W = (int)(PACK16 ? D / 2 : D) * COPIES;
    constexpr int SEG_ITERS = 63;  // wave iterations between flushes (PACK16 carry bound)
    constexpr int PF = 5;          // prefetched wave-iterations (5120 bases) per contig
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    uint32_t *bins = lds + (size_t)wave * W;
    for (int b = lane * 4; b < W; b += 256) *reinterpret_cast<uint4 *>(bins + b) = make_uint4(0, 0, 0, 0);
    uint32_t *mybins = bins + (lane & (COPIES - 1));  // this lane's copy

    const uint64_t S = (uint64_t)gridDim.x * wpb;
    uint64_t c = (uint64_t)blockIdx.x * wpb + wave;
    if (c >= n) return;

    uint32_t ra[PF], rb[PF], na[PF], nb[PF];
    // words of wave-iterations 0..PF-1 of contig [st, en) into (xa, xb).  The loads are
    // UNCONDITIONAL (indices clamped into the stream) so that they issue back to back: a load under
    // its own branch gets a vmcnt(0) at the join and the prefetch degenerates into serial round trips.
    auto issue = [&](uint64_t st, uint64_t en, uint32_t (&xa)[PF], uint32_t (&xb)[PF]) {
        const uint64_t wb = st >> 4;
        uint64_t we = en >= st + K ? (en - K) >> 4 : wb;
        we = we < max_word ? we : max_word;
#pragma unroll
        for (int t = 0; t < PF; ++t) {
            const uint64_t w = wb + 64ull * t + lane;
            const uint64_t wc = w <= we ? w : we;
            xa[t] = packed[wc];
            xb[t] = packed[wc + 1];
        }
    };
    // count one contig whose first PF iterations sit in (xa, xb); later iterations (contigs longer
    // than 1024*PF bases) are loaded as they come
    auto process = [&](uint64_t cc, uint64_t start, uint64_t end, uint32_t (&xa)[PF], uint32_t (&xb)[PF]) {
        uint32_t *row_out = counts + cc * D;
        uint32_t cnt = 0;
        bool first = true;
        if (end >= start + K) {
            const uint64_t last = end - K;  // last window start
            const uint64_t wb = start >> 4, we = last >> 4;
            const uint64_t niter = (we - wb) / 64 + 1;
            int since_flush = 0;
            for (uint64_t t = 0; t < niter; ++t) {
                const uint64_t w = wb + 64 * t + lane;
                const bool active = w <= we;
                const uint64_t wc = active ? w : we;
                uint32_t a, b;
                if (t < PF) {  // wave-uniform: pick ring slot t (select chain keeps the ring in registers)
                    a = xa[0];
                    b = xb[0];
#pragma unroll
                    for (int j = 1; j < PF; ++j) {
                        a = (t == (uint64_t)j) ? xa[j] : a;
                        b = (t == (uint64_t)j) ? xb[j] : b;
                    }
                } else {
                    a = packed[wc];
                    b = packed[wc + 1];
                }
                cnt += phk_count_word<K, COPIES, PACK16, MASK>(mybins, a, b, active, wc, start, last, mask);
                if (++since_flush == SEG_ITERS && t + 1 < niter) {  // keep 16-bit halves from carrying
                    phk_flush_bins<K, COPIES, PACK16>(bins, row_out, first, lane, pieces);
                    first = false;
                    since_flush = 0;
                }
            }
        }
        phk_flush_bins<K, COPIES, PACK16>(bins, row_out, first, lane, pieces);
        if (nwin) {
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) cnt += __shfl_xor(cnt, s);
            if (lane == 0) {
                if (pieces) { if (cnt) atomicAdd(nwin + cc, cnt); }
                else nwin[cc] = cnt;
            }
        }
    };

    // two rings in ping-pong: while item c is counted out of one, item c+S streams into the other
    uint64_t s0, e0;
    bounds(c, s0, e0);
    uint64_t s1 = 0, e1 = 0;
    if (c + S < n) bounds(c + S, s1, e1);
    issue(s0, e0, ra, rb);
    for (;;) {
        // --- even phase: process (s0,e0) from ra/rb, stream item c+S into na/nb
        uint64_t s2 = 0, e2 = 0;
        if (c + 2 * S < n) bounds(c + 2 * S, s2, e2);
        if (c + S < n) issue(s1, e1, na, nb);
        process(cid(c), s0, e0, ra, rb);
        c += S;
        if (c >= n) break;
        // --- odd phase: process (s1,e1) from na/nb, stream item c+S into ra/rb
        uint64_t s3 = 0, e3 = 0;
        if (c + 2 * S < n) bounds(c + 2 * S, s3, e3);
        if (c + S < n) issue(s2, e2, ra, rb);
        process(cid(c), s1, e1, na, nb);
        c += S;
        if (c >= n) break;
        s0 = s2; e0 = e2;
        s1 = s3; e1 = e3;
    }
}

// ------------------------------------------------------------------------------------
// Length-bucketed contig order for ragged batches.  The slot kernel pads every group of SLOTS contigs to its longest
// member, so a batch of mixed lengths in arbitrary order wastes most of its stages.  A counting sort by the number of
// 1024-window stages (descending: the longest groups start first, the short tail fills the machine at the end) makes
// every group homogeneous.  Three small kernels, all gated ON THE DEVICE by the batch statistics: for a batch that is
// not ragged they return at once (no host round trip decides this).
// What is sorted are work ITEMS (contig, piece): a contig of more than long_thr windows is cut into pieces of piece_w
// windows, each an item of its own with its own histogram column in the slot kernel, added onto the contig's (zeroed) row
// with atomics at the flush -- one 500 kb contig is shared by 15 columns instead of setting the length of its group.
//   key(item) = min(ceil(W_item / 1024), SORT_KEYS - 1)
// ------------------------------------------------------------------------------------
#define SORT_KEYS 2048
__device__ __forceinline__ uint32_t phk_windows_key(uint64_t w) {
    const uint64_t key = (w + 1023) >> 10;
    return (uint32_t)(key < SORT_KEYS - 1 ? key : SORT_KEYS - 1);
}
// windows of contig c and the number of items it is cut into (1: counted whole)
__device__ __forceinline__ uint64_t phk_contig_windows(const uint64_t *offsets, uint64_t c, int k, uint32_t long_thr,
                                                       uint32_t piece_w, uint32_t &pieces) {
    const uint64_t len = offsets[c + 1] - offsets[c];
    const uint64_t w = len >= (uint64_t)k ? len - k + 1 : 0;
    pieces = (w > long_thr && piece_w) ? (uint32_t)((w + piece_w - 1) / piece_w) : 1u;
    return w;
}
// windows of piece pc of a contig of w windows cut into `pieces`
__device__ __forceinline__ uint64_t phk_piece_windows(uint64_t w, uint32_t pieces, uint32_t pc, uint32_t piece_w) {
    if (pieces == 1) return w;
    return pc + 1 < pieces ? (uint64_t)piece_w : w - (uint64_t)(pieces - 1) * piece_w;
}

// The planning kernel: ONE launch in front of the count kernels (round 5; rounds 2-4 spent two memsets and four launches
// here: statistics, key histogram, scan, scatter -- all of them no-ops on a batch of similar lengths).
//   * batch statistics for the stand-down rule (phk_slots_apply): padded / counted windows over groups of `gs` contigs;
//   * sort != 0: the key histogram of the work items, whatever the batch looks like (a block's few non-empty keys);
//   * the LAST block to finish (a ticket) reads the totals back with device-scope loads, and -- only for a batch the
//     statistics call ragged -- turns the histogram into the scatter's cursors and writes the item count;
//   * every block zeroes its share of the NEXT call's control block and of up to two word ranges the caller names (the
//     scorer's NaN counter and call totals): no launch of the chain is preceded by a memset.
// Control block of a call (uint32 words; the count kernels take a pointer to word 0 as `long_count`):
#define CTL_LONG 0      // hand-over items appended by the slot kernels (contigs >> mean, in pieces)
#define CTL_ITEMS 1     // work items of a sorted batch
#define CTL_STATS 2     // two uint64: padded windows, counted windows
#define CTL_TICKET 6
#define CTL_CURSOR 16   // SORT_KEYS words: key histogram, then the scatter's cursors
#define CTL_WORDS (16 + SORT_KEYS)
#define PLAN_PER_BLOCK 2048   // contigs per block and round (a multiple of 32: groups never straddle blocks)
__global__ __launch_bounds__(256) void phk_count_plan_kernel(const uint64_t *__restrict__ offsets, uint64_t n, int k,
                                                             uint32_t gs,  // contigs per slot workgroup (16 or 32)
                                                             uint32_t long_thr, uint32_t piece_w, int sort,
                                                             uint32_t *__restrict__ ctl, uint32_t *__restrict__ ctl_next,
                                                             uint32_t *__restrict__ z0, uint32_t z0n, uint32_t *__restrict__ z1, uint32_t z1n) {
    __shared__ uint32_t h[SORT_KEYS];
    __shared__ unsigned long long s_pad[4], s_used[4];
    __shared__ uint32_t s_last;
    const int tid = threadIdx.x;
    {
        const uint32_t g0 = blockIdx.x * 256u + (uint32_t)tid, gn = gridDim.x * 256u;
        for (uint32_t i = g0; i < (uint32_t)CTL_WORDS; i += gn) ctl_next[i] = 0;
        for (uint32_t i = g0; i < z0n; i += gn) z0[i] = 0;
        for (uint32_t i = g0; i < z1n; i += gn) z1[i] = 0;
    }
    if (sort)
        for (int i = tid; i < SORT_KEYS; i += 256) h[i] = 0;
    __syncthreads();
    unsigned long long pad = 0, used = 0;
    for (uint64_t base = (uint64_t)blockIdx.x * PLAN_PER_BLOCK; base < n; base += (uint64_t)gridDim.x * PLAN_PER_BLOCK) {
        // (all of a round's offsets are requested before any is used: as a load - use - load chain the eight rounds cost eight
        // exposed memory round trips, 25 us per 1M contigs)
        uint64_t o0[PLAN_PER_BLOCK / 256], o1[PLAN_PER_BLOCK / 256];
#pragma unroll
        for (int it = 0; it < PLAN_PER_BLOCK / 256; ++it) {
            const uint64_t c = base + (uint64_t)it * 256 + tid;
            const uint64_t cc = c < n ? c : n - 1;
            o0[it] = offsets[cc];
            o1[it] = offsets[cc + 1];
        }
#pragma unroll
        for (int it = 0; it < PLAN_PER_BLOCK / 256; ++it) {
            const uint64_t c = base + (uint64_t)it * 256 + tid;
            const bool have = c < n;
            uint64_t w = 0;
            if (have) {
                const uint64_t len = o1[it] - o0[it];
                w = len >= (uint64_t)k ? len - k + 1 : 0;
            }
            const unsigned long long wq = (have && w <= long_thr) ? w : 0ull;   // (longer ones leave the slot kernels)
            used += wq;
            unsigned long long mx = wq;
            for (uint32_t sft = 1; sft < gs; sft <<= 1) {
                const unsigned long long o = __shfl_xor(mx, (int)sft);
                mx = o > mx ? o : mx;
            }
            if (have && (c % gs) == 0) pad += (unsigned long long)gs * (((mx + 1023) >> 10) << 10);
            if (sort) {
                // (a batch of similar lengths has ONE key: 256 atomics on one LDS word per round cost this kernel 30 us per 1M
                // contigs -- a wave whose live lanes agree on the key adds their number once)
                const uint32_t np = (have && w > long_thr && piece_w) ? (uint32_t)((w + piece_w - 1) / piece_w) : 1u;
                const uint32_t key = phk_windows_key(phk_piece_windows(w, np, np - 1, piece_w));
                const unsigned long long live = __ballot(have);
                const uint32_t key0 = __builtin_amdgcn_readfirstlane(key);   // (the first ACTIVE lane's: all 64 are active here)
                if (__all(!have || (key == key0 && np == 1)) && (live & 1ull)) {
                    if ((tid & 63) == 0) atomicAdd(&h[key0], (uint32_t)__popcll(live));
                } else if (have) {
                    if (np > 1) atomicAdd(&h[phk_windows_key(piece_w)], np - 1);
                    atomicAdd(&h[key], 1u);
                }
            }
        }
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
        pad += __shfl_xor(pad, sft);
        used += __shfl_xor(used, sft);
    }
    if ((tid & 63) == 0) {
        s_pad[tid >> 6] = pad;
        s_used[tid >> 6] = used;
    }
    __syncthreads();
    unsigned long long *stats = reinterpret_cast<unsigned long long *>(ctl + CTL_STATS);
    if (tid == 0) {   // one pair of atomics per workgroup: they all land on one line and retire one after the other
        pad = s_pad[0] + s_pad[1] + s_pad[2] + s_pad[3];
        used = s_used[0] + s_used[1] + s_used[2] + s_used[3];
        if (pad) {
            atomicAdd(stats, pad);
            atomicAdd(stats + 1, used);
        }
    }
    if (sort)
        for (int i = tid; i < SORT_KEYS; i += 256)
            if (h[i]) atomicAdd(ctl + CTL_CURSOR + i, h[i]);
    // ---- the last block to get here finishes the plan.  What the other blocks publish are device-scope atomics only (performed
    // at the memory side); each wave waits for its own to be done, then the block's ticket is drawn: no cache write-back or
    // invalidate is needed on this side (a __threadfence() here cost every block ~4 us), and the reader uses sc1 loads ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(ctl + CTL_TICKET, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last || !sort) return;
    const unsigned long long pad_t = __hip_atomic_load(stats, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long used_t = __hip_atomic_load(stats + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (used_t * 10ull >= pad_t * 6ull) return;   // (phk_slots_apply: not ragged -- nobody reads the cursors)
    // cursor[key] = number of items with a LARGER key (descending order: the longest groups start first)
    for (int i = tid; i < SORT_KEYS; i += 256)
        h[i] = __hip_atomic_load(ctl + CTL_CURSOR + (SORT_KEYS - 1 - i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid < 64) {   // exclusive scan of 2048 counts by one wave: 32 per lane, then across the lanes
        uint32_t loc = 0;
        for (int i = 0; i < SORT_KEYS / 64; ++i) loc += h[tid * (SORT_KEYS / 64) + i];
        uint32_t inc = loc;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const uint32_t o = __shfl_up(inc, sft);
            if (tid >= sft) inc += o;
        }
        uint32_t run = inc - loc;
        for (int i = 0; i < SORT_KEYS / 64; ++i) {
            const uint32_t cnt = h[tid * (SORT_KEYS / 64) + i];
            h[tid * (SORT_KEYS / 64) + i] = run;
            run += cnt;
        }
        if (tid == 63) ctl[CTL_ITEMS] = run;
    }
    __syncthreads();
    for (int i = tid; i < SORT_KEYS; i += 256) ctl[CTL_CURSOR + (SORT_KEYS - 1 - i)] = h[i];
}

// every block reserves, per key, a range for its items with one global atomic, then places them; the row (and window
// count) of a contig that is cut into pieces is zeroed here, ahead of the pieces' atomic adds
__global__ __launch_bounds__(256) void phk_sort_scatter_kernel(const uint64_t *__restrict__ offsets, uint64_t n, int k,
                                                               uint32_t long_thr, uint32_t piece_w,
                                                               const unsigned long long *__restrict__ stats,
                                                               uint32_t *__restrict__ cursor, uint2 *__restrict__ items,
                                                               uint32_t *__restrict__ counts, uint32_t *__restrict__ nwin) {
    if (phk_slots_apply(stats)) return;
    __shared__ uint32_t h[SORT_KEYS];   // per-block count, then the block's base position, per key
    const uint64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = per_block * blockIdx.x, hi = lo + per_block < n ? lo + per_block : n;
    const uint64_t D = 1ull << (2 * k);
    for (int i = threadIdx.x; i < SORT_KEYS; i += blockDim.x) h[i] = 0;
    __syncthreads();
    for (uint64_t c = lo + threadIdx.x; c < hi; c += blockDim.x) {
        uint32_t np;
        const uint64_t w = phk_contig_windows(offsets, c, k, long_thr, piece_w, np);
        if (np > 1) atomicAdd(&h[phk_windows_key(piece_w)], np - 1);
        atomicAdd(&h[phk_windows_key(phk_piece_windows(w, np, np - 1, piece_w))], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SORT_KEYS; i += blockDim.x)
        if (h[i]) h[i] = atomicAdd(cursor + i, h[i]);
    __syncthreads();
    for (uint64_t c = lo + threadIdx.x; c < hi; c += blockDim.x) {
        uint32_t np;
        const uint64_t w = phk_contig_windows(offsets, c, k, long_thr, piece_w, np);
        for (uint32_t pc = 0; pc < np; ++pc)
            items[atomicAdd(&h[phk_windows_key(phk_piece_windows(w, np, pc, piece_w))], 1u)] = make_uint2((uint32_t)c, pc);
        if (np > 1) {
            uint4 *row = reinterpret_cast<uint4 *>(counts + c * D);
            for (uint64_t i = 0; i < D / 4; ++i) row[i] = make_uint4(0, 0, 0, 0);
            if (nwin) nwin[c] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------
// The slot kernels: one histogram COLUMN per contig.
//
// The wave-per-contig kernel above is bound by LDS bank conflicts: its 64 lanes add into ONE histogram, so a wave-wide
// ds_add lands on random banks (about 7.7 LDS cycles per instruction at the best replication,
// profiles/r01/count_lds_study.md), and replicas have to be summed again at the flush.  In the kernels below the 32
// histograms in a workgroup's LDS belong to 32 DIFFERENT contigs,
//     bins[code][slot]   (slot = lane & 31, so the LDS bank of an add is its lane's slot, whatever the code)
// every ds_add is conflict free (2 LDS cycles per wave instruction), there is nothing to reduce at the flush, and the
// address of a bin costs two VALU operations: ((word >> s) & (D-1) << 7) | slot * 4.  k = 5: 16 contigs per workgroup
// (a 1024-bin column set of 32 would not fit; bank = slot + 16 (code & 1): 3 LDS cycles per instruction on average).
// A lane reads its 64-base units straight from memory (the vector cache serves the other lanes' pieces of a line) and no
// wave waits for another before the flush.  In a batch of similar lengths the few contigs much longer than the mean are
// appended to `long_list` for the wave-per-contig kernel, which follows; a ragged batch is walked as a sorted list of
// (contig, piece) items, a long contig being cut into pieces that are columns like any other.
// (Rounds 2-4 also kept the first form of these kernels, which staged the input through LDS with a barrier per stage --
// phk_count_slots_kernel, 0.90 ms per 1M x 5 kb against 0.66 -- behind count_lanes = 1 / 2 for comparison; removed in
// round 5, the record is in profiles/r04/README.md.)
// ------------------------------------------------------------------------------------
#define PHK_PAIRS_DEFAULT 'P'   // count_lanes unset: the two-windows-per-add kernel (1024 threads) for k = 4 batches without a mask
// LDS-only workgroup barrier: waits for this wave's LDS operations, NOT for its global loads
__device__ __forceinline__ void phk_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------------------------
// TWO WINDOWS PER ADD (k = 4, no validity mask, batches the statistics do not call ragged; round 4).
// The slot kernel above is bound by the CU's LDS adder: one conflict-free wave-wide ds_add_u32 per ~4-6 cycles, one add per
// window (tools/micro/lds_add_rate.hip).  Two consecutive windows -- the 4-mers at bases p and p + 1 -- are
// one 5-mer (bases p .. p + 4): its upper 8 bits are the first window's code, its lower 8 bits the second's.  So the
// contig's windows are taken in PAIRS (p = first base, + 2, + 4, ..), each pair is ONE add into a histogram of stride-2
// 5-mers, and the 4-mer counts are that histogram's two marginals, formed at the flush:
//     count4[x] = sum_b n5[4 x + b]  +  sum_a n5[256 a + x]          (+ 1 for the last window of a contig with an odd number)
// The 1024 5-mer bins are 16-bit halves of 512 words -- bins[code5 >> 1][slot], increment 1 << 16 (code5 & 1) -- 64 KiB for
// 32 contigs; a half holds at most W / 2 <= 65 535 (longer contigs are handed on, as the slot kernel hands on its long
// ones).  Pairs are aligned to the CONTIG's first base: a contig that starts on an odd base has its words
// re-aligned by one base (v_alignbit + v_cndmask per word), after which every shift is an immediate as in the slot
// kernel.  Per add: 2 address operations + 2 for the increment, i.e. the slot kernel's VALU work per window and half its
// LDS adds; the flush reads every bin word twice (4 LDS reads per 4-mer instead of 1).
// ------------------------------------------------------------------------------------
template <int NTH>
__global__ __launch_bounds__(NTH) void phk_count_pairs_kernel(const uint32_t *__restrict__ packed, const uint64_t *__restrict__ offsets,
                                                             uint64_t n, uint64_t max_word, uint32_t long_thr,
                                                             uint32_t *__restrict__ counts, uint32_t *__restrict__ nwin,
                                                             uint2 *__restrict__ long_list, uint32_t *__restrict__ long_count,
                                                             uint32_t piece_w) {
    constexpr int K = 4;
    constexpr uint32_t D = 256, NWORD = 512;   // 4-mer bins of a row; pair words of a histogram column
    constexpr int SLOTS = 32;
    constexpr int PARTS = NTH / SLOTS;      // lanes per contig
    constexpr int XPT = (int)D / PARTS;     // 4-mer codes a thread flushes
    static_assert(XPT % 4 == 0, "flush geometry");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // bins [NWORD][SLOTS] | single [SLOTS]
    uint32_t *single_s = lds + NWORD * SLOTS;
    if (!phk_slots_apply(reinterpret_cast<const unsigned long long *>(long_count + 2))) return;   // ragged: the sorted slot kernel's
    const int t = threadIdx.x;
    const int slot = t & (SLOTS - 1), part = t / SLOTS;
    for (uint32_t b = t * 4; b < NWORD * SLOTS; b += 4 * NTH) *reinterpret_cast<uint4 *>(lds + b) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const uint32_t colb = (uint32_t)slot * 4u;
    const uint32_t pthr = long_thr < 131070u ? long_thr : 131070u;   // a 16-bit half holds W / 2
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    // the pair whose 5-mer starts at base o (even, <= 8) of `src`: bin word (code5 >> 1) at bits [15:7] of the byte address,
    // increment 1 or 65536 by the 5-mer's last bit
    auto pair_addr = [&](uint32_t src, int o) { return (lds_u32 *)(uintptr_t)(((src >> (16 - 2 * o)) & (0x1FFu << 7)) | colb); };
    auto pair_inc = [&](uint32_t src, int o) {   // (bfe + mad: hipcc turns the C form into and + compare + select)
        uint32_t bit, inc;
        asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(bit) : "v"(src), "n"(22 - 2 * o));
        asm("v_mad_u32_u24 %0, %1, %2, 1" : "=v"(inc) : "v"(bit), "v"(65535u));
        return inc;
    };
#if defined(PHK_DIAGNOSTIC_BUILD) && defined(PAIRS_ABL) && PAIRS_ABL == 1   // timing only: no LDS adds (the operands are kept alive)
    auto add1 = [&](lds_u32 *p, uint32_t val) { asm volatile("" ::"v"(p), "v"(val)); };
#else
    auto add1 = [&](lds_u32 *p, uint32_t val) { __hip_atomic_fetch_add(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
#endif
    const uint64_t wlast = max_word + 1;   // the pad word: the last one that exists

    for (uint64_t batch = blockIdx.x; batch * SLOTS < n; batch += gridDim.x) {
        const uint64_t c = batch * SLOTS + slot;
        const bool have = c < n;
        const uint64_t st = have ? offsets[c] : 0;
        const uint64_t en = have ? offsets[c + 1] : 0;
        const uint64_t len = en - st;
        uint32_t W = len >= (uint64_t)K ? (uint32_t)((len - K + 1) < 0xFFFFFFFFull ? (len - K + 1) : 0xFFFFFFFFull) : 0;
        const bool handed_over = W > pthr;
        if (handed_over) {
            if (part == 0) {
                const uint32_t np = piece_w ? (W + piece_w - 1) / piece_w : 1u;
                const uint32_t base = atomicAdd(long_count, np);
                for (uint32_t pc = 0; pc < np; ++pc) long_list[base + pc] = make_uint2((uint32_t)c, pc);
            }
            W = 0;
        }
        const uint32_t par = (uint32_t)st & 1u;
        const uint32_t npair = W >> 1;
        // Units of four words (64 bases, 32 pairs), aligned in the stream; lane `part` of the contig takes units part, part +
        // PARTS, ..  No staging and no barrier until the flush: a lane reads its words straight from memory (the vector
        // cache serves the other lanes' pieces of a line).  Positions relative to the contig's first unit fit 32 bits
        // (W <= 131 070).
        const uint64_t U0 = st >> 6;
        const uint32_t rst = (uint32_t)(st - (U0 << 6));                 // the contig's first base
        const uint32_t rPL = rst + 2u * npair - 2u;                      // first base of its last pair (npair > 0)
        const uint32_t nunit = npair ? (rPL >> 6) + 1u : 0u;
        const uint32_t *pc = packed + 4 * U0;                            // the contig's first unit
        // the word after a unit is read only where the stream still has one (its bases are needed only then)
        const uint32_t wlim = (uint32_t)((wlast - 4 * U0) < 0x7FFFFFFFull ? (wlast - 4 * U0) : 0x7FFFFFFFull);
        if (part == 0) {   // the unpaired last window of a contig with an odd number of them
            uint32_t sg = 0xFFFFFFFFu;
            if (W & 1u) {
                const uint64_t lb = st + W - 1, wl = lb >> 4;
                const uint64_t f = ((uint64_t)packed[wl] << 32) | packed[wl + 1];
                sg = (uint32_t)(f >> (56 - 2 * (uint32_t)(lb & 15))) & 0xFFu;
            }
            single_s[slot] = sg;
        }
        auto load5 = [&](uint32_t j, uint32_t (&w)[5]) {
            const uint32_t w0 = 4u * j;
            if (w0 + 3 <= wlim) {
                const uint4 a = *reinterpret_cast<const uint4 *>(pc + w0);
                w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
            } else {   // the last words of the whole stream
                w[0] = pc[w0]; w[1] = pc[w0 + 1 < wlim ? w0 + 1 : wlim]; w[2] = pc[w0 + 2 < wlim ? w0 + 2 : wlim]; w[3] = pc[wlim];
            }
            w[4] = pc[w0 + 4 < wlim ? w0 + 4 : wlim];
        };
        uint32_t cur[5] = {0, 0, 0, 0, 0}, nxt[5] = {0, 0, 0, 0, 0};
        uint32_t j = (uint32_t)part;
        if (j < nunit) load5(j, cur);
        while (__any(j < nunit)) {
            const bool live = j < nunit;
            if (j + PARTS < nunit) load5(j + PARTS, nxt);
            const uint32_t s0 = 64u * j + par;                            // pair i of the unit starts at s0 + 2 i, i < 32
            const bool all = live && s0 >= rst && s0 + 62u <= rPL;
            // the unit's words, re-aligned to the contig's parity (of the fifth only the top bases are used)
            uint32_t sw[5];
#pragma unroll
            for (int i = 0; i < 4; ++i) sw[i] = par ? __builtin_amdgcn_alignbit(cur[i], cur[i + 1], 30) : cur[i];
            sw[4] = par ? cur[4] << 2 : cur[4];
            if (!__any(live && !all)) {   // wave-uniform: every live lane's unit is interior to its contig
                if (all) {
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) {
                        const uint32_t y = sw[wd], u = __builtin_amdgcn_alignbit(y, sw[wd + 1], 16);
#pragma unroll
                        for (int jp = 0; jp < 8; ++jp) {   // pairs at bases 0, 2, .., 8 of y and 10, 12, 14 = 2, 4, 6 of u
                            const uint32_t src = jp < 5 ? y : u;
                            const int o = jp < 5 ? 2 * jp : 2 * jp - 8;
                            add1(pair_addr(src, o), pair_inc(src, o));
                        }
                    }
                }
            } else if (live) {            // a lane at an edge of its contig: the pair's bit times the increment
                const uint32_t ilo = rst > s0 ? (rst - s0) >> 1 : 0u;
                const uint32_t ihi = (rPL - s0) >> 1 < 31u ? (rPL - s0) >> 1 : 31u;
                const uint32_t pm = (ihi - ilo == 31u) ? ~0u : (((1u << (ihi - ilo + 1)) - 1u) << (31 - ihi));   // bit 31 - i: pair i is counted
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) {
                    const uint32_t y = sw[wd], u = __builtin_amdgcn_alignbit(y, sw[wd + 1], 16);
#pragma unroll
                    for (int jp = 0; jp < 8; ++jp) {
                        const uint32_t src = jp < 5 ? y : u;
                        const int o = jp < 5 ? 2 * jp : 2 * jp - 8;
                        add1(pair_addr(src, o), pair_inc(src, o) * __builtin_amdgcn_ubfe(pm, 31 - (8 * wd + jp), 1));
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) cur[i] = nxt[i];
            j += PARTS;
        }
        phk_lds_barrier();  // every wave's adds (and the single windows) have landed
        // ---- flush: thread (slot, g) forms the 4-mer counts x in [g XPT, (g + 1) XPT) of contig `slot` from the two marginals ----
        {
            const uint32_t x0 = (uint32_t)part * XPT;
            uint32_t out[XPT];
            const uint32_t *col = lds + slot;
            // (v_sad_u16 with a zero operand adds both halves of a word onto an accumulator; v_mad_u32_u16 x 1 one half, chosen
            // by op_sel: the unpacking costs no instruction of its own)
#pragma unroll
            for (int i = 0; i < XPT; ++i) {                    // first window of the pair: 5-mers 4 x .. 4 x + 3 = words 2 x, 2 x + 1
                const uint32_t w0 = col[(2 * (x0 + i)) * SLOTS], w1 = col[(2 * (x0 + i) + 1) * SLOTS];
                out[i] = __builtin_amdgcn_sad_u16(w1, 0u, __builtin_amdgcn_sad_u16(w0, 0u, 0u));
            }
            const uint32_t one = 1u;
#pragma unroll
            for (int mm = 0; mm < XPT / 2; ++mm)               // second window: 5-mers 256 a + x = half x & 1 of word 128 a + x / 2
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const uint32_t w = col[(128 * a + x0 / 2 + mm) * SLOTS];
                    asm("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(out[2 * mm]) : "v"(w), "v"(one));
                    asm("v_mad_u32_u16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(out[2 * mm + 1]) : "v"(w), "v"(one));
                }
            const uint32_t sg = single_s[slot];
#pragma unroll
            for (int i = 0; i < XPT; ++i) out[i] += (sg == x0 + i) ? 1u : 0u;
            phk_lds_barrier();   // every thread has read what it needs: the columns may be cleared
            uint32_t *zc = lds + (2 * x0) * SLOTS + slot;
#pragma unroll
            for (int i = 0; i < 2 * XPT; ++i) zc[i * SLOTS] = 0;
            uint32_t *rowo = counts + c * D + x0;
#if defined(PHK_DIAGNOSTIC_BUILD) && defined(PAIRS_ABL) && PAIRS_ABL == 2   // timing only: rows are stored for one batch in 64
            if (have && (!handed_over || piece_w) && (batch & 63) == 0) {
#else
            if (have && (!handed_over || piece_w)) {           // (a contig handed over in pieces gets its zero row here)
#endif
#pragma unroll
                for (int i = 0; i < XPT / 4; ++i)
                    *reinterpret_cast<uint4 *>(rowo + 4 * i) = make_uint4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
            }
            if (nwin && have && part == 0 && (!handed_over || piece_w)) nwin[c] = W;
        }
        phk_lds_barrier();   // the columns are clear and single_s may be rewritten
    }
}

// ------------------------------------------------------------------------------------
// The slot kernel WITHOUT staging and stage barriers (k = 5, no validity mask, batches the statistics do not call ragged;
// round 4): what made the two-windows-per-add kernel above fast was, as much as its halved adds, that a lane reads its
// words straight from memory and no wave waits for another before the flush.  The bins described above
// (bins[code][slot], 16 contigs per workgroup at k = 5), one add per window, 64 windows per lane and round.
// ------------------------------------------------------------------------------------
template <int K, int SLOTS, int NTH, bool MASK, bool ORDERED>
__global__ __launch_bounds__(NTH) void phk_count_direct_kernel(const uint32_t *__restrict__ packed, const uint32_t *__restrict__ mask,
                                                              const uint64_t *__restrict__ offsets,
                                                              uint64_t n, uint64_t max_word, uint32_t long_thr,
                                                              uint32_t *__restrict__ counts, uint32_t *__restrict__ nwin,
                                                              uint2 *__restrict__ long_list, uint32_t *__restrict__ long_count,
                                                              const uint2 *__restrict__ order,   // sorted (contig, piece) items of a ragged batch, or NULL
                                                              uint32_t piece_w,
                                                              uint32_t skip_plain,             // 1: a batch that is not ragged is another kernel's (pairs)
                                                              uint4 *__restrict__ frag8,      // the scorer's int8 operand (PhkPrep8), or NULL
                                                              uint32_t *__restrict__ big8) {
    constexpr uint32_t D = 1u << (2 * K);
    constexpr int PARTS = NTH / SLOTS;      // lanes per contig
    constexpr int XPT = (int)D / PARTS;     // codes a thread flushes
    constexpr int SHB = SLOTS == 32 ? 7 : 6;  // log2 of a bin row in bytes
    static_assert(XPT % 4 == 0 && K <= 5, "flush geometry / window fits the funnel");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // bins [D][SLOTS] | counted windows [SLOTS] | piece owner [SLOTS]
    uint32_t *nwin_s = lds + D * SLOTS;
    uint32_t *split_s = nwin_s + SLOTS;
    // Two instances are launched back to back and decide on the device which of them counts the batch: the plain walk
    // (ORDERED = false) when the statistics allow it, else the sorted work list (ORDERED = true).
    // The plain instance carries none of the sorted walk's code: with it the k = 5 kernel ran 15 % slower.
    const bool plain = phk_slots_apply(reinterpret_cast<const unsigned long long *>(long_count + 2));
    if (ORDERED) {
        if (plain || !order) return;
        n = long_count[1];   // items, counted by the sort kernels
        frag8 = nullptr;
        big8 = nullptr;
    } else {
        if (!plain) {
            if (big8 && blockIdx.x == 0 && threadIdx.x == 0) big8[n] = 2u;   // (nothing prepared for the scorer)
            return;
        }
        if (skip_plain) return;
        order = nullptr;
    }
    const int t = threadIdx.x;
    const int slot = t & (SLOTS - 1), part = t / SLOTS;
    for (uint32_t b = t * 4; b < D * SLOTS; b += 4 * NTH) *reinterpret_cast<uint4 *>(lds + b) = make_uint4(0, 0, 0, 0);
    if (t < SLOTS) nwin_s[t] = 0;
    __syncthreads();
    const uint32_t colb = (uint32_t)slot * 4u;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    auto bin = [&](uint32_t src, int o) {   // the window starting at base o (< 8) of `src`
        constexpr uint32_t msk = (D - 1u) << SHB;
        const int sh = 32 - 2 * K - 2 * o - SHB;
        return (lds_u32 *)(uintptr_t)(((src >> sh) & msk) | colb);
    };
    auto add1 = [&](lds_u32 *p, uint32_t val) { __hip_atomic_fetch_add(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    const uint64_t wlast = max_word + 1;   // the pad word: the last one that exists
    const uint64_t mlast = (max_word >> 1) + 1;   // last mask word that exists (ceil(T / 32) + 1 words)

    for (uint64_t batch = blockIdx.x; batch * SLOTS < n; batch += gridDim.x) {
        const uint64_t ci = batch * SLOTS + slot;
        const bool have = ci < n;
        const uint2 item = (ORDERED && have) ? order[ci] : make_uint2(0, 0);
        const uint64_t c = have ? (ORDERED ? (uint64_t)item.x : ci) : 0;
        uint64_t st = have ? offsets[c] : 0;
        const uint64_t en = have ? offsets[c + 1] : 0;
        const uint64_t len = en - st;
        uint32_t W = len >= (uint64_t)K ? (uint32_t)((len - K + 1) < 0xFFFFFFFFull ? (len - K + 1) : 0xFFFFFFFFull) : 0;
        // an item of the sorted work list is a whole contig or one piece of a long one: the piece's windows start at
        // st + piece * piece_w, and its histogram is added onto the contig's row (zeroed by the sort) at the flush
        const bool split = ORDERED && piece_w && W > long_thr;
        if (split) {
            const uint64_t first = (uint64_t)item.y * piece_w;
            st += first;
            W = (uint32_t)((uint64_t)W - first < piece_w ? (uint64_t)W - first : piece_w);
        }
        const bool handed_over = !ORDERED && W > long_thr;
        if (ORDERED && part == 0) split_s[slot] = split ? (uint32_t)c + 1u : 0u;
        if (handed_over) {
            if (part == 0) {
                const uint32_t np = piece_w ? (W + piece_w - 1) / piece_w : 1u;
                const uint32_t base = atomicAdd(long_count, np);
                for (uint32_t pc = 0; pc < np; ++pc) long_list[base + pc] = make_uint2((uint32_t)c, pc);
            }
            W = 0;
        }
        // units of four words (64 windows), aligned in the stream; positions relative to the contig's first unit
        const uint64_t U0 = st >> 6;
        const uint32_t rst = (uint32_t)(st - (U0 << 6));
        const uint64_t rlast64 = (uint64_t)rst + W - 1;                  // the last window (W > 0)
        const uint32_t rlast = rlast64 < 0xFFFFFFC0ull ? (uint32_t)rlast64 : 0xFFFFFFC0u;   // (long_thr / piece_w keep W far below this)
        const uint32_t nunit = W ? (rlast >> 6) + 1u : 0u;
        const uint32_t *pc = packed + 4 * U0;
        const uint32_t wlim = (uint32_t)((wlast - 4 * U0) < 0x7FFFFFFFull ? (wlast - 4 * U0) : 0x7FFFFFFFull);
        const uint32_t *pm = MASK ? mask + 2 * U0 : nullptr;            // validity words of the first unit (2 per unit + the next)
        const uint32_t mlim = MASK ? (uint32_t)((mlast - 2 * U0) < 0x7FFFFFFFull ? (mlast - 2 * U0) : 0x7FFFFFFFull) : 0u;
        auto load5 = [&](uint32_t j, uint32_t (&w)[5], uint32_t (&mk)[3]) {
            const uint32_t w0 = 4u * j;
            if (w0 + 3 <= wlim) {
                const uint4 a = *reinterpret_cast<const uint4 *>(pc + w0);
                w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
            } else {   // the last words of the whole stream
                w[0] = pc[w0]; w[1] = pc[w0 + 1 < wlim ? w0 + 1 : wlim]; w[2] = pc[w0 + 2 < wlim ? w0 + 2 : wlim]; w[3] = pc[wlim];
            }
            w[4] = pc[w0 + 4 < wlim ? w0 + 4 : wlim];
            if (MASK) {
#pragma unroll
                for (int i = 0; i < 3; ++i) mk[i] = pm[2 * j + i < mlim ? 2 * j + i : mlim];
            }
        };
        uint32_t cur[5] = {0, 0, 0, 0, 0}, nxt[5] = {0, 0, 0, 0, 0}, mcur[3] = {0, 0, 0}, mnxt[3] = {0, 0, 0};
        uint32_t cnt_ok = 0;
        uint32_t j = (uint32_t)part;
        if (j < nunit) load5(j, cur, mcur);
        while (__any(j < nunit)) {
            const bool live = j < nunit;
            if (j + PARTS < nunit) load5(j + PARTS, nxt, mnxt);
            const uint32_t fb = 64u * j;                                  // window i of the unit starts at fb + i
            bool all = live && fb >= rst && fb + 63u <= rlast;
            uint64_t wv = 0;                                              // bit 63 - i: window i is counted
            if (MASK ? live : (live && !all)) {   // (without a mask only the units at a contig's edges need their window bits)
                const uint32_t lo = rst > fb ? rst - fb : 0u;
                const uint32_t hi = rlast - fb < 63u ? rlast - fb : 63u;
                wv = (hi - lo == 63u) ? ~0ull : (((1ull << (hi - lo + 1)) - 1ull) << (63 - hi));
                if (MASK) {
                    const uint64_t vb = ((uint64_t)mcur[0] << 32) | mcur[1];   // bit 63 - i: base i of the unit valid
                    uint64_t w = vb;
#pragma unroll
                    for (int jj = 1; jj < K; ++jj) w &= (vb << jj) | ((uint64_t)mcur[2] >> (32 - jj));
                    wv &= w;
                    all = all && w == ~0ull;
                    cnt_ok += (uint32_t)__popcll(wv);
                }
            }
            if (!__any(live && !all)) {   // wave-uniform: every live lane's unit is interior to its contig (and all valid)
                if (all) {
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) {
                        const uint32_t y = cur[wd], u = __builtin_amdgcn_alignbit(y, cur[wd + 1], 16);
#pragma unroll
                        for (int jw = 0; jw < 16; ++jw) add1(bin(jw < 8 ? y : u, jw & 7), 1u);
                    }
                }
            } else if (live) {            // a lane at an edge of its contig / with an invalid base: the window's bit instead of 1
                if (!MASK && all) wv = ~0ull;   // (an interior unit in a wave that has an edge unit elsewhere)
                const uint32_t vhi = (uint32_t)(wv >> 32), vlo = (uint32_t)wv;
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) {
                    const uint32_t y = cur[wd], u = __builtin_amdgcn_alignbit(y, cur[wd + 1], 16);
#pragma unroll
                    for (int jw = 0; jw < 16; ++jw) {
                        const int wi = 16 * wd + jw;
                        add1(bin(jw < 8 ? y : u, jw & 7), __builtin_amdgcn_ubfe(wi < 32 ? vhi : vlo, 31 - (wi & 31), 1));
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) cur[i] = nxt[i];
            if (MASK) {
#pragma unroll
                for (int i = 0; i < 3; ++i) mcur[i] = mnxt[i];
            }
            j += PARTS;
        }
        if (MASK && cnt_ok) atomicAdd(nwin_s + slot, cnt_ok);
        phk_lds_barrier();  // every wave's adds have landed
        // ---- pieces first: wave w adds the columns of slots w, w + NTH / 64, .. onto their contigs' rows, 64 consecutive codes per
        // instruction (coalesced global atomics)
        if (ORDERED) {
            for (int sl = t >> 6; sl < SLOTS; sl += NTH / 64) {
                const uint32_t cs = split_s[sl];
                if (!cs) continue;
                uint32_t *rowp = counts + (uint64_t)(cs - 1u) * D;
                for (uint32_t code = (uint32_t)(t & 63); code < D; code += 64) {
                    const uint32_t v = lds[code * SLOTS + sl];
                    if (v) atomicAdd(rowp + code, v);
                }
            }
            phk_lds_barrier();
        }
        {   // flush: thread (slot, g) writes codes [g XPT, (g + 1) XPT) of contig `slot` and clears them
            uint32_t *cellb = lds + ((uint32_t)part * XPT) * SLOTS + slot;
            uint32_t *rowo = counts + c * D + (uint32_t)part * XPT;
            const uint32_t Wc = MASK ? nwin_s[slot] : W;   // counted windows = the row sum
            // ... and, for the scorer's int8 sweep (score_i8.hip), the same codes as int8 around the row's centre in fragment
            // order: piece (k-step s = dims / 32, half h) of query c at [(c / 32) (D / 32) + s][32 h + c % 32]
            const int cen = (int)phk_row_center(Wc, D);
            uint32_t mx = 0, l1 = 0;
            constexpr int PZ = XPT >= 16 ? XPT / 16 : 1, CPZ = XPT >= 16 ? 4 : XPT / 4;   // 16-code pieces per thread, uint4 loads per piece
#pragma unroll
            for (uint32_t pz = 0; pz < (uint32_t)PZ; ++pz) {
                uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
                for (uint32_t i = 0; i < (uint32_t)CPZ; ++i) {
                    uint32_t *cell = cellb + (16 * pz + 4 * i) * SLOTS;
                    const uint4 o = make_uint4(cell[0], cell[SLOTS], cell[2 * SLOTS], cell[3 * SLOTS]);
                    cell[0] = 0; cell[SLOTS] = 0; cell[2 * SLOTS] = 0; cell[3 * SLOTS] = 0;
                    // (a contig handed over in pieces gets its zero row here: the pieces are added onto it atomically)
                    if (!split && have && (!handed_over || piece_w)) *reinterpret_cast<uint4 *>(rowo + 16 * pz + 4 * i) = o;
                    if (XPT >= 16) {
                        const uint32_t cc[4] = {o.x, o.y, o.z, o.w};
                        uint32_t word = 0;
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const int d = (int)cc[b] - cen;
                            const uint32_t ad = (uint32_t)(d < 0 ? -d : d);
                            mx = max(mx, ad);
                            l1 += min(ad, 127u);
                            const int q = d < -127 ? -127 : (d > 127 ? 127 : d);
                            word |= ((uint32_t)q & 0xFFu) << (8 * b);
                        }
                        pk[i] = word;
                    }
                }
                if (XPT >= 16 && frag8 && have && !handed_over) {
                    const uint32_t dim16 = (uint32_t)part * PZ + pz;   // which 16 dimensions of the row
                    frag8[((c >> 5) * (D / 32) + (dim16 >> 1)) * 64 + 32u * (dim16 & 1u) + (uint32_t)(c & 31)] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                }
            }
            if (big8 && have) {
                if (handed_over) {   // counted later, in pieces: the scorer's own fragment kernel takes this row
                    if (part == 0) {
                        big8[c] = PHK_PREP8_MISSING;
                        atomicOr(big8 + n, 1u);
                    }
                } else {
                    if (mx > 127u) atomicOr(big8 + c, 0x80000000u);
                    if (2ull * Wc + D > PHK_I8_L1_MAX) atomicAdd(big8 + c, l1);
                }
            }
            if (nwin && have && part == 0) {
                if (split) atomicAdd(nwin + c, Wc);
                else if (!handed_over || piece_w) nwin[c] = Wc;
            }
        }
        phk_lds_barrier();
        if (MASK) {
            if (t < SLOTS) nwin_s[t] = 0;
            phk_lds_barrier();
        }
    }
}

// replication / packing per k: keep a wave's bins <= 32 KiB
template <int K> struct PhkCountCfg {
    static constexpr bool pack16 = K >= 6;
    static constexpr int copies = K <= 4 ? 4 : 1;  // measured best, profiles/r01/count_lds_study.md
};

template <int K, int COPIES, bool P16>
static int launch_count_cfg(phk_ctx *ctx, const uint32_t *d_packed, const uint32_t *d_mask,
                            const uint64_t *d_offsets, uint64_t n, uint64_t max_word, uint32_t *d_counts,
                            uint32_t *d_nwin, const uint2 *d_list = nullptr, const uint32_t *d_list_count = nullptr,
                            uint32_t piece_w = 0) {
    constexpr uint32_t D = 1u << (2 * K);
    constexpr size_t wave_bytes = (size_t)(P16 ? D / 2 : D) * COPIES * 4u;
    // waves per block so that a block's bins stay <= 64 KiB
    const int wpb = wave_bytes * 4 <= 65536 ? 4 : (wave_bytes * 2 <= 65536 ? 2 : 1);
    const size_t lds = wave_bytes * wpb;
    int per_cu = (int)((160u * 1024u) / lds);
    per_cu = per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu);
    uint64_t blocks = phk_div_up(n, wpb);
    const uint64_t cap = (uint64_t)ctx->num_cus * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) return PHK_OK;
    if (d_mask) {
        PHK_LAUNCH(ctx, "phk_count_kernel",
                   phk_count_kernel<K, COPIES, P16, true><<<dim3((unsigned)blocks), dim3(64 * wpb), lds, ctx->stream>>>(
                       d_packed, d_mask, d_offsets, n, max_word, d_counts, d_nwin, d_list, d_list_count, piece_w));
    } else {
        PHK_LAUNCH(ctx, "phk_count_kernel",
                   phk_count_kernel<K, COPIES, P16, false><<<dim3((unsigned)blocks), dim3(64 * wpb), lds, ctx->stream>>>(
                       d_packed, d_mask, d_offsets, n, max_word, d_counts, d_nwin, d_list, d_list_count, piece_w));
    }
    return PHK_OK;
}

template <int K>
static int launch_count_k(phk_ctx *ctx, const uint32_t *d_packed, const uint32_t *d_mask,
                          const uint64_t *d_offsets, uint64_t n, uint64_t max_word, uint32_t *d_counts,
                          uint32_t *d_nwin) {
    // (the replication / packing of a k was settled by the LDS study of round 1, profiles/r01/count_lds_study.md; the knob that
    // selected other built variants -- 60 kernel instances -- is gone since round 5)
    return launch_count_cfg<K, PhkCountCfg<K>::copies, PhkCountCfg<K>::pack16>(ctx, d_packed, d_mask, d_offsets, n,
                                                                              max_word, d_counts, d_nwin);
}

// Per-device set-up, called from phk_create with the context's device current: raise the dynamic LDS limit of the
// k = 5 slot kernel and verify, for every slot kernel instance, that it has no static LDS -- its bin addresses are
// formed as integers on the assumption that the dynamic array starts at LDS address 0.  If a toolchain ever lays
// the kernel out differently the slot kernel is simply not used on this context (the wave-per-contig kernel
// serves every k): a host-side refusal instead of a device-side abort.
template <typename Kern>
static int slots_instance_ok(Kern kern, bool *ok) {
    hipFuncAttributes fa;
    PHK_HIP(hipFuncGetAttributes(&fa, (const void *)kern));
    if (fa.sharedSizeBytes != 0) *ok = false;
    return PHK_OK;
}

int phk_count_init_device(phk_ctx *ctx) {
    PHK_HIP(hipFuncSetAttribute((const void *)phk_count_pairs_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    PHK_HIP(hipFuncSetAttribute((const void *)phk_count_pairs_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    bool ok = true;
#define PHK_DIRECT_INIT1(K_, S_, T_, M_, O_)                                                                                         \
    PHK_HIP(hipFuncSetAttribute((const void *)phk_count_direct_kernel<K_, S_, T_, M_, O_>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); \
    PHK_TRY(slots_instance_ok(phk_count_direct_kernel<K_, S_, T_, M_, O_>, &ok))
#define PHK_DIRECT_INIT(K_, S_, T_)                                                                                                  \
    PHK_DIRECT_INIT1(K_, S_, T_, false, false); PHK_DIRECT_INIT1(K_, S_, T_, false, true);                                         \
    PHK_DIRECT_INIT1(K_, S_, T_, true, false); PHK_DIRECT_INIT1(K_, S_, T_, true, true)
    PHK_DIRECT_INIT(3, 32, 512);
    PHK_DIRECT_INIT(4, 32, 512);
    PHK_DIRECT_INIT(4, 32, 1024);
    PHK_DIRECT_INIT(5, 16, 512);
    PHK_DIRECT_INIT(5, 16, 1024);
#undef PHK_DIRECT_INIT
#undef PHK_DIRECT_INIT1
    PHK_TRY(slots_instance_ok(phk_count_pairs_kernel<512>, &ok));
    PHK_TRY(slots_instance_ok(phk_count_pairs_kernel<1024>, &ok));
    ctx->slots_lds0 = ok;
    return PHK_OK;
}

// mean_bases: the batch's mean contig length when `d_offsets` addresses a sub-range of the stream (T then only bounds
// the loads); 0 = T / n
int phk_launch_count(phk_ctx *ctx, const uint32_t *d_packed, const uint32_t *d_mask, uint64_t T,
                     const uint64_t *d_offsets, uint64_t n, int k, uint32_t *d_counts,
                     uint32_t *d_nwin, uint64_t mean_bases) {
    // phk_count_score_dev may have armed the int8 operand hand-over (PhkPrep8): it stays armed for the scorer ONLY if the one
    // kernel that writes the fragments is launched below -- every other way out of this function (tiny batches, empty batches,
    // knobs, the wave-per-contig kernel) leaves it disarmed, and the scorer prepares its operand itself
    const bool prep8_asked = ctx->prep8.armed;
    ctx->prep8.armed = false;
    PHK_REQUIRE(k >= 1, "phk_count: k must be >= 1 (got %d)", k);
    if (k > PHK_MAX_K) {
        phk_set_error("phk_count: k=%d is above PHK_MAX_K=%d (4^k bins no longer fit LDS)", k, PHK_MAX_K);
        return PHK_ERR_UNSUPPORTED;
    }
    if (n == 0) return PHK_OK;
    PHK_REQUIRE(d_packed && d_offsets && d_counts, "phk_count: NULL device pointer");
    PHK_REQUIRE(((uintptr_t)d_counts & 15) == 0, "phk_count: counts must be 16-byte aligned");
    if (T == 0) {  // only empty contigs: all-zero rows, nothing to read
        PHK_HIP(hipMemsetAsync(d_counts, 0, n * phk_pow4(k) * sizeof(uint32_t), ctx->stream));
        if (d_nwin) PHK_HIP(hipMemsetAsync(d_nwin, 0, n * sizeof(uint32_t), ctx->stream));
        return PHK_OK;
    }
    const uint64_t max_word = (T - 1) >> 4;  // last word holding a base; word max_word + 1 exists (pad)
    // k = 3, 4 without invalid bases: slot kernel (32 contigs per workgroup, conflict-free LDS adds); contigs
    // more than 4x the batch mean go on to the wave-per-contig kernel through a device list
    const char lanes_knob = ctx->knobs.count_lanes;
    if (k >= 3 && k <= 5 && max_word >= 64 && n < (1ull << 32) && lanes_knob != '0' && ctx->slots_lds0) {
        const uint32_t slots = k == 5 ? 16u : 32u;
        // count_lanes: 'f' = the unstaged slot kernel whatever the batch looks like (tests), 'q' / 'Q' = the same with the two-windows-per-add
        // kernel (512 / 1024 threads) for k = 4 without a mask; 'p' / 'P' = that kernel where the statistics allow
        const bool forced = lanes_knob == 'q' || lanes_knob == 'Q' || lanes_knob == 'f';   // ('f': the unstaged slot kernel, forced)
        const bool sorted = ctx->knobs.count_sort && !forced;
        // contigs much longer than the batch mean leave the slot kernel (a workgroup runs as many stages as its longest
        // contig): with count_sort they go to the wave-per-contig kernel as PIECES of 32768 windows, each its own work
        // item adding onto the zeroed row, so that one 500 kb contig is shared by 15 waves instead of pinning one
        const uint32_t piece_w = sorted ? 32768u : 0u;
        const uint64_t mean_len = (mean_bases ? mean_bases : T / n) + 1;
        uint64_t thr64 = 4 * mean_len + 1024;
        if (sorted && thr64 < 2ull * piece_w) thr64 = 2ull * piece_w;
        const uint32_t long_thr = thr64 < 0xFFFFFFFFull ? (uint32_t)thr64 : 0xFFFFFFFFu;
        // workspace: the call's control block (PhkCountCtl words, see phk_count_plan_kernel) in WS_CTL; hand-over items
        // (contig, piece) and the sorted contig order in WS_LONG
        const uint64_t max_items = n + (piece_w ? T / piece_w : 0) + 16;
        void *ws, *ctlv;
        PHK_TRY(phk_ws(ctx, WS_LONG, (4 * max_items + 16) * sizeof(uint32_t), &ws));
        PHK_TRY(phk_ws(ctx, WS_CTL, 2 * CTL_WORDS * sizeof(uint32_t), &ctlv));
        if (ctx->ctl_dirty || ctx->ctl_gen != ctx->ws[WS_CTL].gen) {   // first use of this allocation, or a call that failed half way
            PHK_HIP(hipMemsetAsync(ctlv, 0, 2 * CTL_WORDS * sizeof(uint32_t), ctx->stream));
            ctx->ctl_gen = ctx->ws[WS_CTL].gen;
        }
        ctx->ctl_dirty = true;   // (cleared where this function returns with everything launched)
        uint32_t *d_long_count = (uint32_t *)ctlv + (ctx->ctl_epoch & 1) * CTL_WORDS;
        uint32_t *d_ctl_next = (uint32_t *)ctlv + ((ctx->ctl_epoch + 1) & 1) * CTL_WORDS;
        uint2 *d_long_list = (uint2 *)ws;
        uint32_t *d_cursor = d_long_count + CTL_CURSOR;
        uint2 *d_order = (uint2 *)((uint32_t *)ws + 2 * max_items);
        {
            uint64_t pb = phk_div_up(forced ? 1 : n, PLAN_PER_BLOCK);
            pb = pb > 1024 ? 1024 : (pb < 1 ? 1 : pb);
            PHK_LAUNCH(ctx, "phk_count_plan_kernel",
                       phk_count_plan_kernel<<<dim3((unsigned)pb), dim3(256), 0, ctx->stream>>>(
                           d_offsets, forced ? 0 : n, k, slots, long_thr, piece_w, sorted ? 1 : 0, d_long_count, d_ctl_next,
                           ctx->plan_zero[0], ctx->plan_zero_words[0], ctx->plan_zero[1], ctx->plan_zero_words[1]));
            ctx->ctl_epoch += 1;
            ctx->plan_zero_taken = ctx->plan_zero[0] != nullptr || ctx->plan_zero[1] != nullptr;
            ctx->plan_zero[0] = ctx->plan_zero[1] = nullptr;
            ctx->plan_zero_words[0] = ctx->plan_zero_words[1] = 0;
        }
        if (sorted) {   // returns at once on the device unless the statistics call the batch ragged
            const unsigned long long *st = (const unsigned long long *)(d_long_count + CTL_STATS);
            uint64_t sb = phk_div_up(n, 2048);
            sb = sb > 1024 ? 1024 : (sb < 1 ? 1 : sb);
            PHK_LAUNCH(ctx, "phk_sort_scatter_kernel",
                       phk_sort_scatter_kernel<<<dim3((unsigned)sb), dim3(256), 0, ctx->stream>>>(d_offsets, n, k, long_thr, piece_w, st,
                                                                                                  d_cursor, d_order, d_counts, d_nwin));
        }
        const uint2 *d_ord = sorted ? d_order : nullptr;
        // k = 4 without a validity mask: a batch the statistics do not call ragged is counted two windows per add
        // (phk_count_pairs_kernel); the slot kernel below then only serves the ragged case (both decide on the device)
        uint32_t skip_plain = 0;
        const char pk = lanes_knob == 'p' || lanes_knob == 'q' ? 'p' : (lanes_knob == 'P' || lanes_knob == 'Q' ? 'P' : (lanes_knob == 0 ? PHK_PAIRS_DEFAULT : 0));
        if (k == 4 && !d_mask && pk) {
            const int nth = pk == 'P' ? 1024 : 512;   // 2 workgroups per CU either way: 32 / 16 waves
            const size_t plds = (size_t)512 * 32 * 4 + 32 * 4;
            const unsigned pfit = (unsigned)((160u * 1024u - 1024u) / plds);
            const unsigned pper = nth == 1024 ? (pfit > 2 ? 2 : pfit) : (pfit > 4 ? 4 : pfit);
            uint64_t pblocks = phk_div_up(n, 32);
            const uint64_t pcap = (uint64_t)ctx->num_cus * pper;
            if (pblocks > pcap) pblocks = pcap;
            if (nth == 1024) {
                PHK_LAUNCH(ctx, "phk_count_pairs_kernel",
                           (phk_count_pairs_kernel<1024><<<dim3((unsigned)pblocks), dim3(1024), plds, ctx->stream>>>(
                               d_packed, d_offsets, n, max_word, long_thr, d_counts, d_nwin, d_long_list, d_long_count, piece_w)));
            } else {
                PHK_LAUNCH(ctx, "phk_count_pairs_kernel",
                           (phk_count_pairs_kernel<512><<<dim3((unsigned)pblocks), dim3(512), plds, ctx->stream>>>(
                               d_packed, d_offsets, n, max_word, long_thr, d_counts, d_nwin, d_long_list, d_long_count, piece_w)));
            }
            skip_plain = 1;
        }
        // The unstaged slot kernel (phk_count_direct_kernel) for everything else the slot kernel used to count: masked batches,
        // the sorted walk of ragged batches, k = 3 and k = 5.
        // At k = 5 without a mask it also writes the scorer's int8 operand when phk_count_score_dev armed it for this matrix.
        {
            PhkPrep8 &pp = ctx->prep8;
            const bool prep = k == 5 && !d_mask && prep8_asked && pp.counts == d_counts && pp.n == n && pp.D == 1024;
            uint4 *frag8 = prep ? (uint4 *)pp.frag : nullptr;
            uint32_t *big8 = prep ? pp.big : nullptr;
            pp.armed = prep;   // (stays armed only if the kernel that prepares it is launched)
            const size_t dlds = (size_t)phk_pow4(k) * slots * 4 + 2 * slots * 4;
            const unsigned dfit = (unsigned)((160u * 1024u - 1024u) / dlds);
            // shapes, measured (1M contigs, ms): k = 3 unmasked 0.70 (slot kernel 0.79); k = 4 masked 1.00 with 512 threads, 1.22 with 1024
            // (slot kernel 1.13-1.15); k = 5 masked 2.68 with 1024 (slot kernel 3.87); ragged k = 4 / k = 5: 0.94 / 1.12 (0.92 / 1.58)
            const bool small = lanes_knob == 'd' || lanes_knob == 'p' || lanes_knob == 'q' || (k == 4 && lanes_knob != 'D');
            const unsigned dth = (k == 3 || small) ? 512u : 1024u;
            const unsigned dper = dfit > (2048u / dth) ? 2048u / dth : dfit;                 // (at most 32 waves per CU)
            uint64_t dblocks = phk_div_up(sorted ? max_items : n, slots);
            const uint64_t dcap = (uint64_t)ctx->num_cus * (dper < 1 ? 1 : dper);
            if (dblocks > dcap) dblocks = dcap;
#define PHK_DIRECT1(K_, S_, T_, M_, O_)                                                                                            \
            PHK_LAUNCH(ctx, "phk_count_direct_kernel",                                                                           \
                       (phk_count_direct_kernel<K_, S_, T_, M_, O_><<<dim3((unsigned)dblocks), dim3(T_), dlds, ctx->stream>>>(    \
                           d_packed, M_ ? d_mask : nullptr, d_offsets, n, max_word, long_thr, d_counts, d_nwin, d_long_list, d_long_count, d_ord, piece_w, skip_plain, frag8, big8)))
#define PHK_DIRECT(K_, S_, T_)                                                                                                   \
            if (d_mask) {                                                                                                        \
                PHK_DIRECT1(K_, S_, T_, true, false);                                                                            \
                if (d_ord) { PHK_DIRECT1(K_, S_, T_, true, true); }                                                              \
            } else {                                                                                                             \
                if (!skip_plain) { PHK_DIRECT1(K_, S_, T_, false, false); }   /* (else: the pairs kernel's, launched above) */    \
                if (d_ord) { PHK_DIRECT1(K_, S_, T_, false, true); }                                                             \
            }                                                                                                                    \
            {                                                                                                                    \
                const int rc_ = launch_count_cfg<K_, PhkCountCfg<K_>::copies, PhkCountCfg<K_>::pack16>(                          \
                    ctx, d_packed, d_mask, d_offsets, n, max_word, d_counts, d_nwin, d_long_list, d_long_count, piece_w);         \
                if (rc_ == PHK_OK) ctx->ctl_dirty = false;                                                                       \
                return rc_;                                                                                                      \
            }
            if (k == 3) { PHK_DIRECT(3, 32, 512); }
            if (k == 4) { if (small) { PHK_DIRECT(4, 32, 512); } PHK_DIRECT(4, 32, 1024); }
            if (small) { PHK_DIRECT(5, 16, 512); }
            PHK_DIRECT(5, 16, 1024);
#undef PHK_DIRECT
#undef PHK_DIRECT1
        }
    }
    switch (k) {
        case 1: return launch_count_k<1>(ctx, d_packed, d_mask, d_offsets, n, max_word, d_counts, d_nwin);
        case 2: return launch_count_k<2>(ctx, d_packed, d_mask, d_offsets, n, max_word, d_counts, d_nwin);
        case 3: return launch_count_k<3>(ctx, d_packed, d_mask, d_offsets, n, max_word, d_counts, d_nwin);
        case 4: return launch_count_k<4>(ctx, d_packed, d_mask, d_offsets, n, max_word, d_counts, d_nwin);
        case 5: return launch_count_k<5>(ctx, d_packed, d_mask, d_offsets, n, max_word, d_counts, d_nwin);
        case 6: return launch_count_k<6>(ctx, d_packed, d_mask, d_offsets, n, max_word, d_counts, d_nwin);
        default: return launch_count_k<7>(ctx, d_packed, d_mask, d_offsets, n, max_word, d_counts, d_nwin);
    }
}

// ------------------------------------------------------------------------------------
// widen uint32 -> int64 (the reference returns NumPy's default int, scripts/kmer.py:46)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void phk_widen_kernel(const uint32_t *__restrict__ in, uint64_t count,
                                                        int64_t *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) out[i] = (int64_t)in[i];
}

int phk_launch_widen(phk_ctx *ctx, const uint32_t *d_in, uint64_t count, int64_t *d_out) {
    if (count == 0) return PHK_OK;
    uint64_t blocks = phk_div_up(count, 256);
    if (blocks > (uint64_t)ctx->num_cus * 16) blocks = (uint64_t)ctx->num_cus * 16;
    PHK_LAUNCH(ctx, "phk_widen_kernel",
               phk_widen_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(d_in, count, d_out));
    return PHK_OK;
}

// ------------------------------------------------------------------------------------
// normalise: one wavefront per row; out = (double)c / (double)rowsum  (0/0 -> NaN)
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void phk_normalize_int_kernel(const T *__restrict__ counts, uint64_t n,
                                                                uint64_t D, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t total = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n; r += total) {
        const T *row = counts + r * D;
        long long s = 0;  // exact: integer row sums of a count matrix are far below 2^63
        for (uint64_t j = lane; j < D; j += 64) s += (long long)row[j];
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh);
        const double ds = (double)s, ry = 1.0 / ds;
        for (uint64_t j = lane; j < D; j += 64) out[r * D + j] = phk_div_row((double)row[j], ds, ry);
    }
}

// float rows: the row sum follows NumPy's pairwise summation (what np.sum does on a
// contiguous float64 row: 8 running partial sums per <=128-element block, blocks combined by
// halving) so that renormalising float rows matches the reference bit for bit.
// (The halving is a recursion in NumPy; here its frames live in an explicit stack in LDS, one per wave, walked by lane 0:
// as a recursive device function it was the library's last kernel with a call stack in scratch memory.)
__device__ __forceinline__ double phk_np_pairwise_leaf(const double *a, uint64_t n) {   // n <= 128
    if (n < 8) {
        double r = 0.0;  // NumPy starts from the first element; 0.0 + a0 is exact
        for (uint64_t i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    uint64_t i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i];
    return res;
}

#define PW_DEPTH 48   // frames: a row of 2^40 elements halves 34 times
struct PwStack {
    uint64_t off[PW_DEPTH], len[PW_DEPTH];
    double left[PW_DEPTH];
    uint32_t state[PW_DEPTH];   // 0 new, 1 waiting for the left half, 2 waiting for the right half
};

__device__ double phk_np_pairwise_sum(const double *a, uint64_t n, PwStack &st) {
    int sp = 0;
    st.off[0] = 0; st.len[0] = n; st.state[0] = 0;
    double ret = 0.0;
    while (sp >= 0) {
        const uint64_t o = st.off[sp], m = st.len[sp];
        const uint32_t state = st.state[sp];
        if (state == 0) {
            if (m <= 128) {
                ret = phk_np_pairwise_leaf(a + o, m);
                --sp;
            } else {
                uint64_t n2 = m / 2;
                n2 -= n2 % 8;
                st.state[sp] = 1;
                ++sp;
                st.off[sp] = o; st.len[sp] = n2; st.state[sp] = 0;
            }
        } else if (state == 1) {
            uint64_t n2 = m / 2;
            n2 -= n2 % 8;
            st.left[sp] = ret;
            st.state[sp] = 2;
            ++sp;
            st.off[sp] = o + n2; st.len[sp] = m - n2; st.state[sp] = 0;
        } else {
            ret = st.left[sp] + ret;
            --sp;
        }
    }
    return ret;
}

__global__ __launch_bounds__(256) void phk_normalize_f64_kernel(const double *__restrict__ rows, uint64_t n,
                                                                uint64_t D, double *__restrict__ out) {
    __shared__ PwStack stacks[4];
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t total = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n; r += total) {
        const double *row = rows + r * D;
        double s = 0.0;
        if (lane == 0) s = phk_np_pairwise_sum(row, D, stacks[threadIdx.x >> 6]);
        s = __shfl(s, 0);
        for (uint64_t j = lane; j < D; j += 64) out[r * D + j] = row[j] / s;
    }
}

static unsigned norm_blocks(phk_ctx *ctx, uint64_t n) {
    uint64_t blocks = phk_div_up(n, 4);
    if (blocks > (uint64_t)ctx->num_cus * 8) blocks = (uint64_t)ctx->num_cus * 8;
    return (unsigned)blocks;
}

int phk_launch_normalize_u32(phk_ctx *ctx, const uint32_t *d_counts, uint64_t n, uint64_t D,
                             double *d_out) {
    if (n == 0 || D == 0) return PHK_OK;
    PHK_LAUNCH(ctx, "phk_normalize_int_kernel",
               phk_normalize_int_kernel<uint32_t><<<dim3(norm_blocks(ctx, n)), dim3(256), 0, ctx->stream>>>(
                   d_counts, n, D, d_out));
    return PHK_OK;
}

int phk_launch_normalize_i64(phk_ctx *ctx, const int64_t *d_counts, uint64_t n, uint64_t D,
                             double *d_out) {
    if (n == 0 || D == 0) return PHK_OK;
    PHK_LAUNCH(ctx, "phk_normalize_int_kernel",
               phk_normalize_int_kernel<int64_t><<<dim3(norm_blocks(ctx, n)), dim3(256), 0, ctx->stream>>>(
                   d_counts, n, D, d_out));
    return PHK_OK;
}

int phk_launch_normalize_f64(phk_ctx *ctx, const double *d_rows, uint64_t n, uint64_t D,
                             double *d_out) {
    if (n == 0 || D == 0) return PHK_OK;
    PHK_LAUNCH(ctx, "phk_normalize_f64_kernel",
               phk_normalize_f64_kernel<<<dim3(norm_blocks(ctx, n)), dim3(256), 0, ctx->stream>>>(
                   d_rows, n, D, d_out));
    return PHK_OK;
}

// ------------------------------------------------------------------------------------
// column gather: out[r][j] = in[r][perm[j]]  (reverse / complement / reverse-complement count vectors,
// scripts/transform_kmers.py:68-88)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void phk_permute_columns_kernel(const int64_t *__restrict__ in, uint64_t n, uint64_t D,
                                                                  const uint32_t *__restrict__ perm,
                                                                  int64_t *__restrict__ out) {
    const uint64_t total = n * D;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / D, j = i % D;
        out[i] = in[r * D + perm[j]];
    }
}

int phk_launch_permute_columns(phk_ctx *ctx, const int64_t *d_in, uint64_t n, uint64_t D, const uint32_t *d_perm,
                               int64_t *d_out) {
    if (n == 0 || D == 0) return PHK_OK;
    uint64_t blocks = phk_div_up(n * D, 256);
    if (blocks > (uint64_t)ctx->num_cus * 16) blocks = (uint64_t)ctx->num_cus * 16;
    PHK_LAUNCH(ctx, "phk_permute_columns_kernel",
               phk_permute_columns_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(d_in, n, D, d_perm, d_out));
    return PHK_OK;
}

// ------------------------------------------------------------------------------------
// verification helper: row sums against an expected value and word-wise comparison of two count matrices,
// both on the device (full-size batches are tens of GB: they are never brought to the host to be checked)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void phk_check_counts_kernel(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b,
                                                               uint64_t n, uint64_t D, unsigned long long expect,
                                                               unsigned long long *__restrict__ result) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t total = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    unsigned long long bad_rows = 0, bad_words = 0;
    for (uint64_t r = wave; r < n; r += total) {
        unsigned long long s = 0, diff = 0;
        for (uint64_t j = lane; j < D; j += 64) {
            const uint32_t x = a[r * D + j];
            s += x;
            if (b) diff += x != b[r * D + j];
        }
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) {
            s += __shfl_xor(s, sh);
            diff += __shfl_xor(diff, sh);
        }
        bad_rows += (expect != ~0ull && s != expect) ? 1 : 0;
        bad_words += diff;
    }
    if (lane == 0) {
        if (bad_rows) atomicAdd(result, bad_rows);
        if (bad_words) atomicAdd(result + 1, bad_words);
    }
}

int phk_launch_check_counts(phk_ctx *ctx, const uint32_t *d_counts, const uint32_t *d_other, uint64_t n, uint64_t D,
                            uint64_t expected_rowsum, uint64_t *d_result) {
    PHK_REQUIRE(d_result && (n == 0 || d_counts), "phk_check_counts_dev: NULL pointer");
    PHK_HIP(hipMemsetAsync(d_result, 0, 2 * sizeof(uint64_t), ctx->stream));
    if (n == 0 || D == 0) return PHK_OK;
    uint64_t blocks = phk_div_up(n, 4);
    if (blocks > (uint64_t)ctx->num_cus * 16) blocks = (uint64_t)ctx->num_cus * 16;
    PHK_LAUNCH(ctx, "phk_check_counts_kernel",
               phk_check_counts_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(
                   d_counts, d_other, n, D, (unsigned long long)expected_rowsum, (unsigned long long *)d_result));
    return PHK_OK;
}
