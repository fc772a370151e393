// batch.hip -- device-resident contig batches: the drop-in facade's data path.
//
// What a PhaMers user calls -- phamer_scorer.load_data + score_points on a FASTA input (scripts/phamer.py:131,139,
// 579), kmer.count_file (scripts/kmer.py:114-140) -- used to bounce every intermediate through host memory (ASCII
// up, int64 counts down, float64 rows up again, scores down).  A phk_batch keeps the path on the device: the
// sequence bytes go up ONCE (pinned in place, chunked, the copies overlapped with the packer on a second stream),
// counts and row sums stay in HBM, the length screen is a device row gather, and only what the caller asks for comes
// back: the scores, and the counts once for the features cache.
#include <string.h>

#include <chrono>
#include <functional>
#include <vector>

#include "phk_common.h"
#include "score_model.h"

struct phk_batch {
    uint64_t n = 0, D = 0, T = 0;
    int k = 0;
    uint32_t *d_counts = nullptr;   // [n][D]
    uint32_t *d_nwin = nullptr;     // [n] row sums (= counted windows)
    bool any_invalid = false;       // some base of the source batch was not one of the symbols (inherited by a selection:
                                    // "may hold invalid bases", not re-derived per row)
    std::vector<uint64_t> len;      // [n] bases per contig (host): what a selection's total_bases is summed from
};

#define BATCH_CHUNK PHK_STAGE_BYTES   // bases per upload chunk = one staging buffer (a multiple of 32: chunks pack independently)

static void batch_release(phk_batch *b) {
    if (!b) return;
    if (b->d_counts) (void)hipFree(b->d_counts);
    if (b->d_nwin) (void)hipFree(b->d_nwin);
    delete b;
}

// The sequence bytes of a batch -> device, packed, counted.  `bases` (when the caller holds them in one buffer) or `fill`
// (when they are produced chunk by chunk: phk_batch_from_fasta_file parses the file straight into the staging buffers)
// supplies the bytes: fill(o, len, dst) writes bases [o, o + len) of the concatenated sequences to dst; it is called for
// consecutive chunks in increasing order, from this thread.
int phk_batch_build(phk_ctx *ctx, const char *bases, const std::function<void(uint64_t, uint64_t, char *)> *fill,
                    const uint64_t *offsets, uint64_t n, int k, const char *symbols4, phk_batch **out) {
    PHK_ENTER(ctx, "phk_batch_build");   // (every caller's device work starts here: phk_batch_from_ascii, _from_fasta, _from_fasta_file)
    PHK_REQUIRE(out, "phk_batch_from_ascii: NULL out");
    PHK_REQUIRE(k >= 1, "phk_batch_from_ascii: k must be >= 1 (got %d)", k);
    if (k > PHK_MAX_K) {
        phk_set_error("phk_batch_from_ascii: k=%d is above PHK_MAX_K=%d", k, PHK_MAX_K);
        return PHK_ERR_UNSUPPORTED;
    }
    PHK_REQUIRE(n == 0 || offsets, "phk_batch_from_ascii: NULL offsets");
    const char *sym = symbols4 ? symbols4 : "ATGC";
    PHK_REQUIRE(strlen(sym) == 4, "phk_batch_from_ascii: symbols must be exactly 4 characters");
    if (n) {
        PHK_REQUIRE(offsets[0] == 0, "phk_batch_from_ascii: offsets[0] must be 0");
        for (uint64_t c = 0; c < n; ++c)
            PHK_REQUIRE(offsets[c + 1] >= offsets[c], "phk_batch_from_ascii: offsets must be non-decreasing");
    }
    const uint64_t T = n ? offsets[n] : 0;
    PHK_REQUIRE(T == 0 || bases || fill, "phk_batch_from_ascii: NULL bases");
    phk_batch *b = new phk_batch();
    b->n = n;
    b->k = k;
    b->D = phk_pow4(k);
    b->T = T;
    b->len.resize(n);
    for (uint64_t c = 0; c < n; ++c) b->len[c] = offsets[c + 1] - offsets[c];
    if (n == 0) {
        *out = b;
        return PHK_OK;
    }
    int rc = PHK_OK;
    hipStream_t copy_stream = nullptr;
    hipEvent_t copied[2] = {nullptr, nullptr}, packed_ev[2] = {nullptr, nullptr};
    void *d_chunk[2] = {nullptr, nullptr};
    auto body = [&]() -> int {
        if (hipMalloc(&b->d_counts, n * b->D * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc(&b->d_nwin, n * sizeof(uint32_t)) != hipSuccess) {
            phk_set_error("phk_batch: cannot allocate %llu x %llu counts on the device", (unsigned long long)n,
                          (unsigned long long)b->D);
            return PHK_ERR_NOMEM;
        }
        void *d_packed, *d_mask, *d_off, *d_flags;
        PHK_TRY(phk_ws(ctx, WS_PACKED, (phk_div_up(T, 16) + 1) * 4, &d_packed));
        PHK_TRY(phk_ws(ctx, WS_MASK, (phk_div_up(T, 32) + 1) * 4, &d_mask));
        PHK_TRY(phk_ws(ctx, WS_OFFSETS, (n + 1) * 8, &d_off));
        const uint64_t nchunks = T ? phk_div_up(T, BATCH_CHUNK) : 0;
        PHK_TRY(phk_ws(ctx, WS_FLAGS, (nchunks + 16) * 4, &d_flags));
        PHK_HIP(hipMemcpyAsync(d_off, offsets, (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        if (T) {
            const uint64_t chunk_bytes = T < BATCH_CHUNK ? ((T + 63) & ~63ull) : BATCH_CHUNK;
            for (int i = 0; i < (nchunks > 1 ? 2 : 1); ++i)
                if (hipMalloc(&d_chunk[i], chunk_bytes) != hipSuccess) return PHK_ERR_NOMEM;
            // A multi-chunk upload goes through two pinned staging buffers of the context, filled by host threads while
            // the previous chunk is on the bus.  Not hipHostRegister on the caller's buffer, and not a copy straight from
            // it either (the runtime then pins the pages itself): a 5 GB buffer the device has once been given costs
            // 0.26-0.6 s to free afterwards instead of 0.05 (measured, tools/diag/fasta_free_time.py), with every HIP call
            // of the process waiting meanwhile.
            const bool staged = nchunks > 1 || !bases;
            if (staged) PHK_TRY(phk_stage_ensure(ctx));
            PHK_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
            for (int i = 0; i < 2; ++i) {
                PHK_HIP(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming));
                PHK_HIP(hipEventCreateWithFlags(&packed_ev[i], hipEventDisableTiming));
            }
            for (uint64_t c = 0; c < nchunks; ++c) {
                const int s = (int)(c & 1);
                const uint64_t o = c * BATCH_CHUNK, len = T - o < BATCH_CHUNK ? T - o : BATCH_CHUNK;
                if (c >= 2) PHK_HIP(hipStreamWaitEvent(copy_stream, packed_ev[s], 0));   // the packer is done with this buffer
                const char *src = bases ? bases + o : nullptr;
                if (staged) {
                    if (c >= 2) PHK_HIP(hipEventSynchronize(copied[s]));   // the bus is done with this staging buffer
                    char *dst = (char *)ctx->stage[s];
                    if (fill) {
                        (*fill)(o, len, dst);
                    } else {
                        const uint64_t piece = 1ull << 20;
                        phk_parallel_for(phk_div_up(len, piece), [&](uint64_t i) {
                            const uint64_t a = i * piece, m = len - a < piece ? len - a : piece;
                            memcpy(dst + a, src + a, m);
                        });
                    }
                    src = dst;
                }
                PHK_HIP(hipMemcpyAsync(d_chunk[s], src, len, hipMemcpyHostToDevice, copy_stream));
                PHK_HIP(hipEventRecord(copied[s], copy_stream));
                PHK_HIP(hipStreamWaitEvent(ctx->stream, copied[s], 0));
                PHK_TRY(phk_launch_pack(ctx, (const char *)d_chunk[s], len, sym, (uint32_t *)d_packed + o / 16,
                                        (uint32_t *)d_mask + o / 32, (uint32_t *)d_flags + c));
                PHK_HIP(hipEventRecord(packed_ev[s], ctx->stream));
            }
        }
        std::vector<uint32_t> flags(nchunks, 0);
        if (nchunks) PHK_HIP(hipMemcpyAsync(flags.data(), d_flags, nchunks * 4, hipMemcpyDeviceToHost, ctx->stream));
        PHK_HIP(hipStreamSynchronize(ctx->stream));
        for (uint32_t f : flags) b->any_invalid = b->any_invalid || f != 0;
        PHK_TRY(phk_launch_count(ctx, (const uint32_t *)d_packed, b->any_invalid ? (const uint32_t *)d_mask : nullptr, T,
                                 (const uint64_t *)d_off, n, k, b->d_counts, b->d_nwin));
        PHK_HIP(hipStreamSynchronize(ctx->stream));
        return PHK_OK;
    };
    rc = body();
    if (copy_stream) {
        (void)hipStreamSynchronize(copy_stream);
        (void)hipStreamDestroy(copy_stream);
    }
    for (int i = 0; i < 2; ++i) {
        if (copied[i]) (void)hipEventDestroy(copied[i]);
        if (packed_ev[i]) (void)hipEventDestroy(packed_ev[i]);
        if (d_chunk[i]) (void)hipFree(d_chunk[i]);
    }
    if (rc != PHK_OK) {
        batch_release(b);
        return rc;
    }
    *out = b;
    return PHK_OK;
}

// The batch of a FASTA file from its RAW bytes (phk_batch_from_fasta_file / _part since round 5): the file goes up as it is --
// host threads only copy it into the pinned staging buffers, at memory speed, while the previous chunk is on the bus -- and
// phk_deline_pack_kernel reads the sequences out of it on the device.  The raw buffer (the file's size) lives on the device
// until the packed stream exists.
int phk_raw_to_device(phk_ctx *ctx, const char *raw, uint64_t raw_bytes, const uint8_t **d_raw) {
    PHK_ENTER(ctx, "phk_raw_to_device");
    void *d = nullptr;
    PHK_TRY(phk_ws(ctx, WS_ASCII, raw_bytes + 64, &d));
    if (raw_bytes) {
        PHK_TRY(phk_copy_to_device(ctx, d, raw, raw_bytes));
        PHK_HIP(hipStreamSynchronize(ctx->stream));
    }
    *d_raw = (const uint8_t *)d;
    return PHK_OK;
}

int phk_batch_build_raw(phk_ctx *ctx, const uint8_t *d_raw, const char *side, uint64_t side_bytes,
                        const uint64_t *rbegin, const uint32_t *rlw, const uint32_t *rtl, const uint64_t *offsets, uint64_t n,
                        int k, const char *symbols4, phk_batch **out) {
    PHK_ENTER(ctx, "phk_batch_build_raw");
    PHK_REQUIRE(out, "phk_batch: NULL out");
    PHK_REQUIRE(k >= 1, "phk_batch: k must be >= 1 (got %d)", k);
    if (k > PHK_MAX_K) {
        phk_set_error("phk_batch: k=%d is above PHK_MAX_K=%d", k, PHK_MAX_K);
        return PHK_ERR_UNSUPPORTED;
    }
    const char *sym = symbols4 ? symbols4 : "ATGC";
    PHK_REQUIRE(strlen(sym) == 4, "phk_batch: symbols must be exactly 4 characters");
    const uint64_t T = n ? offsets[n] : 0;
    phk_batch *b = new phk_batch();
    b->n = n;
    b->k = k;
    b->D = phk_pow4(k);
    b->T = T;
    b->len.resize(n);
    for (uint64_t c = 0; c < n; ++c) b->len[c] = offsets[c + 1] - offsets[c];
    if (n == 0) {
        *out = b;
        return PHK_OK;
    }
    void *d_meta = nullptr;
    const bool timing = getenv("PHK_INGEST_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (timing) {
            (void)hipStreamSynchronize(ctx->stream);
            fprintf(stderr, "[phk ingest]   %-26s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
    };
    auto body = [&]() -> int {
        if (hipMalloc(&b->d_counts, n * b->D * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc(&b->d_nwin, n * sizeof(uint32_t)) != hipSuccess) {
            phk_set_error("phk_batch: cannot allocate %llu x %llu counts on the device", (unsigned long long)n,
                          (unsigned long long)b->D);
            return PHK_ERR_NOMEM;
        }
        void *d_packed, *d_mask, *d_off, *d_flags;
        PHK_TRY(phk_ws(ctx, WS_PACKED, (phk_div_up(T, 16) + 1) * 4, &d_packed));
        PHK_TRY(phk_ws(ctx, WS_MASK, (phk_div_up(T, 32) + 1) * 4, &d_mask));
        PHK_TRY(phk_ws(ctx, WS_OFFSETS, (n + 1) * 8, &d_off));
        PHK_TRY(phk_ws(ctx, WS_FLAGS, 64, &d_flags));
        if (hipMalloc(&d_meta, n * 16 + side_bytes + 64) != hipSuccess) {
            phk_set_error("phk_batch: cannot allocate %llu bytes for the record layout on the device", (unsigned long long)(n * 16 + side_bytes));
            return PHK_ERR_NOMEM;
        }
        lap("device allocations");
        uint64_t *d_rbegin = (uint64_t *)d_meta;
        uint32_t *d_rlw = (uint32_t *)(d_rbegin + n), *d_rtl = d_rlw + n;
        uint8_t *d_side = (uint8_t *)(d_rtl + n);
        PHK_HIP(hipMemcpyAsync(d_off, offsets, (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        PHK_HIP(hipMemcpyAsync(d_rbegin, rbegin, n * 8, hipMemcpyHostToDevice, ctx->stream));
        PHK_HIP(hipMemcpyAsync(d_rlw, rlw, n * 4, hipMemcpyHostToDevice, ctx->stream));
        PHK_HIP(hipMemcpyAsync(d_rtl, rtl, n * 4, hipMemcpyHostToDevice, ctx->stream));
        if (side_bytes) PHK_TRY(phk_copy_to_device(ctx, d_side, side, side_bytes));
        lap("layout + side on the device");
        PHK_TRY(phk_launch_deline_pack(ctx, d_raw, d_side, (const uint64_t *)d_off, n, d_rbegin, d_rlw, d_rtl, T, sym,
                                       (uint32_t *)d_packed, (uint32_t *)d_mask, (uint32_t *)d_flags));
        uint32_t flag = 0;
        PHK_HIP(hipMemcpyAsync(&flag, d_flags, 4, hipMemcpyDeviceToHost, ctx->stream));
        PHK_HIP(hipStreamSynchronize(ctx->stream));
        b->any_invalid = flag != 0;
        lap("de-lined and packed");
        PHK_TRY(phk_launch_count(ctx, (const uint32_t *)d_packed, b->any_invalid ? (const uint32_t *)d_mask : nullptr, T,
                                 (const uint64_t *)d_off, n, k, b->d_counts, b->d_nwin));
        PHK_HIP(hipStreamSynchronize(ctx->stream));
        lap("counted");
        return PHK_OK;
    };
    const int rc = body();
    if (rc != PHK_OK) (void)hipStreamSynchronize(ctx->stream);
    if (d_meta) (void)hipFree(d_meta);
    lap("layout freed");
    if (rc != PHK_OK) {
        batch_release(b);
        return rc;
    }
    *out = b;
    return PHK_OK;
}

extern "C" int phk_batch_from_ascii(phk_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n, int k,
                                    const char *symbols4, phk_batch **out) {
    PHK_ENTER(ctx, "phk_batch_from_ascii");
    return phk_batch_build(ctx, bases, nullptr, offsets, n, k, symbols4, out);
}

__global__ void phk_rowsum_kernel(const uint32_t *__restrict__ counts, uint64_t N, uint64_t D, uint32_t *__restrict__ out);

// A batch from a count matrix the host already holds -- the features cache of an earlier run, read back by
// fileIO.read_feature_file (scripts/phamer.py:132-136): int64 counts [n][4^k] go up once, narrowed to uint32 on the way into
// the pinned staging buffers, row sums are formed on the device, and the run scores from the same resident integers as a
// run that counted the FASTA.  PHK_ERR_UNSUPPORTED when D is not 4^k (k <= PHK_MAX_K) or an entry is negative / >= 2^32
// (the facade then keeps the reference's float rows).
extern "C" int phk_batch_from_counts(phk_ctx *ctx, const int64_t *counts, uint64_t n, uint64_t D, phk_batch **out) {
    PHK_ENTER(ctx, "phk_batch_from_counts");
    PHK_REQUIRE(out && (n == 0 || counts), "phk_batch_from_counts: NULL");
    int k = 0;
    while (k <= PHK_MAX_K && phk_pow4(k) != D) ++k;
    if (k < 1 || k > PHK_MAX_K) {
        phk_set_error("phk_batch_from_counts: %llu columns is not 4^k for 1 <= k <= %d", (unsigned long long)D, PHK_MAX_K);
        return PHK_ERR_UNSUPPORTED;
    }
    phk_batch *b = new phk_batch();
    b->n = n;
    b->k = k;
    b->D = D;
    b->len.assign(n, 0);
    if (n == 0) {
        *out = b;
        return PHK_OK;
    }
    int rc = PHK_OK;
    hipEvent_t done[2] = {nullptr, nullptr};
    auto body = [&]() -> int {
        if (hipMalloc(&b->d_counts, n * D * sizeof(uint32_t)) != hipSuccess || hipMalloc(&b->d_nwin, n * sizeof(uint32_t)) != hipSuccess) {
            phk_set_error("phk_batch: cannot allocate %llu x %llu counts on the device", (unsigned long long)n, (unsigned long long)D);
            return PHK_ERR_NOMEM;
        }
        PHK_TRY(phk_stage_ensure(ctx));
        for (int i = 0; i < 2; ++i) PHK_HIP(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        const uint64_t total = n * D, per = BATCH_CHUNK / sizeof(uint32_t), nchunks = phk_div_up(total, per);
        std::vector<uint8_t> bad(nchunks, 0);
        for (uint64_t c = 0; c < nchunks; ++c) {
            const int s = (int)(c & 1);
            const uint64_t o = c * per, m = total - o < per ? total - o : per;
            if (c >= 2) PHK_HIP(hipEventSynchronize(done[s]));   // the bus is done with this staging buffer
            uint32_t *dst = (uint32_t *)ctx->stage[s];
            const int64_t *src = counts + o;
            const uint64_t piece = 1ull << 18;
            std::vector<uint8_t> pbad(phk_div_up(m, piece), 0);
            phk_parallel_for(phk_div_up(m, piece), [&](uint64_t i) {
                const uint64_t a = i * piece, e = a + piece < m ? a + piece : m;
                uint64_t any = 0;
                for (uint64_t j = a; j < e; ++j) {
                    const uint64_t v = (uint64_t)src[j];
                    any |= v >> 32;            // (negative or >= 2^32)
                    dst[j] = (uint32_t)v;
                }
                pbad[i] = any != 0;
            });
            for (uint8_t f : pbad) bad[c] |= f;
            if (bad[c]) {
                phk_set_error("phk_batch_from_counts: an entry is negative or does not fit 32 bits");
                return PHK_ERR_UNSUPPORTED;
            }
            PHK_HIP(hipMemcpyAsync(b->d_counts + o, dst, m * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            PHK_HIP(hipEventRecord(done[s], ctx->stream));
        }
        for (uint64_t r0 = 0; r0 < n; r0 += 1ull << 24) {   // (one wave per row: 2^24 rows per launch keep the grid below 2^32 threads)
            const uint64_t nr = n - r0 < (1ull << 24) ? n - r0 : 1ull << 24;
            PHK_LAUNCH(ctx, "phk_rowsum_kernel",
                       phk_rowsum_kernel<<<dim3((unsigned)phk_div_up(nr, 4)), dim3(256), 0, ctx->stream>>>(b->d_counts + r0 * D, nr, D, b->d_nwin + r0));
        }
        PHK_HIP(hipStreamSynchronize(ctx->stream));
        return PHK_OK;
    };
    rc = body();
    if (rc != PHK_OK) (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 2; ++i)
        if (done[i]) (void)hipEventDestroy(done[i]);
    if (rc != PHK_OK) {
        batch_release(b);
        return rc;
    }
    *out = b;
    return PHK_OK;
}

extern "C" int phk_batch_shape(const phk_batch *b, uint64_t *n, uint64_t *D, uint64_t *total_bases, int *any_invalid) {
    PHK_REQUIRE(b, "phk_batch_shape: NULL");
    if (n) *n = b->n;
    if (D) *D = b->D;
    if (total_bases) *total_bases = b->T;
    if (any_invalid) *any_invalid = b->any_invalid ? 1 : 0;
    return PHK_OK;
}

extern "C" int phk_batch_device_ptrs(const phk_batch *b, const uint32_t **d_counts, const uint32_t **d_rowsums) {
    PHK_REQUIRE(b, "phk_batch_device_ptrs: NULL");
    if (d_counts) *d_counts = b->d_counts;
    if (d_rowsums) *d_rowsums = b->d_nwin;
    return PHK_OK;
}

extern "C" int phk_batch_counts_i64(phk_ctx *ctx, const phk_batch *b, int64_t *counts) {
    PHK_ENTER(ctx, "phk_batch_counts_i64");
    PHK_REQUIRE(b && (b->n == 0 || counts), "phk_batch_counts_i64: NULL");
    // widen on the device in slices so that the device buffer stays small (four staging chunks each: the copy pipelines)
    const uint64_t rows_per = b->D ? ((256ull << 20) / (b->D * 8) > 0 ? (256ull << 20) / (b->D * 8) : 1) : 1;
    void *d_wide;
    PHK_TRY(phk_ws(ctx, WS_WIDE, (rows_per < b->n ? rows_per : b->n) * b->D * 8, &d_wide));
    for (uint64_t r = 0; r < b->n; r += rows_per) {
        const uint64_t m = b->n - r < rows_per ? b->n - r : rows_per;
        PHK_TRY(phk_launch_widen(ctx, b->d_counts + r * b->D, m * b->D, (int64_t *)d_wide));
        PHK_TRY(phk_copy_to_host(ctx, counts + r * b->D, d_wide, m * b->D * 8));
    }
    return PHK_OK;
}

extern "C" int phk_batch_counts_u32(phk_ctx *ctx, const phk_batch *b, uint32_t *counts) {
    PHK_ENTER(ctx, "phk_batch_counts_u32");
    PHK_REQUIRE(b && (b->n == 0 || counts), "phk_batch_counts_u32: NULL");
    if (b->n == 0) return PHK_OK;
    return phk_copy_to_host(ctx, counts, b->d_counts, b->n * b->D * sizeof(uint32_t));
}

extern "C" int phk_batch_normalized(phk_ctx *ctx, const phk_batch *b, double *rows) {
    PHK_ENTER(ctx, "phk_batch_normalized");
    PHK_REQUIRE(b && (b->n == 0 || rows), "phk_batch_normalized: NULL");
    const uint64_t rows_per = b->D ? ((256ull << 20) / (b->D * 8) > 0 ? (256ull << 20) / (b->D * 8) : 1) : 1;
    void *d_q;
    PHK_TRY(phk_ws(ctx, WS_Q64, (rows_per < b->n ? rows_per : b->n) * b->D * 8, &d_q));
    for (uint64_t r = 0; r < b->n; r += rows_per) {
        const uint64_t m = b->n - r < rows_per ? b->n - r : rows_per;
        PHK_TRY(phk_launch_normalize_u32(ctx, b->d_counts + r * b->D, m, b->D, (double *)d_q));
        PHK_TRY(phk_copy_to_host(ctx, rows + r * b->D, d_q, m * b->D * 8));
    }
    return PHK_OK;
}

__global__ __launch_bounds__(256) void phk_gather_rows_kernel(const uint32_t *__restrict__ counts,
                                                              const uint32_t *__restrict__ nwin,
                                                              const uint64_t *__restrict__ rows, uint64_t m, uint64_t D,
                                                              uint32_t *__restrict__ out_counts,
                                                              uint32_t *__restrict__ out_nwin) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t total = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < m; i += total) {
        const uint64_t r = rows[i];
        const uint4 *src = reinterpret_cast<const uint4 *>(counts + r * D);
        uint4 *dst = reinterpret_cast<uint4 *>(out_counts + i * D);
        for (uint64_t j = lane; j < D / 4; j += 64) dst[j] = src[j];
        for (uint64_t j = (D / 4) * 4 + lane; j < D; j += 64) out_counts[i * D + j] = counts[r * D + j];
        if (lane == 0) out_nwin[i] = nwin[r];
    }
}

// the rows `rows[0..m)` of a batch as a new batch (phamer_scorer.screen_by_length's row filter, scripts/phamer.py:
// 144-157, on the device)
extern "C" int phk_batch_select(phk_ctx *ctx, const phk_batch *b, const uint64_t *rows, uint64_t m, phk_batch **out) {
    PHK_ENTER(ctx, "phk_batch_select");
    PHK_REQUIRE(b && out && (m == 0 || rows), "phk_batch_select: NULL");
    for (uint64_t i = 0; i < m; ++i) PHK_REQUIRE(rows[i] < b->n, "phk_batch_select: row %llu out of range", (unsigned long long)rows[i]);
    phk_batch *s = new phk_batch();
    s->n = m;
    s->k = b->k;
    s->D = b->D;
    s->any_invalid = b->any_invalid;
    s->len.resize(m);
    for (uint64_t i = 0; i < m; ++i) {
        s->len[i] = b->len[rows[i]];
        s->T += s->len[i];
    }
    if (m == 0) {
        *out = s;
        return PHK_OK;
    }
    void *d_rows;
    int rc = phk_ws(ctx, WS_OFFSETS, m * 8, &d_rows);
    if (rc == PHK_OK && (hipMalloc(&s->d_counts, m * s->D * sizeof(uint32_t)) != hipSuccess ||
                         hipMalloc(&s->d_nwin, m * sizeof(uint32_t)) != hipSuccess))
        rc = PHK_ERR_NOMEM;
    if (rc != PHK_OK) {
        batch_release(s);
        return rc;
    }
    auto body = [&]() -> int {
        PHK_HIP(hipMemcpyAsync(d_rows, rows, m * 8, hipMemcpyHostToDevice, ctx->stream));
        uint64_t blocks = phk_div_up(m, 4);
        if (blocks > (uint64_t)ctx->num_cus * 16) blocks = (uint64_t)ctx->num_cus * 16;
        PHK_LAUNCH(ctx, "phk_gather_rows_kernel",
                   phk_gather_rows_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(
                       b->d_counts, b->d_nwin, (const uint64_t *)d_rows, m, b->D, s->d_counts, s->d_nwin));
        PHK_HIP(hipStreamSynchronize(ctx->stream));
        return PHK_OK;
    };
    rc = body();
    if (rc != PHK_OK) {
        batch_release(s);
        return rc;
    }
    *out = s;
    return PHK_OK;
}

// ---- count-vector transforms and per-file column sums on resident counts ----
// Column sums of a batch (kmer.count_directory's per-file np.sum(file_counts, axis=0), scripts/kmer.py:170-173): a block
// sums a stripe of rows with one thread per column (consecutive threads = consecutive columns: coalesced) and adds its
// partial sums to the [D] result.
__global__ __launch_bounds__(256) void phk_column_sums_kernel(const uint32_t *__restrict__ counts, uint64_t n, uint64_t D,
                                                              uint64_t rows_per_block, unsigned long long *__restrict__ sums) {
    const uint64_t col = (uint64_t)blockIdx.y * 256 + threadIdx.x;
    if (col >= D) return;
    const uint64_t r0 = (uint64_t)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    unsigned long long acc = 0;
    for (uint64_t r = r0; r < r1; ++r) acc += counts[r * D + col];
    if (acc) atomicAdd(sums + col, acc);
}

extern "C" int phk_batch_column_sums(phk_ctx *ctx, const phk_batch *b, int64_t *sums) {
    PHK_ENTER(ctx, "phk_batch_column_sums");
    PHK_REQUIRE(b && sums, "phk_batch_column_sums: NULL");
    const uint64_t D = b->D;
    void *d_s;
    PHK_TRY(phk_ws(ctx, WS_WIDE, D * 8, &d_s));
    PHK_HIP(hipMemsetAsync(d_s, 0, D * 8, ctx->stream));
    if (b->n) {
        uint64_t stripes = (uint64_t)ctx->num_cus * 8 / phk_div_up(D, 256);
        stripes = stripes < 1 ? 1 : (stripes > b->n ? b->n : stripes);
        const uint64_t rows_per = phk_div_up(b->n, stripes);
        PHK_LAUNCH(ctx, "phk_column_sums_kernel",
                   phk_column_sums_kernel<<<dim3((unsigned)phk_div_up(b->n, rows_per), (unsigned)phk_div_up(D, 256)), dim3(256), 0, ctx->stream>>>(
                       b->d_counts, b->n, D, rows_per, (unsigned long long *)d_s));
    }
    PHK_HIP(hipMemcpyAsync(sums, d_s, D * 8, hipMemcpyDeviceToHost, ctx->stream));
    PHK_HIP(hipStreamSynchronize(ctx->stream));
    return PHK_OK;
}

// out[r][j] = counts[r][table[j]] for every row of a resident batch, and the new rows' sums (the reference's tables are
// not permutations, so a row's sum can change): one wave per row, the row read once into LDS.
__global__ __launch_bounds__(256) void phk_gather_columns_kernel(const uint32_t *__restrict__ counts, uint64_t n, uint64_t D,
                                                                 const uint32_t *__restrict__ table,
                                                                 uint32_t *__restrict__ out, uint32_t *__restrict__ out_nwin) {
    extern __shared__ uint32_t gc_row[];   // one row of D per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t nw = blockDim.x >> 6;
    uint32_t *row = gc_row + (uint64_t)wave * D;
    const uint64_t total = (uint64_t)gridDim.x * nw;
    for (uint64_t r = (uint64_t)blockIdx.x * nw + wave; r < n; r += total) {
        for (uint64_t j = lane; j < D; j += 64) row[j] = counts[r * D + j];
        __builtin_amdgcn_wave_barrier();
        uint32_t s = 0;
        for (uint64_t j = lane; j < D; j += 64) {
            const uint32_t v = row[table[j]];
            out[r * D + j] = v;
            s += v;
        }
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh);
        if (lane == 0) out_nwin[r] = s;
        __builtin_amdgcn_wave_barrier();
    }
}

// transform_kmers.transform_kmers (scripts/transform_kmers.py:68-88) on a resident batch: a new batch whose column j is
// the source's column table[j] (host table, D entries, each < D).
extern "C" int phk_batch_gather_columns(phk_ctx *ctx, const phk_batch *b, const uint32_t *table, phk_batch **out) {
    PHK_ENTER(ctx, "phk_batch_gather_columns");
    PHK_REQUIRE(b && table && out, "phk_batch_gather_columns: NULL");
    const uint64_t D = b->D;
    for (uint64_t j = 0; j < D; ++j)
        PHK_REQUIRE(table[j] < D, "phk_batch_gather_columns: index %u is out of bounds for %llu columns", table[j], (unsigned long long)D);
    phk_batch *s = new phk_batch();
    s->n = b->n; s->k = b->k; s->D = D; s->T = b->T; s->any_invalid = b->any_invalid; s->len = b->len;
    if (b->n == 0) {
        *out = s;
        return PHK_OK;
    }
    void *d_tab;
    int rc = phk_ws(ctx, WS_OFFSETS, D * 4, &d_tab);
    if (rc == PHK_OK && (hipMalloc(&s->d_counts, s->n * D * sizeof(uint32_t)) != hipSuccess ||
                         hipMalloc(&s->d_nwin, s->n * sizeof(uint32_t)) != hipSuccess))
        rc = PHK_ERR_NOMEM;
    auto body = [&]() -> int {
        PHK_HIP(hipMemcpyAsync(d_tab, table, D * 4, hipMemcpyHostToDevice, ctx->stream));
        const uint64_t nw = 4 * D * sizeof(uint32_t) <= 65536 ? 4 : 1;   // waves (rows in LDS) per block: 64 KiB of dynamic LDS
        uint64_t blocks = phk_div_up(s->n, nw);
        if (blocks > (uint64_t)ctx->num_cus * 16) blocks = (uint64_t)ctx->num_cus * 16;
        PHK_LAUNCH(ctx, "phk_gather_columns_kernel",
                   phk_gather_columns_kernel<<<dim3((unsigned)blocks), dim3((unsigned)(64 * nw)), nw * D * sizeof(uint32_t), ctx->stream>>>(
                       b->d_counts, s->n, D, (const uint32_t *)d_tab, s->d_counts, s->d_nwin));
        PHK_HIP(hipStreamSynchronize(ctx->stream));
        return PHK_OK;
    };
    if (rc == PHK_OK) rc = body();
    if (rc != PHK_OK) {
        batch_release(s);
        return rc;
    }
    *out = s;
    return PHK_OK;
}

// phamer_scorer.score_points on a device-resident batch (scripts/phamer.py:177-195): scores[n] to the host.
// A zero-count row (the reference's NaN row) makes the call fail with PHK_ERR_NAN, as phk_score does.
extern "C" int phk_batch_score(phk_ctx *ctx, const phk_model *model, const phk_batch *b, int method, double *scores) {
    PHK_ENTER(ctx, "phk_batch_score");
    PHK_REQUIRE(model && b, "phk_batch_score: NULL model/batch");
    PHK_REQUIRE(b->D == model->D, "phk_batch_score: batch has %llu columns, model %llu", (unsigned long long)b->D,
                (unsigned long long)model->D);
    if (b->n == 0) return PHK_OK;
    PHK_REQUIRE(scores, "phk_batch_score: NULL scores");
    void *d_s, *d_flags;
    PHK_TRY(phk_ws(ctx, WS_OUT, b->n * 8 + 64, &d_s));
    PHK_TRY(phk_ws(ctx, WS_FLAGS, 64, &d_flags));
    PHK_TRY(phk_score_rows(ctx, model, nullptr, b->d_counts, b->d_nwin, b->n, method, (double *)d_s, (uint32_t *)d_flags));
    uint32_t nan_rows = 0;
    PHK_HIP(hipMemcpyAsync(&nan_rows, d_flags, 4, hipMemcpyDeviceToHost, ctx->stream));
    PHK_HIP(hipMemcpyAsync(scores, d_s, b->n * 8, hipMemcpyDeviceToHost, ctx->stream));
    PHK_HIP(hipStreamSynchronize(ctx->stream));
    if (nan_rows) {
        phk_set_error("phk_batch_score: %u row(s) have no counted window (NaN after normalisation)", nan_rows);
        return PHK_ERR_NAN;
    }
    return PHK_OK;
}

extern "C" int phk_batch_free(phk_ctx *ctx, phk_batch *b) {
    if (!b) return PHK_OK;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    batch_release(b);
    return PHK_OK;
}
