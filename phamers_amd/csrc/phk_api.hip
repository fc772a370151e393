// phk_api.hip -- the extern "C" surface of libphamers_hip.so (include/phamers_hip.h):
// context / workspace / timing plumbing and the host- and device-pointer entry points.
#include <stdlib.h>
#include <string.h>

#include "phk_common.h"
#include "score_model.h"

static thread_local char g_err[512] = "";

void phk_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *phk_last_error(void) { return g_err; }
// A diagnostic build (-DPHK_DIAGNOSTIC_BUILD: the only switch under which the timer / probe code of the kernels compiles)
// reports a different version, so such a library cannot pass __graft_entry__.build() or be loaded by phamers_amd._lib.
#ifdef PHK_DIAGNOSTIC_BUILD
extern "C" int phk_abi_version(void) { return PHK_ABI_VERSION | 0x40000000; }
#else
extern "C" int phk_abi_version(void) { return PHK_ABI_VERSION; }
#endif

extern "C" int phk_device_count(int *count) {
    PHK_REQUIRE(count, "phk_device_count: NULL");
    PHK_HIP(hipGetDeviceCount(count));
    return PHK_OK;
}

// one knob by name (the part after PHK_ of its environment variable, lower case)
static int set_knob(PhkKnobs &k, const char *key, const char *value) {
    const char *v = value ? value : "";
    if (!strcmp(key, "count_lanes")) k.count_lanes = v[0];
    else if (!strcmp(key, "force_exact")) k.force_exact = v[0] == '1';
    else if (!strcmp(key, "proposal")) snprintf(k.proposal, sizeof(k.proposal), "%s", v);
    else if (!strcmp(key, "cx_cfg")) snprintf(k.cx_cfg, sizeof(k.cx_cfg), "%s", v);
    else if (!strcmp(key, "rerank")) k.rerank = v[0];
    else if (!strcmp(key, "count_sort")) k.count_sort = v[0] != '0';
    else if (!strcmp(key, "score_batch")) k.score_batch = strtoull(v, nullptr, 10);
    else if (!strcmp(key, "tail_aside")) k.tail_aside = v[0] != '0';
    else if (!strcmp(key, "ws_fail")) k.ws_fail = v[0] ? atoi(v) : 0;
    else if (!strcmp(key, "gen_groups")) k.gen_groups = v[0] ? atoi(v) : 0;
    else if (!strcmp(key, "gen_seq")) k.gen_seq = v[0] == '1';
    else if (!strcmp(key, "i8_insert")) k.i8_insert = (v[0] >= '0' && v[0] <= '2') ? v[0] : 0;
    else return PHK_ERR_ARG;
    return PHK_OK;
}

static void knobs_from_env(PhkKnobs &k) {
    static const char *const names[][2] = {{"count_lanes", "PHK_COUNT_LANES"}, {"force_exact", "PHK_FORCE_EXACT"},
                                           {"proposal", "PHK_PROPOSAL"}, {"cx_cfg", "PHK_CX_CFG"},
                                           {"rerank", "PHK_RERANK"}, {"count_sort", "PHK_COUNT_SORT"},
                                           {"score_batch", "PHK_SCORE_BATCH"}, {"tail_aside", "PHK_TAIL_ASIDE"},
                                           {"gen_groups", "PHK_GEN_GROUPS"}, {"gen_seq", "PHK_GEN_SEQ"}, {"i8_insert", "PHK_I8_INSERT"}};
    for (auto &n : names) {
        const char *e = getenv(n[1]);
        if (e) (void)set_knob(k, n[0], e);
    }
}

extern "C" int phk_set_option(phk_ctx *ctx, const char *key, const char *value) {
    PHK_REQUIRE(ctx && key, "phk_set_option: NULL");
    if (set_knob(ctx->knobs, key, value) != PHK_OK) {
        phk_set_error("phk_set_option: unknown option '%s'", key);
        return PHK_ERR_ARG;
    }
    return PHK_OK;
}

extern "C" int phk_create(int device_id, void *stream, phk_ctx **out) {
    PHK_REQUIRE(out, "phk_create: NULL out");
    int ndev = 0;
    PHK_HIP(hipGetDeviceCount(&ndev));
    PHK_REQUIRE(device_id >= 0 && device_id < ndev, "phk_create: device %d not present (%d devices)",
                device_id, ndev);
    PHK_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    PHK_HIP(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        phk_set_error("phk_create: device %d is %s; this library is built for gfx950 only", device_id,
                      prop.gcnArchName);
        return PHK_ERR_UNSUPPORTED;
    }
    phk_ctx *ctx = new phk_ctx();
    ctx->device = device_id;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    knobs_from_env(ctx->knobs);
    // kernel attributes are per device: set them for THIS context's device now, not lazily at a first launch
    {
        int rc = phk_count_init_device(ctx);
        if (rc == PHK_OK) rc = phk_score_f16_init_device(ctx);
        if (rc == PHK_OK) rc = phk_score_i8_init_device(ctx);
        if (rc == PHK_OK) rc = phk_score_mfma_init_device(ctx);
        if (rc != PHK_OK) {
            delete ctx;
            return rc;
        }
    }
    if (stream) {
        ctx->stream = (hipStream_t)stream;
        ctx->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            phk_set_error("phk_create: hipStreamCreate failed: %s", hipGetErrorString(e));
            delete ctx;
            return PHK_ERR_HIP;
        }
        ctx->own_stream = true;
    }
    *out = ctx;
    return PHK_OK;
}

extern "C" int phk_destroy(phk_ctx *ctx) {
    if (!ctx) return PHK_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < WS_SLOTS; ++i)
        if (ctx->ws[i].ptr) (void)hipFree(ctx->ws[i].ptr);
    for (auto &t : ctx->timed)
        for (auto e : t.ev) (void)hipEventDestroy(e);
    for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (void *p : ctx->stage)
        if (p) (void)hipHostFree(p);
    if (ctx->aux) {
        (void)hipStreamSynchronize(ctx->aux);
        (void)hipStreamDestroy(ctx->aux);
    }
    for (auto e : ctx->ev_fork)
        if (e) (void)hipEventDestroy(e);
    for (auto e : ctx->ev_tail)
        if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PHK_OK;
}

int phk_stage_ensure(phk_ctx *ctx) {
    if (ctx->stage[0] && ctx->stage[1] && ctx->stage_bytes == PHK_STAGE_BYTES) return PHK_OK;
    for (void *&p : ctx->stage) {   // (a size other than the one every user assumes cannot occur; were it to, start over)
        if (p) (void)hipHostFree(p);
        p = nullptr;
    }
    ctx->stage_bytes = 0;
    for (int i = 0; i < 2; ++i)
        if (hipHostMalloc(&ctx->stage[i], PHK_STAGE_BYTES, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            ctx->stage[i] = nullptr;
            if (i == 1) {
                (void)hipHostFree(ctx->stage[0]);
                ctx->stage[0] = nullptr;
            }
            phk_set_error("cannot allocate the pinned staging buffers (2 x %llu bytes)", (unsigned long long)PHK_STAGE_BYTES);
            return PHK_ERR_NOMEM;
        }
    ctx->stage_bytes = PHK_STAGE_BYTES;
    return PHK_OK;
}

// Device -> host for the matrices a caller asks back (counts for the features cache, normalised rows): through the
// context's two pinned staging buffers, copied on into the caller's array by host threads while the next chunk is on the
// bus.  A plain hipMemcpy into pageable memory pins the destination's pages inside the runtime, and such an array then
// costs ~0.07 s per GB to free (tools/diag/fasta_free_time.py measured the same for uploads) -- in the command line that
// was the last thing the features-cache thread did before the run could end.
int phk_copy_to_host(phk_ctx *ctx, void *dst, const void *d_src, uint64_t bytes) {
    if (bytes < (2 * PHK_STAGE_BYTES)) {
        PHK_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        PHK_HIP(hipStreamSynchronize(ctx->stream));
        return PHK_OK;
    }
    PHK_TRY(phk_stage_ensure(ctx));
    hipEvent_t landed[2] = {nullptr, nullptr};
    int rc = PHK_OK;
    auto body = [&]() -> int {
        for (int i = 0; i < 2; ++i) PHK_HIP(hipEventCreateWithFlags(&landed[i], hipEventDisableTiming));
        const uint64_t nchunks = phk_div_up(bytes, PHK_STAGE_BYTES);
        auto issue = [&](uint64_t c) -> int {
            const uint64_t o = c * PHK_STAGE_BYTES, m = bytes - o < PHK_STAGE_BYTES ? bytes - o : PHK_STAGE_BYTES;
            PHK_HIP(hipMemcpyAsync(ctx->stage[c & 1], (const char *)d_src + o, m, hipMemcpyDeviceToHost, ctx->stream));
            PHK_HIP(hipEventRecord(landed[c & 1], ctx->stream));
            return PHK_OK;
        };
        PHK_TRY(issue(0));
        for (uint64_t c = 0; c < nchunks; ++c) {
            PHK_HIP(hipEventSynchronize(landed[c & 1]));
            if (c + 1 < nchunks) PHK_TRY(issue(c + 1));   // (into the other buffer, which the host is done with)
            const uint64_t o = c * PHK_STAGE_BYTES, m = bytes - o < PHK_STAGE_BYTES ? bytes - o : PHK_STAGE_BYTES;
            const char *src = (const char *)ctx->stage[c & 1];
            char *out = (char *)dst + o;
            const uint64_t piece = 1ull << 20;
            phk_parallel_for(phk_div_up(m, piece), [&](uint64_t i) {
                const uint64_t a = i * piece, e = m - a < piece ? m - a : piece;
                memcpy(out + a, src + a, e);
            });
        }
        return PHK_OK;
    };
    rc = body();
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 2; ++i)
        if (landed[i]) (void)hipEventDestroy(landed[i]);
    return rc;
}


// Host -> device, the same way round: host threads fill a staging buffer while the previous one is on the bus.
int phk_copy_to_device(phk_ctx *ctx, void *d_dst, const void *src, uint64_t bytes) {
    if (bytes < (2 * PHK_STAGE_BYTES)) {
        PHK_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return PHK_OK;   // (stream ordered, as the callers' small copies always were)
    }
    PHK_TRY(phk_stage_ensure(ctx));
    hipEvent_t gone[2] = {nullptr, nullptr};
    auto body = [&]() -> int {
        for (int i = 0; i < 2; ++i) PHK_HIP(hipEventCreateWithFlags(&gone[i], hipEventDisableTiming));
        const uint64_t nchunks = phk_div_up(bytes, PHK_STAGE_BYTES);
        for (uint64_t c = 0; c < nchunks; ++c) {
            const uint64_t o = c * PHK_STAGE_BYTES, m = bytes - o < PHK_STAGE_BYTES ? bytes - o : PHK_STAGE_BYTES;
            if (c >= 2) PHK_HIP(hipEventSynchronize(gone[c & 1]));   // the bus is done with this staging buffer
            char *st = (char *)ctx->stage[c & 1];
            const char *in = (const char *)src + o;
            const uint64_t piece = 1ull << 20;
            phk_parallel_for(phk_div_up(m, piece), [&](uint64_t i) {
                const uint64_t a = i * piece, e = m - a < piece ? m - a : piece;
                memcpy(st + a, in + a, e);
            });
            PHK_HIP(hipMemcpyAsync((char *)d_dst + o, st, m, hipMemcpyHostToDevice, ctx->stream));
            PHK_HIP(hipEventRecord(gone[c & 1], ctx->stream));
        }
        return PHK_OK;
    };
    const int rc = body();
    (void)hipStreamSynchronize(ctx->stream);   // (the staging buffers are free again when this returns)
    for (int i = 0; i < 2; ++i)
        if (gone[i]) (void)hipEventDestroy(gone[i]);
    return rc;
}

extern "C" int phk_sync(phk_ctx *ctx) {
    PHK_ENTER(ctx, "phk_sync");
    PHK_HIP(hipStreamSynchronize(ctx->stream));
    return PHK_OK;
}

int phk_ws(phk_ctx *ctx, int slot, uint64_t bytes, void **out) {
    PhkBuf &b = ctx->ws[slot];
    if (bytes == 0) bytes = 16;
    if (b.bytes < bytes) {
        if (b.ptr) {
            // kernels already enqueued may still read the old buffer
            PHK_HIP(hipStreamSynchronize(ctx->stream));
            PHK_HIP(hipFree(b.ptr));
            b.ptr = nullptr;
            b.bytes = 0;
        }
        uint64_t want = (bytes + 255) & ~255ull;
        if (ctx->knobs.ws_fail > 0 && --ctx->knobs.ws_fail == 0) want = 1ull << 60;   // (tests: see PhkKnobs::ws_fail)
        hipError_t e = hipMalloc(&b.ptr, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            b.ptr = nullptr;
            phk_set_error("workspace slot %d: hipMalloc(%llu) failed: %s", slot, (unsigned long long)want,
                          hipGetErrorString(e));
            return PHK_ERR_NOMEM;
        }
        b.bytes = want;
        b.gen += 1;
    }
    *out = b.ptr;
    return PHK_OK;
}

extern "C" int phk_malloc(phk_ctx *ctx, uint64_t bytes, void **dptr) {
    PHK_ENTER(ctx, "phk_malloc");
    PHK_REQUIRE(dptr, "phk_malloc: NULL");
    PHK_HIP(hipMalloc(dptr, bytes ? bytes : 16));
    return PHK_OK;
}
extern "C" int phk_free(phk_ctx *ctx, void *dptr) {
    PHK_ENTER(ctx, "phk_free");
    if (dptr) {
        PHK_HIP(hipStreamSynchronize(ctx->stream));
        PHK_HIP(hipFree(dptr));
    }
    return PHK_OK;
}
extern "C" int phk_memcpy_h2d(phk_ctx *ctx, void *dst, const void *src, uint64_t bytes) {
    PHK_ENTER(ctx, "phk_memcpy_h2d");
    PHK_REQUIRE(bytes == 0 || (dst && src), "phk_memcpy_h2d: NULL");
    if (bytes) {
        PHK_TRY(phk_copy_to_device(ctx, dst, src, bytes));
        PHK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return PHK_OK;
}
extern "C" int phk_memcpy_d2h(phk_ctx *ctx, void *dst, const void *src, uint64_t bytes) {
    PHK_ENTER(ctx, "phk_memcpy_d2h");
    PHK_REQUIRE(bytes == 0 || (dst && src), "phk_memcpy_d2h: NULL");
    if (bytes) PHK_TRY(phk_copy_to_host(ctx, dst, src, bytes));
    return PHK_OK;
}

// ---- kernel timing --------------------------------------------------------------------
static int fold_events(phk_ctx *ctx, PhkTimed &t) {
    for (size_t i = 0; i + 1 < t.ev.size(); i += 2) {
        PHK_HIP(hipEventSynchronize(t.ev[i + 1]));
        float ms = 0.f;
        PHK_HIP(hipEventElapsedTime(&ms, t.ev[i], t.ev[i + 1]));
        t.ms += ms;
        t.launches += 1;
        ctx->ev_pool.push_back(t.ev[i]);
        ctx->ev_pool.push_back(t.ev[i + 1]);
    }
    t.ev.clear();
    return PHK_OK;
}

static int get_event(phk_ctx *ctx, hipEvent_t *e) {
    if (!ctx->ev_pool.empty()) {
        *e = ctx->ev_pool.back();
        ctx->ev_pool.pop_back();
        return PHK_OK;
    }
    PHK_HIP(hipEventCreate(e));
    return PHK_OK;
}

int phk_prof_begin(phk_ctx *ctx, const char *name, int *slot) {
    int idx = -1;
    for (size_t i = 0; i < ctx->timed.size(); ++i)
        if (ctx->timed[i].name == name) { idx = (int)i; break; }
    if (idx < 0) {
        ctx->timed.push_back(PhkTimed());
        ctx->timed.back().name = name;
        idx = (int)ctx->timed.size() - 1;
    }
    PhkTimed &t = ctx->timed[idx];
    if (t.ev.size() >= 4096) PHK_TRY(fold_events(ctx, t));
    hipEvent_t a, b;
    PHK_TRY(get_event(ctx, &a));
    PHK_TRY(get_event(ctx, &b));
    t.ev.push_back(a);
    t.ev.push_back(b);
    PHK_HIP(hipEventRecord(a, ctx->stream));
    *slot = idx;
    return PHK_OK;
}

int phk_prof_end(phk_ctx *ctx, int slot) {
    PhkTimed &t = ctx->timed[slot];
    PHK_HIP(hipEventRecord(t.ev.back(), ctx->stream));
    return PHK_OK;
}

extern "C" int phk_profile_enable(phk_ctx *ctx, int on) {
    PHK_REQUIRE(ctx, "phk_profile_enable: NULL ctx");
    ctx->profile = on != 0;
    return PHK_OK;
}
extern "C" int phk_profile_reset(phk_ctx *ctx) {
    PHK_ENTER(ctx, "phk_profile_reset");
    PHK_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &t : ctx->timed) {
        for (auto e : t.ev) ctx->ev_pool.push_back(e);
        t.ev.clear();
    }
    ctx->timed.clear();
    return PHK_OK;
}
extern "C" int phk_profile_count(phk_ctx *ctx, int *count) {
    PHK_REQUIRE(ctx && count, "phk_profile_count: NULL");
    *count = (int)ctx->timed.size();
    return PHK_OK;
}
extern "C" int phk_profile_get(phk_ctx *ctx, int idx, char *name, int cap, double *total_ms,
                               uint64_t *launches) {
    PHK_ENTER(ctx, "phk_profile_get");
    PHK_REQUIRE(idx >= 0 && idx < (int)ctx->timed.size(), "phk_profile_get: bad index");
    PhkTimed &t = ctx->timed[idx];
    PHK_TRY(fold_events(ctx, t));
    if (name && cap > 0) {
        strncpy(name, t.name.c_str(), cap - 1);
        name[cap - 1] = 0;
    }
    if (total_ms) *total_ms = t.ms;
    if (launches) *launches = t.launches;
    return PHK_OK;
}

// ---- host API ---------------------------------------------------------------------------
extern "C" int phk_count_ascii(phk_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n,
                               int k, const char *symbols4, int64_t *counts) {
    PHK_ENTER(ctx, "phk_count_ascii");
    PHK_REQUIRE(k >= 1, "phk_count_ascii: k must be >= 1 (got %d)", k);
    if (k > PHK_MAX_K) {
        phk_set_error("phk_count_ascii: k=%d is above PHK_MAX_K=%d", k, PHK_MAX_K);
        return PHK_ERR_UNSUPPORTED;
    }
    if (n == 0) return PHK_OK;
    PHK_REQUIRE(offsets && counts, "phk_count_ascii: NULL offsets/counts");
    const char *sym = symbols4 ? symbols4 : "ATGC";
    PHK_REQUIRE(strlen(sym) == 4, "phk_count_ascii: symbols must be exactly 4 characters");
    PHK_REQUIRE(offsets[0] == 0, "phk_count_ascii: offsets[0] must be 0");
    for (uint64_t c = 0; c < n; ++c)
        PHK_REQUIRE(offsets[c + 1] >= offsets[c], "phk_count_ascii: offsets must be non-decreasing");
    const uint64_t T = offsets[n];
    PHK_REQUIRE(T == 0 || bases, "phk_count_ascii: NULL bases");
    const uint64_t D = phk_pow4(k);
    void *d_ascii, *d_packed, *d_mask, *d_off, *d_counts, *d_wide, *d_flags;
    PHK_TRY(phk_ws(ctx, WS_ASCII, T, &d_ascii));
    PHK_TRY(phk_ws(ctx, WS_PACKED, (phk_div_up(T, 16) + 1) * 4, &d_packed));
    PHK_TRY(phk_ws(ctx, WS_MASK, (phk_div_up(T, 32) + 1) * 4, &d_mask));
    PHK_TRY(phk_ws(ctx, WS_OFFSETS, (n + 1) * 8, &d_off));
    PHK_TRY(phk_ws(ctx, WS_COUNTS, n * D * 4, &d_counts));
    PHK_TRY(phk_ws(ctx, WS_WIDE, n * D * 8, &d_wide));
    PHK_TRY(phk_ws(ctx, WS_FLAGS, 64, &d_flags));
    if (T) PHK_TRY(phk_copy_to_device(ctx, d_ascii, bases, T));
    PHK_HIP(hipMemcpyAsync(d_off, offsets, (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    PHK_TRY(phk_launch_pack(ctx, (const char *)d_ascii, T, sym, (uint32_t *)d_packed, (uint32_t *)d_mask,
                            (uint32_t *)d_flags));
    uint32_t any_invalid = 1;
    PHK_HIP(hipMemcpyAsync(&any_invalid, d_flags, 4, hipMemcpyDeviceToHost, ctx->stream));
    PHK_HIP(hipStreamSynchronize(ctx->stream));
    PHK_TRY(phk_launch_count(ctx, (const uint32_t *)d_packed, any_invalid ? (const uint32_t *)d_mask : nullptr,
                             T, (const uint64_t *)d_off, n, k, (uint32_t *)d_counts, nullptr));
    PHK_TRY(phk_launch_widen(ctx, (const uint32_t *)d_counts, n * D, (int64_t *)d_wide));
    return phk_copy_to_host(ctx, counts, d_wide, n * D * 8);
}

extern "C" int phk_normalize_i64(phk_ctx *ctx, const int64_t *counts, uint64_t n, uint64_t D, double *out) {
    PHK_ENTER(ctx, "phk_normalize_i64");
    if (n == 0 || D == 0) return PHK_OK;
    PHK_REQUIRE(counts && out, "phk_normalize_i64: NULL pointer");
    void *d_in, *d_out;
    PHK_TRY(phk_ws(ctx, WS_WIDE, n * D * 8, &d_in));
    PHK_TRY(phk_ws(ctx, WS_Q64, n * D * 8, &d_out));
    PHK_TRY(phk_copy_to_device(ctx, d_in, counts, n * D * 8));
    PHK_TRY(phk_launch_normalize_i64(ctx, (const int64_t *)d_in, n, D, (double *)d_out));
    return phk_copy_to_host(ctx, out, d_out, n * D * 8);
}

extern "C" int phk_normalize_f64(phk_ctx *ctx, const double *rows, uint64_t n, uint64_t D, double *out) {
    PHK_ENTER(ctx, "phk_normalize_f64");
    if (n == 0 || D == 0) return PHK_OK;
    PHK_REQUIRE(rows && out, "phk_normalize_f64: NULL pointer");
    void *d_in, *d_out;
    PHK_TRY(phk_ws(ctx, WS_WIDE, n * D * 8, &d_in));
    PHK_TRY(phk_ws(ctx, WS_Q64, n * D * 8, &d_out));
    PHK_TRY(phk_copy_to_device(ctx, d_in, rows, n * D * 8));
    PHK_TRY(phk_launch_normalize_f64(ctx, (const double *)d_in, n, D, (double *)d_out));
    return phk_copy_to_host(ctx, out, d_out, n * D * 8);
}

extern "C" int phk_permute_columns_i64(phk_ctx *ctx, const int64_t *rows, uint64_t n, uint64_t D, const uint32_t *perm,
                                       int64_t *out) {
    PHK_ENTER(ctx, "phk_permute_columns_i64");
    if (n == 0 || D == 0) return PHK_OK;
    PHK_REQUIRE(rows && perm && out, "phk_permute_columns_i64: NULL pointer");
    for (uint64_t j = 0; j < D; ++j) PHK_REQUIRE(perm[j] < D, "phk_permute_columns_i64: perm[%llu] out of range", (unsigned long long)j);
    void *d_in, *d_out, *d_perm;
    PHK_TRY(phk_ws(ctx, WS_WIDE, n * D * 8, &d_in));
    PHK_TRY(phk_ws(ctx, WS_Q64, n * D * 8, &d_out));
    PHK_TRY(phk_ws(ctx, WS_OFFSETS, D * 4, &d_perm));
    PHK_TRY(phk_copy_to_device(ctx, d_in, rows, n * D * 8));
    PHK_HIP(hipMemcpyAsync(d_perm, perm, D * 4, hipMemcpyHostToDevice, ctx->stream));
    PHK_TRY(phk_launch_permute_columns(ctx, (const int64_t *)d_in, n, D, (const uint32_t *)d_perm, (int64_t *)d_out));
    return phk_copy_to_host(ctx, out, d_out, n * D * 8);
}

extern "C" int phk_score(phk_ctx *ctx, const phk_model *model, const double *Q, uint64_t N, int method,
                         double *scores) {
    PHK_ENTER(ctx, "phk_score");
    PHK_REQUIRE(model, "phk_score: NULL model");
    if (N == 0) return PHK_OK;
    PHK_REQUIRE(Q && scores, "phk_score: NULL pointer");
    const uint64_t D = model->D;
    void *d_q, *d_s, *d_flags;
    PHK_TRY(phk_ws(ctx, WS_WIDE, N * D * 8, &d_q));
    PHK_TRY(phk_ws(ctx, WS_COUNTS, N * 8, &d_s));
    PHK_TRY(phk_ws(ctx, WS_FLAGS, 64, &d_flags));
    PHK_TRY(phk_copy_to_device(ctx, d_q, Q, N * D * 8));
    PHK_TRY(phk_score_rows(ctx, model, (const double *)d_q, nullptr, nullptr, N, method, (double *)d_s,
                           (uint32_t *)d_flags));
    uint32_t nan_rows = 0;
    PHK_HIP(hipMemcpyAsync(&nan_rows, d_flags, 4, hipMemcpyDeviceToHost, ctx->stream));
    PHK_HIP(hipStreamSynchronize(ctx->stream));
    if (nan_rows) {
        phk_set_error("phk_score: %u query row(s) contain NaN (zero-count contigs?)", nan_rows);
        return PHK_ERR_NAN;
    }
    PHK_HIP(hipMemcpyAsync(scores, d_s, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    PHK_HIP(hipStreamSynchronize(ctx->stream));
    return PHK_OK;
}

// ---- device API -------------------------------------------------------------------------
extern "C" int phk_pack_ascii_dev(phk_ctx *ctx, const char *d_bases, uint64_t total_bases,
                                  const char *symbols4, uint32_t *d_packed, uint32_t *d_mask,
                                  uint32_t *d_any_invalid) {
    PHK_ENTER(ctx, "phk_pack_ascii_dev");
    const char *sym = symbols4 ? symbols4 : "ATGC";
    PHK_REQUIRE(strlen(sym) == 4, "phk_pack_ascii_dev: symbols must be exactly 4 characters");
    return phk_launch_pack(ctx, d_bases, total_bases, sym, d_packed, d_mask, d_any_invalid);
}

extern "C" int phk_count_dev(phk_ctx *ctx, const uint32_t *d_packed, const uint32_t *d_mask,
                             uint64_t total_bases, const uint64_t *d_offsets, uint64_t n, int k,
                             uint32_t *d_counts, uint32_t *d_nwin) {
    PHK_ENTER(ctx, "phk_count_dev");
    return phk_launch_count(ctx, d_packed, d_mask, total_bases, d_offsets, n, k, d_counts, d_nwin);
}

extern "C" int phk_normalize_dev(phk_ctx *ctx, const uint32_t *d_counts, uint64_t n, uint64_t D,
                                 double *d_out) {
    PHK_ENTER(ctx, "phk_normalize_dev");
    PHK_REQUIRE(n == 0 || (d_counts && d_out), "phk_normalize_dev: NULL pointer");
    return phk_launch_normalize_u32(ctx, d_counts, n, D, d_out);
}

extern "C" int phk_score_dev(phk_ctx *ctx, const phk_model *model, const double *d_Q, uint64_t N,
                             int method, double *d_scores, uint32_t *d_status) {
    PHK_ENTER(ctx, "phk_score_dev");
    PHK_REQUIRE(model, "phk_score_dev: NULL model");
    PHK_REQUIRE(N == 0 || d_Q, "phk_score_dev: NULL query pointer");
    return phk_score_rows(ctx, model, d_Q, nullptr, nullptr, N, method, d_scores, d_status);
}

extern "C" int phk_score_counts_dev(phk_ctx *ctx, const phk_model *model, const uint32_t *d_counts,
                                    uint64_t N, int method, double *d_scores, uint32_t *d_status) {
    PHK_ENTER(ctx, "phk_score_counts_dev");
    PHK_REQUIRE(model, "phk_score_counts_dev: NULL model");
    PHK_REQUIRE(N == 0 || d_counts, "phk_score_counts_dev: NULL counts pointer");
    return phk_score_rows(ctx, model, nullptr, d_counts, nullptr, N, method, d_scores, d_status);
}

extern "C" int phk_count_score_dev(phk_ctx *ctx, const phk_model *model, const uint32_t *d_packed,
                                   const uint32_t *d_mask, uint64_t total_bases,
                                   const uint64_t *d_offsets, uint64_t n, int k, int method,
                                   uint32_t *d_counts, double *d_scores, uint32_t *d_status) {
    PHK_ENTER(ctx, "phk_count_score_dev");
    PHK_REQUIRE(model, "phk_count_score_dev: NULL model");
    PHK_REQUIRE(phk_pow4(k) == model->D, "phk_count_score_dev: 4^k (k=%d) != model dimension %llu", k,
                (unsigned long long)model->D);
    void *d_nwin;  // row sums straight from the count kernel (saves the scorer a pass over the counts)
    PHK_TRY(phk_ws(ctx, WS_NWIN, n * sizeof(uint32_t), &d_nwin));
    uint32_t *nwin = (uint32_t *)d_nwin;
    // (Rounds 2-4 kept a chunk pipeline here -- count of chunk i + 1 on this stream beside the scoring of chunk i on a second
    // one.  It lost at every chunk count, 5.32 ms unchunked against 5.71 / 6.39 / 7.94 with 2 / 4 / 8 chunks: every chunk paid
    // the latency-bound tail of the scoring chain again.  Removed in round 5; what does overlap is a batch's tail with the
    // next batch's sweep inside phk_score_fast.)
    // k = 5: the count kernel's flush also writes the int8 operand of the scorer's sweep (PhkPrep8, phk_common.h)
    ctx->prep8.armed = false;
    if (k == 5 && !d_mask && n > 0 && phk_model_has_fast(model) && model->d_A8 && !model->bf_stale && !ctx->knobs.force_exact &&
        !ctx->knobs.proposal[0] && !ctx->knobs.count_lanes) {
        const uint64_t D = model->D;
        void *frag, *big;
        PHK_TRY(phk_ws(ctx, WS_FRAG8, phk_div_up(n, 32) * 32 * D, &frag));
        PHK_TRY(phk_ws(ctx, WS_BIG8, (n + 1) * sizeof(uint32_t), &big));
        PHK_HIP(hipMemsetAsync(big, 0, (n + 1) * sizeof(uint32_t), ctx->stream));
        ctx->prep8.counts = d_counts; ctx->prep8.n = n; ctx->prep8.D = D;
        ctx->prep8.frag = frag; ctx->prep8.big = (uint32_t *)big;
        ctx->prep8.armed = true;
    }
    // the scorer's NaN counter and the call totals of its statistics are zeroed by the count planner's kernel where there is
    // one (no memset in front of either stage); phk_score_rows does it itself otherwise
    ctx->plan_zero[0] = d_status;
    ctx->plan_zero_words[0] = d_status ? 1u : 0u;
    const bool totals_ready = phk_model_has_fast(model) && !ctx->knobs.force_exact && ctx->ws[WS_SCTL].ptr && !ctx->score_ctl_dirty &&
                              ctx->score_ctl_gen == ctx->ws[WS_SCTL].gen;
    ctx->plan_zero[1] = totals_ready ? (uint32_t *)ctx->ws[WS_SCTL].ptr : nullptr;
    ctx->plan_zero_words[1] = totals_ready ? 32u : 0u;
    ctx->plan_zero_taken = false;
    int rc = phk_launch_count(ctx, d_packed, d_mask, total_bases, d_offsets, n, k, d_counts, nwin);
    ctx->plan_zero[0] = ctx->plan_zero[1] = nullptr;
    ctx->plan_zero_words[0] = ctx->plan_zero_words[1] = 0;
    ctx->score_totals_zeroed = ctx->plan_zero_taken;   // (consumed by phk_score_rows)
    ctx->score_totals_only_status = ctx->plan_zero_taken && !totals_ready;
    ctx->plan_zero_taken = false;
    if (rc == PHK_OK) rc = phk_score_rows(ctx, model, nullptr, d_counts, nwin, n, method, d_scores, d_status);
    ctx->prep8.armed = false;
    ctx->score_totals_zeroed = ctx->score_totals_only_status = false;
    return rc;
}

extern "C" int phk_check_counts_dev(phk_ctx *ctx, const uint32_t *d_counts, const uint32_t *d_other, uint64_t n, uint64_t D,
                                    uint64_t expected_rowsum, uint64_t *d_result) {
    PHK_ENTER(ctx, "phk_check_counts_dev");
    return phk_launch_check_counts(ctx, d_counts, d_other, n, D, expected_rowsum, d_result);
}

extern "C" int phk_score_stats(phk_ctx *ctx, uint64_t *n_fallback, uint64_t *n_exact_resolved) {
    PHK_ENTER(ctx, "phk_score_stats");
    uint32_t c[2] = {0, 0};
    if (ctx->last_score_fast && ctx->ws[WS_SCTL].ptr) {   // the first words: totals over the call's batches
        PHK_HIP(hipMemcpyAsync(c, (const uint32_t *)ctx->ws[WS_SCTL].ptr, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
        PHK_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (n_fallback) *n_fallback = c[0];
    if (n_exact_resolved) *n_exact_resolved = c[1];
    return PHK_OK;
}

extern "C" int phk_score_stats_ex(phk_ctx *ctx, uint64_t *out, int n_out) {
    PHK_ENTER(ctx, "phk_score_stats_ex");
    PHK_REQUIRE(out && n_out >= 0, "phk_score_stats_ex: NULL");
    uint32_t c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (ctx->last_score_fast && ctx->ws[WS_SCTL].ptr) {
        PHK_HIP(hipMemcpyAsync(c, (const uint32_t *)ctx->ws[WS_SCTL].ptr, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
        PHK_HIP(hipStreamSynchronize(ctx->stream));
    }
    for (int i = 0; i < n_out; ++i) out[i] = i < 9 ? c[i] : 0;
    return PHK_OK;
}

extern "C" int phk_synth_packed_dev(phk_ctx *ctx, uint64_t seed, uint64_t first_contig, uint64_t n,
                                    uint64_t L, uint32_t invalid_ppm, uint32_t *d_packed,
                                    uint32_t *d_mask, uint64_t *d_offsets) {
    PHK_ENTER(ctx, "phk_synth_packed_dev");
    return phk_launch_synth(ctx, seed, first_contig, n, L, invalid_ppm, d_packed, d_mask, d_offsets);
}

extern "C" int phk_synth_ragged_dev(phk_ctx *ctx, uint64_t seed, uint64_t first_contig, uint64_t n, const uint64_t *d_offsets,
                                    uint64_t total_bases, uint32_t gc_spread_permille, uint32_t invalid_ppm,
                                    uint32_t *d_packed, uint32_t *d_mask) {
    PHK_ENTER(ctx, "phk_synth_ragged_dev");
    if (n == 0) return PHK_OK;
    return phk_launch_synth_ragged(ctx, seed, first_contig, n, d_offsets, total_bases, gc_spread_permille, invalid_ppm,
                                   d_packed, d_mask);
}
