// phk_common.h -- context, workspace, error and kernel-timing plumbing shared by the
// translation units of libphamers_hip.so.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string>
#include <thread>
#include <functional>
#include <vector>

#include "../../include/phamers_hip.h"

#define PHK_WAVE 64

// Diagnostic code (phase timers, tuning overrides of the kernels' shape macros) compiles only in a build that says so;
// such a build reports another ABI version (phk_api.hip) and is refused by the loader.
#if (defined(I8_TIMERS) || defined(I8_NW) || defined(I8_NBUF) || defined(PHK_HI_REFINE) || \
     defined(PHK_RERANK_WAVES) || defined(PAIRS_ABL) || defined(F16H_ABL)) && !defined(PHK_DIAGNOSTIC_BUILD)
#error "kernel tuning / timer macros need -DPHK_DIAGNOSTIC_BUILD (make EXTRA_CXXFLAGS='-DPHK_DIAGNOSTIC_BUILD -D...')"
#endif

void phk_set_error(const char *fmt, ...);

#define PHK_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            phk_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                          __LINE__);                                                       \
            return e__ == hipErrorOutOfMemory ? PHK_ERR_NOMEM : PHK_ERR_HIP;               \
        }                                                                                  \
    } while (0)

#define PHK_REQUIRE(cond, ...)        \
    do {                              \
        if (!(cond)) {                \
            phk_set_error(__VA_ARGS__); \
            return PHK_ERR_ARG;       \
        }                             \
    } while (0)

// first statement of every extern "C" entry that takes a context: the calling thread may have another device current
#define PHK_ENTER(ctx_, fname)                                   \
    do {                                                         \
        PHK_REQUIRE((ctx_) != nullptr, fname ": NULL ctx");      \
        PHK_HIP(hipSetDevice((ctx_)->device));                   \
    } while (0)

#define PHK_TRY(call)               \
    do {                            \
        int rc__ = (call);          \
        if (rc__ != PHK_OK) return rc__; \
    } while (0)

// workspace slots (device scratch owned by the context, grown on demand, reused)
enum PhkSlot {
    WS_ASCII = 0,   // host API: uploaded ASCII bases
    WS_PACKED,      // host API: packed stream
    WS_MASK,        // host API: validity mask
    WS_OFFSETS,     // host API: offsets
    WS_COUNTS,      // host API: uint32 counts
    WS_WIDE,        // host API: int64 / float64 staging
    WS_FLAGS,       // small flags / status words
    WS_Q64,         // scoring: float64 query rows
    WS_SCORES,      // scoring: float64 scores
    WS_DIST,        // scoring (exact path): distance tiles
    WS_QF32,        // scoring (MFMA path): fragment-ordered fp32 queries
    WS_CAND,        // scoring (MFMA path): candidate lists
    WS_NWIN,        // row sums
    WS_LONG,        // counting: contigs handed from the lane-pair kernel to the wave-per-contig kernel
    WS_OUT,         // batch API: scores on their way to the host
    WS_SUB,         // scoring at general D: dense count rows (+ row sums) of a second pass's sub-batch
    WS_QUEUE,       // scoring at general D: the two hand-over queues of the first pass
    WS_FRAG8,       // count -> score at k = 5: the int8 query fragments of the whole call, written by the count kernel's flush
    WS_BIG8,        // ... and their per-row flag words (+ the call's mode word)
    WS_SCTL,        // scoring (MFMA path): the call's statistics totals, the two sets of counters / query lists / striped words
    WS_CTL,         // small control words the kernels keep zeroed themselves (no memset per call): the count planner's two
                    // alternating blocks (PhkCountCtl), see phk_launch_count
    WS_SLOTS
};

struct PhkBuf {
    void *ptr = nullptr;
    uint64_t bytes = 0;
    uint64_t gen = 0;    // counts (re)allocations: a user that keeps state in the buffer sees when it has to start over
};

struct PhkTimed {
    std::string name;
    std::vector<hipEvent_t> ev;  // start/stop pairs not yet folded
    double ms = 0.0;
    uint64_t launches = 0;
};

// Tuning / diagnostic knobs.  Read ONCE from the environment when the context is created (PHK_<NAME>), changed
// afterwards only through phk_set_option(): no launch path calls getenv.
struct PhkKnobs {
    char count_lanes = 0;      // '0' wave-per-contig kernel only, '1' slot kernel by the batch statistics, '2' slot kernel whatever the
                               // batch looks like, 'p' / 'P' (forced: 'q' / 'Q') two-windows-per-add kernel with 512 / 1024 threads (k = 4, no mask)
    bool force_exact = false;  // every model through the float64 path
    char proposal[8] = "";     // "f32" fp32 MFMA, "f16" split-query f16, "cx2" count-exact with 2 MFMAs per k-step
    char cx_cfg[8] = "";       // "<tiles per wave><waves per workgroup>": 14, 24, 28
    char rerank = 0;           // 'w' wave per query, 'g' 16 lanes per query for all
    bool count_sort = true;    // length-bucketed contig order for the slot count kernel on ragged batches
    uint64_t score_batch = 0;  // queries per scoring batch of the MFMA path (0 = default 2^20; tests shrink it)
    char i8_insert = 0;        // two-part int8 sweep, how a tile's values meet the lists: '0' test per eight and per value, '1' test per eight,
                               // '2' no test (0 = by the number of columns, phk_launch_proposal_i8_general)
    bool gen_seq = false;      // int8 sweep: the column groups as successive launches (each group's records L2-resident) instead of a 2-D launch
    int gen_groups = 0;        // column groups of the general-D sweep's 2-D launch (0 = default: PHK_GEN_GROUPS at D >= 2048, else 1)
    int ws_fail = 0;           // fault injection for the tests: the ws_fail-th workspace allocation from now on asks hipMalloc for
                               // 2^60 bytes (a genuine out-of-memory failure on the genuine error path), then the knob is spent
    bool tail_aside = true;    // multi-batch scoring at k = 4: a batch's hand-over kernels on the second stream beside the next
                               // batch's sweep (option "tail_aside" = 0: everything on one stream, for A/B runs)
};

// The int8 operand of the general-D sweep prepared by the count kernel (phk_count_score_dev at k = 5: the flush of
// phk_count_direct_kernel writes each row's centred int8 fragment pieces beside the counts, which saves the scorer a pass
// over the 4 KB count rows).  big[n] is the call's mode word: 0 = every row prepared, 1 = rows flagged PHK_PREP8_MISSING
// are not (contigs handed to the wave-per-contig kernel), 2 = nothing prepared (a ragged batch took the sorted slot
// kernel); phk_split_queries_i8_kernel completes the operand accordingly.
#define PHK_PREP8_MISSING 0x40000000u
#define PHK_I8_L1_MAX 65000u   // |c - c0|_1 of a row the sweep epilogue's 32-bit fold is exact for: (256 * 127 + 128) * 65000 < 2^31
struct PhkPrep8 {
    bool armed = false;
    const uint32_t *counts = nullptr;   // the count matrix the fragments belong to
    uint64_t n = 0, D = 0;
    void *frag = nullptr;               // [ceil(n / 32)][D / 32][64 lanes] x 16 bytes
    uint32_t *big = nullptr;            // [n + 1]
};

struct phk_ctx {
    int device = 0;
    PhkPrep8 prep8;
    PhkKnobs knobs;
    bool slots_lds0 = true;        // the slot count kernel's dynamic LDS starts at address 0 (checked at creation)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // second stream of multi-batch scoring calls (phk_score_fast): batch b's hand-over kernels -- second chance, merge, brute
    // force: a few % of the chip -- run here beside batch b + 1's sweep; forked from / joined to `stream` with events inside
    // the call, so the caller still sees one stream-ordered operation
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork[2] = {nullptr, nullptr}, ev_tail[2] = {nullptr, nullptr};
    // Control words of the count planner (phk_launch_count): two blocks used alternately -- the planning kernel of a call
    // zeroes the block of the NEXT call, so no launch is preceded by a memset.  ctl_gen = the allocation the blocks were
    // zeroed for; ctl_epoch = calls so far (its parity picks the block); ctl_dirty = a call failed half way.
    uint64_t ctl_gen = 0, ctl_epoch = 0;
    bool ctl_dirty = true;
    // the scorer's control words (phk_score_fast, WS_SCTL): zeroed by the kernels that read them last; see there
    uint64_t score_ctl_gen = 0;
    bool score_ctl_dirty = true;
    bool score_totals_zeroed = false;   // this call's NaN counter (and totals) were zeroed by the count planner's kernel ...
    bool score_totals_only_status = false;   // ... the NaN counter only (the totals' workspace did not exist yet)
    // words another entry point wants zeroed by the planning kernel (phk_count_score_dev: the scorer's NaN counter and call
    // totals), consumed by the next phk_launch_count that launches a planning kernel; `taken` says it did
    uint32_t *plan_zero[2] = {nullptr, nullptr};
    uint32_t plan_zero_words[2] = {0, 0};
    bool plan_zero_taken = false;
    int num_cus = 256;
    PhkBuf ws[WS_SLOTS];
    bool profile = false;
    bool last_score_fast = false;  // the last scoring call took the MFMA path: WS_SCTL's first words hold its totals (phk_score_stats)
    std::vector<PhkTimed> timed;
    std::vector<hipEvent_t> ev_pool;
    // pinned staging buffers of the sequence upload (phk_batch_from_ascii), allocated at the first multi-chunk upload
    void *stage[2] = {nullptr, nullptr};
    uint64_t stage_bytes = 0;
};

int phk_ws(phk_ctx *ctx, int slot, uint64_t bytes, void **out);

// Large transfers between caller arrays and the device go through the context's two pinned staging buffers (64 MB each),
// host threads copying on one side while the bus works on the other: an array handed to hipMemcpy as it is has its pages
// pinned inside the runtime and costs 0.05 - 0.1 s per GB to free afterwards (tools/diag/fasta_free_time.py).  Below 128 MB
// a plain stream-ordered copy.  phk_copy_to_host returns with the data in place; phk_copy_to_device is stream ordered.
struct phk_batch;
// (batch.hip) the sequence bytes of a batch -> device, packed and counted; see there
int phk_batch_build(phk_ctx *ctx, const char *bases, const std::function<void(uint64_t, uint64_t, char *)> *fill,
                    const uint64_t *offsets, uint64_t n, int k, const char *symbols4, phk_batch **out);
// (batch.hip) the same from the RAW bytes of a FASTA file (title lines, line ends and all) + one layout entry per record:
// the bytes go up as they are and the device drops what is not sequence (phk_deline_pack_kernel, count.hip).
// phk_raw_to_device puts the file's bytes into the context's WS_ASCII workspace (staged copies; it may run on a thread of its
// own beside the host's index scan -- nothing else uses the context meanwhile); phk_batch_build_raw then takes the record
// layout and the side buffer of the irregular records (rbegin[r] indexes the raw bytes, or `side` where rlw[r] = 2^32 - 1).
int phk_raw_to_device(phk_ctx *ctx, const char *raw, uint64_t raw_bytes, const uint8_t **d_raw);
int phk_batch_build_raw(phk_ctx *ctx, const uint8_t *d_raw, const char *side, uint64_t side_bytes,
                        const uint64_t *rbegin, const uint32_t *rlw, const uint32_t *rtl, const uint64_t *offsets, uint64_t n,
                        int k, const char *symbols4, phk_batch **out);
#define PHK_STAGE_BYTES (64ull << 20)
// the context's two pinned staging buffers of PHK_STAGE_BYTES each, allocated on first use by whichever transfer needs them;
// a partial failure frees what it got (the next call starts from nothing) and returns PHK_ERR_NOMEM
int phk_stage_ensure(phk_ctx *ctx);
int phk_copy_to_host(phk_ctx *ctx, void *dst, const void *d_src, uint64_t bytes);
int phk_copy_to_device(phk_ctx *ctx, void *d_dst, const void *src, uint64_t bytes);

// kernel timing: PHK_LAUNCH(ctx, "name", kernel<<<...>>>(...)) brackets the launch with HIP
// events on the context's stream when profiling is on.
int phk_prof_begin(phk_ctx *ctx, const char *name, int *slot);
int phk_prof_end(phk_ctx *ctx, int slot);

#define PHK_LAUNCH(ctx, name, ...)                         \
    do {                                                   \
        int ps__ = -1;                                     \
        if ((ctx)->profile) PHK_TRY(phk_prof_begin((ctx), (name), &ps__)); \
        __VA_ARGS__;                                       \
        PHK_HIP(hipGetLastError());                        \
        if ((ctx)->profile) PHK_TRY(phk_prof_end((ctx), ps__)); \
    } while (0)

// A grid of 2^32 work-items or more is not launched whole: the excess is dropped without an error (round 5: a 30M-contig
// synthetic batch came out generated for its first 2.5M contigs only).  Launchers whose grid grows with the input's bases or
// rows go through this loop: slices of at most PHK_SLICE_BLOCKS blocks, the kernel is told the slice's first block.
#define PHK_SLICE_BLOCKS (1ull << 22)   // x 1024 threads at most = 2^32 / 1
#define PHK_LAUNCH_SLICED(ctx, name, total_blocks, block0, nblk, ...)                                        \
    for (uint64_t block0 = 0, tb__ = (total_blocks); block0 < tb__; block0 += PHK_SLICE_BLOCKS) {           \
        const unsigned nblk = (unsigned)(tb__ - block0 < PHK_SLICE_BLOCKS ? tb__ - block0 : PHK_SLICE_BLOCKS); \
        PHK_LAUNCH(ctx, name, __VA_ARGS__);                                                                  \
    }

// host-side parallel loop for the model builders (independent iterations, disjoint outputs)
template <typename F>
static inline void phk_parallel_for(uint64_t n, F fn) {
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    if (n < 2 * (uint64_t)nt) nt = 1;
    if (nt == 1) {
        for (uint64_t i = 0; i < n; ++i) fn(i);
        return;
    }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([=]() {
            for (uint64_t i = n * t / nt; i < n * (t + 1) / nt; ++i) fn(i);
        });
    for (auto &th : pool) th.join();
}

// x / T for the elements of ONE count row: T and y = 1.0 / T (one IEEE division) are shared by the row, each element costs
// two Newton steps on the quotient -- 5 multiply-adds instead of the ~35 instructions of a float64 division (the general-D
// decision kernel spent a third of its vector instructions dividing).  The result IS the correctly rounded quotient, bit
// for bit what `x / T` returns: with y the correctly rounded reciprocal and q within one ulp of x / T,
// RN(q + (x - T q) y) = RN(x / T) (Markstein's theorem; fma makes the residual exact), and the first step brings
// q = RN(x y) within one ulp.  0 / 0 stays NaN (y = inf).  Checked on the device by the normalise kernel's bit-exact
// tests against NumPy's division (tests/test_gpu_count.py), and in exact rational arithmetic by tools/diag/div_exact_check.py.
__device__ __forceinline__ double phk_div_row(double x, double T, double y) {
    const double q0 = x * y;
    const double q1 = __builtin_fma(__builtin_fma(-q0, T, x), y, q0);
    return __builtin_fma(__builtin_fma(-q1, T, x), y, q1);
}

// Count rows enter the count-exact MFMA kernels CENTRED by an integer: c_i - c0 with c0 = the integer nearest to T / D
// (T = row sum); every kernel derives c0 from T with this one function (see score_lists.h).
__host__ __device__ __forceinline__ uint32_t phk_row_center(uint32_t T, uint32_t D) { return (T + D / 2) / D; }

static inline uint64_t phk_pow4(int k) { return 1ull << (2 * k); }
static inline uint64_t phk_div_up(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// ---- internal entry points implemented across translation units ----
// per-device kernel attributes (dynamic LDS above 64 KiB), set for the context's device at phk_create
int phk_count_init_device(phk_ctx *ctx);       // count.hip
int phk_score_f16_init_device(phk_ctx *ctx);   // score_f16.hip
int phk_score_mfma_init_device(phk_ctx *ctx);  // score_mfma.hip
// count.hip
int phk_launch_pack(phk_ctx *ctx, const char *d_bases, uint64_t T, const char *symbols4,
                    uint32_t *d_packed, uint32_t *d_mask, uint32_t *d_any_invalid);
int phk_launch_deline_pack(phk_ctx *ctx, const uint8_t *d_raw, const uint8_t *d_side, const uint64_t *d_offsets, uint64_t n, const uint64_t *d_rbegin,
                           const uint32_t *d_rlw, const uint32_t *d_rtl, uint64_t T, const char *symbols4, uint32_t *d_packed,
                           uint32_t *d_mask, uint32_t *d_any_invalid);
int phk_launch_count(phk_ctx *ctx, const uint32_t *d_packed, const uint32_t *d_mask, uint64_t T,
                     const uint64_t *d_offsets, uint64_t n, int k, uint32_t *d_counts,
                     uint32_t *d_nwin, uint64_t mean_bases = 0);
int phk_launch_widen(phk_ctx *ctx, const uint32_t *d_in, uint64_t count, int64_t *d_out);
int phk_launch_normalize_u32(phk_ctx *ctx, const uint32_t *d_counts, uint64_t n, uint64_t D,
                             double *d_out);
int phk_launch_normalize_i64(phk_ctx *ctx, const int64_t *d_counts, uint64_t n, uint64_t D,
                             double *d_out);
int phk_launch_normalize_f64(phk_ctx *ctx, const double *d_rows, uint64_t n, uint64_t D,
                             double *d_out);
int phk_launch_permute_columns(phk_ctx *ctx, const int64_t *d_in, uint64_t n, uint64_t D, const uint32_t *d_perm,
                               int64_t *d_out);
int phk_launch_check_counts(phk_ctx *ctx, const uint32_t *d_counts, const uint32_t *d_other, uint64_t n, uint64_t D,
                            uint64_t expected_rowsum, uint64_t *d_result);
// synth.hip
int phk_launch_synth(phk_ctx *ctx, uint64_t seed, uint64_t first, uint64_t n, uint64_t L,
                     uint32_t invalid_ppm, uint32_t *d_packed, uint32_t *d_mask,
                     uint64_t *d_offsets);
int phk_launch_synth_ragged(phk_ctx *ctx, uint64_t seed, uint64_t first, uint64_t n, const uint64_t *d_offsets,
                            uint64_t total_bases, uint32_t gc_spread_permille, uint32_t invalid_ppm, uint32_t *d_packed,
                            uint32_t *d_mask);
// score.hip
struct phk_model;
int phk_score_rows(phk_ctx *ctx, const phk_model *m, const double *d_Q, const uint32_t *d_counts,
                   const uint32_t *d_rowsum, uint64_t N, int method, double *d_scores, uint32_t *d_status);
