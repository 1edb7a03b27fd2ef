// rcv_internal.h -- shared host-side plumbing of librustcv_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/rustcv_hip.h"

#define RCV_MAX_STAGE 6

// Layout of rcv_ctx::kconst (one 64-KiB device allocation per context: weight tables and the dump lines masked-off lanes store
// to).  Every user takes its offset from here; the asserts keep the regions apart.
constexpr size_t RCV_KC_F7_TAB = 0, RCV_KC_F7_TAB_BYTES = 8192;          // strip kernel: 2 x 4 tables x 64 lanes x 16 B
constexpr size_t RCV_KC_SOBEL_DUMP = 8192, RCV_KC_SOBEL_DUMP_BYTES = 4096;   // 256 lanes x 16 B
constexpr size_t RCV_KC_F7_DUMP = 16384, RCV_KC_F7_DUMP_BYTES = 4096;    // strip kernel: 3 KiB
constexpr size_t RCV_KC_FR_TICKETS = 20480, RCV_KC_FR_TICKETS_BYTES = 8192;   // chained row kernel: four sets of 16 ticket counters, one 128-byte line each
constexpr size_t RCV_KC_FR_TAB = 32768, RCV_KC_FR_TAB_BYTES = 16384;     // row kernel: up to 2 x 2 x 4 tables x 1 KiB (two weight tables)
constexpr size_t RCV_KC_BENCH = 49152, RCV_KC_BENCH_BYTES = 4096;        // rcv__membench read-only dump (256 threads x 16 B)
constexpr size_t RCV_KC_PROBE = 57344, RCV_KC_PROBE_BYTES = 128;         // rcv__clock_probe: 8 x 2 counters
constexpr size_t RCV_KC_FR_DUMP = 61440, RCV_KC_FR_DUMP_BYTES = 1024;    // row kernel: 64 lanes x 16 B
constexpr size_t RCV_KC_FR_TICKETS2 = 65536;   // the ticket sets of the context's second launch lane (the half stream: see rcv_ctx::half), RCV_KC_FR_TICKETS_BYTES
constexpr size_t RCV_KC_BYTES = 65536 + 8192;
static_assert(RCV_KC_F7_TAB + RCV_KC_F7_TAB_BYTES <= RCV_KC_SOBEL_DUMP && RCV_KC_SOBEL_DUMP + RCV_KC_SOBEL_DUMP_BYTES <= RCV_KC_F7_DUMP &&
              RCV_KC_F7_DUMP + RCV_KC_F7_DUMP_BYTES <= RCV_KC_FR_TICKETS && RCV_KC_FR_TICKETS + RCV_KC_FR_TICKETS_BYTES <= RCV_KC_FR_TAB && RCV_KC_FR_TAB + RCV_KC_FR_TAB_BYTES <= RCV_KC_BENCH &&
              RCV_KC_BENCH + RCV_KC_BENCH_BYTES <= RCV_KC_PROBE && RCV_KC_PROBE + RCV_KC_PROBE_BYTES <= RCV_KC_FR_DUMP &&
              RCV_KC_FR_DUMP + RCV_KC_FR_DUMP_BYTES <= RCV_KC_BYTES, "kconst regions overlap");

struct rcv_ctx {
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    // grow-only staging mirrors for RCV_HOST mats, one per argument slot of a call
    uint8_t* stage_buf[RCV_MAX_STAGE];
    size_t stage_cap[RCV_MAX_STAGE];
    // workspace for kernel-internal temporaries (reserve once per call, then carve)
    uint8_t* ws;
    size_t ws_cap, ws_off;
    uint8_t* tmp2;       // second grow-only temporary (an intermediate image of a two-stage fallback whose second stage uses ws)
    size_t tmp2_cap;
    // small device scratch for per-call constants (filter taps, weight tables)
    uint8_t* kconst;     // 64 KiB
    int cu_count;
    // cached banded-weight table of the MFMA filter (rcv_filter7_mfma.hip), lives at RCV_KC_F7_TAB
    bool f7_valid;
    int f7_ksize;
    int f7_mode;         // 0 one table, 1 K = 4Q + R, 2 K = K1 + 2*T2 (which tables the cache holds)
    int16_t f7_k[49];
    // cached launch plan of the strip kernel (segment height, latency variant) for the last geometry
    int f7_plan_rows, f7_plan_nstrips, f7_plan_n, f7_plan_seg_rows;
    bool f7_plan_lat_ok, f7_plan_lat;
    // cached banded-weight tables of the row-streaming MFMA filter (rcv_filter_rows_mfma.hip): four entries, least recently used
    // replaced, each with its own 16-KiB device region (fr_tabs) and a persistent host copy, so that a caller alternating between
    // a few kernels (GaussianBlur, then filter2D, ...) neither rebuilds a table nor synchronises the stream (round 3)
    struct FrTab {
        bool valid, split2;
        int ksize, dmask;
        int16_t k[49];
        unsigned long long stamp;
        hipEvent_t uploaded;      // recorded behind the entry's last upload: its host copy may be rewritten once this has passed
        int8_t host[16384];
    } fr_tab[4];
    // The chained row kernel's launch state, per LANE: lane 0 = the context's stream, lane 1 = the half stream (below).  Ticket counters of
    // lane l live at kconst + (l ? RCV_KC_FR_TICKETS2 : RCV_KC_FR_TICKETS); launch i of a lane draws from set i % 4 and zeroes set (i + 2) % 4.
    // Completion check (rcv_filter_rows_mfma.hip: fr_check_set): launch i + 1 of a lane checks launch i, rcv_chain_flush() the last one before a
    // host-side wait; a launch that left items undone raises *fr_fault (pinned host memory), rcv_chain_poll() turns that into RCV_ERR_DEVICE
    // once, re-zeroes the counters and sends the context's later launches to the one-band-per-wave kernel
    struct ChainLane {
        bool tickets_ready;       // the lane's counters have been zeroed
        bool unchecked;           // its last chained launch has not been checked yet
        unsigned seq;             // its chained launches so far
        unsigned prev[5];         // plan of its last chained launch: cbands, interior strips, edge strips, tapered bands (half, quarter)
    } fr_lane[2];
    unsigned* fr_fault;           // hipHostMalloc'ed word, 0 = fine; written by the device on a fault only
    bool fr_chain_off;            // a fault was seen: no chained launches on this context any more
    unsigned fr_uploads;          // weight-table uploads so far (an upload on the stream must be seen by the half stream before it launches)
    // Two halves of one call on two streams (round 6, profiles/r06_split_halves.txt: -1.4 ... -1.8 %): a chained filter2D call of 16+ frames
    // runs its first half on `stream` and its second half on `half`; the halves of consecutive calls hide each other's tail and launch gap.
    // Nothing is joined per call.  EVERY other entry point joins first (rcv_bind: `stream` waits for `half`), so the contract "everything a
    // context enqueues is ordered on its stream" holds for all that a caller can observe; data hazards between the halves of consecutive
    // split calls are checked on address ranges (R / W hulls of what each stream has pending since the two last met).  Off when the stream
    // has been handed out (rcv_ctx_stream), when another context of the device has work in flight, and with RCV_FR_SPLIT=0.
    hipStream_t half;
    hipEvent_t ev_half, ev_main;
    bool half_busy;               // `half` holds work that `stream` has not waited for
    bool main_unknown;            // `stream` holds work of other entry points since `half` last waited for it
    bool stream_exported;         // rcv_ctx_stream was called: the caller may enqueue behind our back -- no split any more
    int split_n = 0;              // inside rcv_split_run: the frames of the WHOLE call (a half's launch plans its segments for the pair of launches)
    struct Hull { uintptr_t lo, hi; };
    struct HullSet {              // a few byte ranges; more than it holds are merged into their hull (conservative)
        Hull h[6];
        int n;
    } main_r, main_w, half_r, half_w;   // read / written by the split launches pending on each stream since the two last met
    uint8_t* fr_tabs;             // 4 x 16 KiB of device memory (allocated on first use)
    unsigned long long fr_clock;
    // last plan of the LDS-staged warpAffine kernel (rcv_geom.hip: warp_lds_plan), keyed by the matrix
    bool wl_valid = false, wl_ok = false;
    float wl_M[6] = {0, 0, 0, 0, 0, 0};
    int wl_pitch = 0, wl_prow = 0, wl_cpr = 0;
    // last verdict of the staged fused warp -> down-scale kernel's host-side plan check (rcv_warp_resize.hip: wrs_fits), keyed by the
    // matrix and the geometry: the check evaluates ~9 000 sample coordinates, a stream of launches with one map pays it once
    // (round 6: four entries, replaced in turn -- a caller that alternates between a few maps, or whose matrix changes slowly and returns,
    //  does not pay the check on every launch)
    struct WrsEntry {
        bool valid, ok;
        float M[6];
        int geom[5];   // source rows / cols, destination rows / cols, S
    } wrs[4];
    int wrs_next;
    hipStream_t side;            // (created on first use) third stream: the measurement library's clock probe runs beside the launches
    // grow-only pinned staging for small per-call host tables that outlive the call (rcv_text_blend.hip)
    uint8_t* pin;
    size_t pin_cap;
    hipEvent_t pin_ev;           // recorded after the H2D that reads `pin`
    int children;                // live staging rings that use this context's device and stream
    bool zombie;                 // rcv_ctx_destroy was called while children were alive: freed when the last one goes
    int harris_wpc[2];           // cached occupancy (waves per CU) of the fused Harris kernel, mask-only / with response
};

// Environment knobs: DISPATCH OVERRIDES for the tests (send the same shapes through both kernels of a pair), fourteen in all (one of them a fault injection); nothing
// needs them in production and no tuning parameter is among them (those are arguments of the measurement entries in
// librustcv_hip_bench.so).  Read ONCE per process -- a launch-bound call (a single 1080p frame: 6 us) must not pay for getenv -- and
// again on rcv__debug_reload_knobs() (tests, after setenv).  DESIGN.md 5 names them; DESIGN_HISTORY.md 5 lists each with the test that uses it.
struct RcvKnobs {
    int f7_rows;          // RCV_F7_ROWS       row-streaming MFMA kernel: 1 every eligible shape, 0 never, -1 (unset) by size
    int f7_no_lat;        // RCV_F7_NO_LAT     small launches take the pipelined strip kernel instead of its latency variant
    int f7_no_gray;       // RCV_F7_NO_GRAY    one-channel images take the dot4 streaming kernel
    int f7_dual_full;     // RCV_F7_DUAL_FULL  large-weight kernels use K = 4Q + R even where the centre split applies
    int fr_chain;         // RCV_FR_CHAIN      chained-band kernel: 0 never, 1 every eligible launch, -1 (unset) launches that fill the GPU
    int fr_chain_rows;    // RCV_FR_CHAIN_ROWS rows per chained band (0 = 32): band seams at other rows
    int fr_split;         // RCV_FR_SPLIT      0: a chained filter call never runs as two halves on two streams (tests: both forms, same bytes)
    int fr_chain_drop_xcd;  // RCV_FR_CHAIN_DROP_XCD  FAULT INJECTION (test of the completion check): the chained kernel's waves on this XCD leave at once
    int gauss_rows;       // RCV_GAUSS_ROWS    register-window integer Gaussian: 1 every eligible shape, 0 never, -1 (unset) small launches
    int gr_seg;           // RCV_GR_SEG        its rows per segment (0 = per-SIMD plan): segment seams at every height
    int harris_general;   // RCV_HARRIS_GENERAL   1: blockSize 2 on the general-block kernel too
    int warp_lds;         // RCV_WARP_LDS      0: never the LDS-staged warpAffine kernel (the gather kernel on the same maps)
    int warp_gray4;       // RCV_WARP_GRAY4    0: one-channel warpAffine never on the four-frames-per-pass kernel
    int warp_fpg;         // RCV_WARP_FPG      frames per workgroup of the warp kernels (an incomplete last group); + 256 * (s + 1): tile strips of s columns
};
const RcvKnobs& rcv_knobs();

// Every kernel launch goes through RCV_LAUNCH, which logs the kernel's name in a per-thread list: tests assert WHICH kernel an
// entry point dispatched (rcv__debug_kernels / rcv__debug_kernels_reset) instead of guessing it from timings.
void rcv_note_kernel(const char* name);
#define RCV_LAUNCH(kernelName, grid_, block_, lds_, ...)                                                   \
    do {                                                                                                   \
        rcv_note_kernel(#kernelName);                                                                      \
        hipLaunchKernelGGL(kernelName, grid_, block_, (lds_), __VA_ARGS__);                                  \
    } while (0)

// Device copy of a small per-call constant table, valid for the kernel about to be enqueued: the shared 64-KiB kconst area at
// `offset`, uploaded stream-ordered.
int rcv_const_table(rcv_ctx* ctx, const void* host, size_t bytes, size_t offset, const uint8_t** dev);

// Kernel-facing description of a (batch of) strided image(s).
struct View {
    uint8_t* p;      // frame 0
    size_t step;     // bytes per row
    size_t fstride;  // bytes between frames
    size_t cap;      // capacity of one frame
    int rows, cols, ch, esz, n;
};

#define RCV_HIP(call)                                   \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess) {                         \
            (void)hipGetLastError();                    \
            return e_ == hipErrorOutOfMemory ? RCV_ERR_OOM : RCV_ERR_DEVICE; \
        }                                               \
    } while (0)

#define RCV_TRY(expr)                \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ < 0) return rc_;     \
    } while (0)

static inline int rcv_elem_size(int depth) { return depth == RCV_8U ? 1 : (depth == RCV_16S ? 2 : (depth == RCV_32F ? 4 : 0)); }

// ---- one call as two halves on the context's two streams (rcv_ctx::half; implemented in rcv_ctx.hip) ---------------------------------------
struct RcvRanges {            // what ONE half of a call reads and writes
    rcv_ctx::Hull r[3], w[3];
    int nr = 0, nw = 0;
    void read(const View& v, int f0, int f1) { r[nr++] = span(v, f0, f1); }
    void write(const View& v, int f0, int f1) { w[nw++] = span(v, f0, f1); }
    static rcv_ctx::Hull span(const View& v, int f0, int f1)   // frames [f0, f1)
    {
        const uintptr_t base = (uintptr_t)v.p;
        return rcv_ctx::Hull{base + (uintptr_t)f0 * v.fstride, base + (uintptr_t)(f1 - 1) * v.fstride + (uintptr_t)v.rows * v.step};
    }
};
static inline View rcv_view_frames(const View& v, int f0, int f1)
{
    View o = v;
    o.p = v.p + (size_t)f0 * v.fstride;
    o.n = f1 - f0;
    return o;
}
int rcv_split_begin(rcv_ctx* ctx, int n, const RcvRanges& a, const RcvRanges& b);   // RCV_OK: go (device bound, `stream` clear of what half A touches); RCV_ERR_UNSUPPORTED: not split
int rcv_split_half(rcv_ctx* ctx, const RcvRanges& a, const RcvRanges& b, unsigned uploads_before);   // half A is enqueued: `half` made to wait where it must
void rcv_split_done(rcv_ctx* ctx, const RcvRanges& b);
// launch(i): enqueue half i (0: frames [0, n / 2), 1: the rest) on ctx->stream -- for half 1 the context's two streams are swapped around the call, so
// every kernel, table upload and event of the ordinary code path lands on the half stream.  launch(0) == RCV_ERR_UNSUPPORTED: nothing was enqueued, the
// call is not split (the caller goes through rcv_bind and its ordinary path).
template <class F>
int rcv_split_run(rcv_ctx* ctx, int n, const RcvRanges& a, const RcvRanges& b, F&& launch)
{
    RCV_TRY(rcv_split_begin(ctx, n, a, b));
    const unsigned up0 = ctx->fr_uploads;
    ctx->split_n = n;
    int rc = launch(0);
    if (rc < 0) {
        ctx->split_n = 0;
        return rc;
    }
    rc = rcv_split_half(ctx, a, b, up0);
    if (rc < 0) {
        ctx->split_n = 0;
        return rc;
    }
    hipStream_t t = ctx->stream;
    ctx->stream = ctx->half;
    ctx->half = t;
    rc = launch(1);
    t = ctx->stream;
    ctx->stream = ctx->half;
    ctx->half = t;
    ctx->split_n = 0;
    if (rc == RCV_ERR_UNSUPPORTED) return RCV_ERR_UNSUPPORTED;   // (half 1 not taken where half 0 was: the caller's ordinary path redoes the whole call on `stream`)
    RCV_TRY(rc);
    rcv_split_done(ctx, b);
    return RCV_OK;
}

// ---- helpers implemented in rcv_ctx.hip -------------------------------------------------
void rcv_ctx_child_released(rcv_ctx* ctx);                   // a graph / ring of this context was destroyed
int rcv_bind(rcv_ctx* ctx);                                   // hipSetDevice(ctx->device); `stream` joins `half`
int rcv_bind_raw(rcv_ctx* ctx);                               // hipSetDevice only (the split launch path, which keeps the two streams apart)
int rcv_join_half(rcv_ctx* ctx);                              // `stream` waits for what `half` holds
bool rcv_other_context_busy(const rcv_ctx* me);               // another context of the device has work in flight right now (hipStreamQuery)
int rcv_launch_check(rcv_ctx* ctx);                           // hipGetLastError -> code
int rcv_wait(rcv_ctx* ctx);                                   // every host-side wait on the context's stream: flush + synchronize + poll (below)
int rcv_chain_flush(rcv_ctx* ctx);                            // enqueue the completion check of the last chained launch, if it is still unchecked
int rcv_chain_poll(rcv_ctx* ctx);                             // RCV_ERR_DEVICE (once) if a chained launch left items undone
int rcv_ws_reserve(rcv_ctx* ctx, size_t total);               // (re)size the workspace, reset the carve pointer
int rcv_ws_alloc(rcv_ctx* ctx, size_t bytes, uint8_t** out);  // 256-B aligned carve
int rcv_side_reserve(rcv_ctx* ctx, size_t bytes, uint8_t** out);   // the side buffer, grown if needed
int rcv_upload_const(rcv_ctx* ctx, const void* host, size_t bytes, size_t offset); // into ctx->kconst (async, stream ordered)

// Validate a strided mat (step/cap vs rows/cols) and turn it into a View.
int rcv_view_strided(const rcv_mat* m, int want_depth, View* v);
// Batch -> View (device only).
int rcv_view_batch(const rcv_batch* b, int want_depth, View* v);

// Host staging: the single-Mat entry points accept RCV_HOST mats.  A Stage keeps the
// device mirror of each argument; outputs are copied back by stage_finish().
struct StagedMat {
    const rcv_mat* host;  // original (may be device-resident: then dev == *host)
    rcv_mat dev;          // device-resident description
    bool copy_back;
};
struct Stage {
    rcv_ctx* ctx;
    StagedMat m[RCV_MAX_STAGE];
    int count;
    bool any_host;
};
int stage_begin(Stage* s, rcv_ctx* ctx);
int stage_in(Stage* s, const rcv_mat* m, bool upload, bool copy_back, rcv_mat** dev_out);
int stage_finish(Stage* s, int rc);

static inline rcv_batch rcv_single(const rcv_mat* dev)
{
    rcv_batch b;
    b.frame0 = *dev;
    b.frame_stride = 0;
    b.n = 1;
    b.reserved = 0;
    return b;
}

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// Row-segment height for SMALL launches of the VALU-bound register-window kernels (one wave per (strip, segment, frame)); returns
// 0 when the launch is large (more than 4 waves per SIMD at 90-row segments: the caller's own plan for full GPUs applies).  When
// all waves of a launch are resident at once, every SIMD issues for its waves in turn and the launch takes ceil(waves / SIMDs)
// segment times -- not ceil(waves / resident slots); a SIMD with a single wave cannot hide its memory latency (x1.25), with two
// barely (x1.06).  `overhead` = rows a segment streams beyond its own (halo + pipeline fill + set-up).  Measured with the Harris
// pipeline on 1 / 4 / 8 / 16 4K frames: 0.046 -> 0.027, 0.070 -> 0.040, 0.098 -> 0.076, 0.158 -> 0.147 ms.
static inline int rcv_plan_seg_rows(int rows, long long waves_per_seg, int cu_count, int overhead, int min_seg, double lone = 1.25, double two = 1.06,
                                    int occ = 64, int max_per_simd = 4)
{
    // occ: waves of this kernel a SIMD holds at once (the row-streaming MFMA kernel: 2); more waves per SIMD run in rounds
    // lone / two: what one / two waves on a SIMD lose against three or more (they cannot hide each other's latencies).  The VALU
    // register-window kernels: 1.25 / 1.06; the row-streaming MFMA kernel, whose lone wave leaves the SIMD half idle between its
    // dependent MFMA chains: 2.3 / 1.17 (round 3, tools/config2_latency.py: one 4K frame 15.4 us at 32-row bands = one wave per
    // SIMD, 10.8 us at 16-row bands = two).
    const long long simds = 4LL * (cu_count > 0 ? cu_count : 256);
    if (waves_per_seg * ((rows + 89) / 90) > max_per_simd * simds) return 0;
    int seg = rows;
    double best = 1e300;
    for (int ns = 1; ns <= rows; ++ns) {
        const int sr = (rows + ns - 1) / ns;
        if (ns > 1 && sr < min_seg) break;
        const long long per_simd = (waves_per_seg * ((rows + sr - 1) / sr) + simds - 1) / simds;
        const auto pen = [&](long long w) { return w < 2 ? lone : (w < 3 ? two : 1.0); };
        const long long full = per_simd / occ, rest = per_simd % occ;   // rounds of `occ` resident waves + one partial round
        const double cost = (double)(sr + overhead) * ((double)full * occ * pen(occ) + (double)rest * (rest ? pen(rest) : 0.0));
        if (cost < best) {
            best = cost;
            seg = sr;
        }
    }
    return seg;
}
