// rcv_ctx.hip -- context, device memory, staging and host-only helpers of the C ABI.
// include/rustcv_hip.h documents every entry point; precedent for the conventions is
// /root/reference rustcv-camera/src/backend/macos/bridge.h:16-65.
#include "rcv_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <new>
#include <vector>

extern "C" int rcv_abi_version(void) { return RCV_ABI_VERSION; }

// ---- environment knobs, read once ----
static RcvKnobs g_knobs;
static std::once_flag g_knobs_once;
static int env_int(const char* name, int unset)
{
    const char* e = getenv(name);
    return e ? atoi(e) : unset;
}
static void load_knobs()
{
    g_knobs.f7_rows = env_int("RCV_F7_ROWS", -1);
    g_knobs.f7_no_lat = getenv("RCV_F7_NO_LAT") != nullptr;
    g_knobs.f7_no_gray = getenv("RCV_F7_NO_GRAY") != nullptr;
    g_knobs.f7_dual_full = getenv("RCV_F7_DUAL_FULL") != nullptr;
    g_knobs.fr_chain = env_int("RCV_FR_CHAIN", -1);
    g_knobs.fr_chain_rows = env_int("RCV_FR_CHAIN_ROWS", 0);
    g_knobs.fr_split = env_int("RCV_FR_SPLIT", -1);
    g_knobs.fr_chain_drop_xcd = env_int("RCV_FR_CHAIN_DROP_XCD", -1);
    g_knobs.gauss_rows = env_int("RCV_GAUSS_ROWS", -1);
    g_knobs.gr_seg = env_int("RCV_GR_SEG", 0);
    g_knobs.harris_general = env_int("RCV_HARRIS_GENERAL", 0);
    g_knobs.warp_lds = env_int("RCV_WARP_LDS", -1);
    g_knobs.warp_gray4 = env_int("RCV_WARP_GRAY4", -1);
    g_knobs.warp_fpg = env_int("RCV_WARP_FPG", 0);
}
const RcvKnobs& rcv_knobs()
{
    std::call_once(g_knobs_once, load_knobs);   // DeviceGroup's per-GPU threads may all arrive here first
    return g_knobs;
}
// tests and tuning tools only, after setenv: the caller must have no launch in flight on another thread
extern "C" void rcv__debug_reload_knobs(void)
{
    std::call_once(g_knobs_once, [] {});
    load_knobs();
}

// ---- per-thread log of launched kernels (tests: which kernel did this entry point dispatch?) ----
static thread_local char t_kernels[1024];
static thread_local size_t t_kernels_len = 0;
void rcv_note_kernel(const char* name)
{
    const size_t n = strlen(name);
    if (t_kernels_len + n + 2 >= sizeof(t_kernels)) return;   // full: keep the first ones
    if (t_kernels_len) t_kernels[t_kernels_len++] = ';';
    memcpy(t_kernels + t_kernels_len, name, n);
    t_kernels_len += n;
    t_kernels[t_kernels_len] = 0;
}
extern "C" const char* rcv__debug_kernels(void) { return t_kernels; }
extern "C" void rcv__debug_kernels_reset(void)
{
    t_kernels_len = 0;
    t_kernels[0] = 0;
}

extern "C" const char* rcv_strerror(int code)
{
    switch (code) {
    case RCV_OK: return "ok";
    case RCV_NOOP: return "ok (reference length guard: silent no-op)";
    case RCV_ERR_ARG: return "invalid argument";
    case RCV_ERR_UNSUPPORTED: return "unsupported format or parameter";
    case RCV_ERR_SIZE: return "buffer too small for the described image";
    case RCV_ERR_DEVICE: return "HIP device error (no gfx950 device, or a runtime/launch failure)";
    case RCV_ERR_OOM: return "out of memory";
    case RCV_ERR_BUSY: return "staging ring full (retire a frame first)";
    default: return "unknown rustcv_hip status";
    }
}

extern "C" int rcv_device_count(int* n)
{
    if (!n) return RCV_ERR_ARG;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *n = 0;
        return RCV_ERR_DEVICE;
    }
    *n = c;
    return RCV_OK;
}

// Every live context, for one question of the split launch: does ANOTHER context of this device have work in flight right now (a caller that keeps
// two batches in flight covers the launches' tails already)?  Asked with hipStreamQuery on the other contexts' streams -- their handles never change
// after creation, and the query is thread-safe; a hint: no correctness depends on the answer.
static std::mutex g_reg_mu;
static std::vector<rcv_ctx*> g_reg;
bool rcv_other_context_busy(const rcv_ctx* me)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    for (const rcv_ctx* o : g_reg) {
        if (o == me || o->device != me->device) continue;
        if ((o->stream && hipStreamQuery(o->stream) == hipErrorNotReady) || (o->half && hipStreamQuery(o->half) == hipErrorNotReady)) {
            (void)hipGetLastError();
            return true;
        }
    }
    (void)hipGetLastError();
    return false;
}

extern "C" int rcv_ctx_create(int device, rcv_ctx** out)
{
    if (!out) return RCV_ERR_ARG;
    *out = nullptr;
    int count = 0;
    RCV_TRY(rcv_device_count(&count));
    if (device < 0 || device >= count) return RCV_ERR_DEVICE;
    RCV_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    RCV_HIP(hipGetDeviceProperties(&prop, device));
    // gfx950 only: the kernels use CDNA4 instructions and sizes; refuse anything else loudly.
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RCV_ERR_DEVICE;
    rcv_ctx* c = new (std::nothrow) rcv_ctx();
    if (!c) return RCV_ERR_OOM;
    memset(c, 0, sizeof(*c));
    c->device = device;
    c->cu_count = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->half, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_half, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipMalloc((void**)&c->kconst, RCV_KC_BYTES);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rcv_ctx_destroy(c);
        return e == hipErrorOutOfMemory ? RCV_ERR_OOM : RCV_ERR_DEVICE;
    }
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        g_reg.push_back(c);
    }
    *out = c;
    return RCV_OK;
}

// Graphs and staging rings keep a pointer to their context (device, stream).  Destroying the context first -- easy from a
// garbage-collected host language -- must not leave them dangling: the context then only drains its stream and stays
// allocated until its last child is destroyed.
static void ctx_finalize(rcv_ctx* c);

extern "C" void rcv_ctx_destroy(rcv_ctx* c)
{
    if (!c) return;
    if (c->children > 0) {
        (void)hipSetDevice(c->device);
        if (c->half) (void)hipStreamSynchronize(c->half);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        c->zombie = true;
        return;
    }
    ctx_finalize(c);
}

void rcv_ctx_child_released(rcv_ctx* c)
{
    if (!c) return;
    if (c->children > 0) c->children--;
    if (c->zombie && c->children == 0) ctx_finalize(c);
}

static void ctx_finalize(rcv_ctx* c)
{
    (void)hipSetDevice(c->device);
    if (c->half) (void)hipStreamSynchronize(c->half);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        for (size_t i = 0; i < g_reg.size(); ++i)
            if (g_reg[i] == c) {
                g_reg.erase(g_reg.begin() + (long)i);
                break;
            }
    }
    if (c->ws) (void)hipFree(c->ws);
    if (c->tmp2) (void)hipFree(c->tmp2);
    for (int i = 0; i < RCV_MAX_STAGE; ++i)
        if (c->stage_buf[i]) (void)hipFree(c->stage_buf[i]);
    if (c->kconst) (void)hipFree(c->kconst);
    if (c->fr_tabs) (void)hipFree(c->fr_tabs);
    for (int e = 0; e < 4; ++e)
        if (c->fr_tab[e].uploaded) (void)hipEventDestroy(c->fr_tab[e].uploaded);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->fr_fault) (void)hipHostFree(c->fr_fault);
    if (c->pin_ev) (void)hipEventDestroy(c->pin_ev);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->half) (void)hipStreamDestroy(c->half);
    if (c->ev_half) (void)hipEventDestroy(c->ev_half);
    if (c->ev_main) (void)hipEventDestroy(c->ev_main);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int rcv_bind_raw(rcv_ctx* ctx)
{
    if (!ctx || ctx->zombie) return RCV_ERR_ARG;
    RCV_HIP(hipSetDevice(ctx->device));
    return RCV_OK;
}

// `stream` waits for everything `half` holds: after this the context's stream alone orders all of its work again
int rcv_join_half(rcv_ctx* ctx)
{
    if (!ctx->half_busy) return RCV_OK;
    RCV_HIP(hipEventRecord(ctx->ev_half, ctx->half));
    RCV_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_half, 0));
    ctx->half_busy = false;
    ctx->half_r.n = ctx->half_w.n = 0;
    return RCV_OK;
}

// ---- split calls: range bookkeeping ----
static bool hit(const rcv_ctx::Hull& a, const rcv_ctx::Hull& b) { return a.lo < a.hi && b.lo < b.hi && a.lo < b.hi && b.lo < a.hi; }
static bool hit(const rcv_ctx::HullSet& s, const rcv_ctx::Hull& x)
{
    for (int i = 0; i < s.n; ++i)
        if (hit(s.h[i], x)) return true;
    return false;
}
static void add(rcv_ctx::HullSet& s, const rcv_ctx::Hull& x)
{
    for (int i = 0; i < s.n; ++i)
        if (s.h[i].lo <= x.lo && x.hi <= s.h[i].hi) return;            // (the steady state: the same ranges call after call)
    if (s.n < 6) {
        s.h[s.n++] = x;
        return;
    }
    rcv_ctx::Hull m = x;                                               // full: everything becomes one hull
    for (int i = 0; i < s.n; ++i) {
        m.lo = s.h[i].lo < m.lo ? s.h[i].lo : m.lo;
        m.hi = s.h[i].hi > m.hi ? s.h[i].hi : m.hi;
    }
    s.h[0] = m;
    s.n = 1;
}
// does what `x` wants to do collide with what is pending in (pr, pw)?  write-after-read, write-after-write, read-after-write
static bool collides(const RcvRanges& x, const rcv_ctx::HullSet& pr, const rcv_ctx::HullSet& pw)
{
    for (int i = 0; i < x.nw; ++i)
        if (hit(pr, x.w[i]) || hit(pw, x.w[i])) return true;
    for (int i = 0; i < x.nr; ++i)
        if (hit(pw, x.r[i])) return true;
    return false;
}

int rcv_split_begin(rcv_ctx* ctx, int n, const RcvRanges& a, const RcvRanges& b)
{
    if (!ctx || ctx->zombie || n < 16 || rcv_knobs().fr_split == 0 || ctx->stream_exported) return RCV_ERR_UNSUPPORTED;
    // the halves of THIS call must not depend on each other (overlapping frames, in-place)
    for (int i = 0; i < a.nw; ++i) {
        for (int j = 0; j < b.nw; ++j)
            if (hit(a.w[i], b.w[j])) return RCV_ERR_UNSUPPORTED;
        for (int j = 0; j < b.nr; ++j)
            if (hit(a.w[i], b.r[j])) return RCV_ERR_UNSUPPORTED;
    }
    for (int i = 0; i < b.nw; ++i)
        for (int j = 0; j < a.nr; ++j)
            if (hit(b.w[i], a.r[j])) return RCV_ERR_UNSUPPORTED;
    RCV_TRY(rcv_bind_raw(ctx));
    if (rcv_other_context_busy(ctx)) return RCV_ERR_UNSUPPORTED;   // someone else keeps the GPU's tail busy already: two batches in flight
    // half A on `stream` must not touch what `half` still holds
    if (ctx->half_busy && collides(a, ctx->half_r, ctx->half_w)) RCV_TRY(rcv_join_half(ctx));
    return RCV_OK;
}

int rcv_split_half(rcv_ctx* ctx, const RcvRanges& a, const RcvRanges& b, unsigned uploads_before)
{
    // half B on `half`: behind everything on `stream` that it could depend on -- work of other entry points (main_unknown), a table uploaded a
    // moment ago, a pending split launch whose ranges it touches (half A of this very call was checked against B in rcv_split_begin)
    const bool fork = ctx->main_unknown || ctx->fr_uploads != uploads_before || collides(b, ctx->main_r, ctx->main_w);
    if (fork) {
        hipError_t e = hipEventRecord(ctx->ev_main, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->half, ctx->ev_main, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return RCV_ERR_DEVICE;
        }
        ctx->main_unknown = false;
        ctx->main_r.n = ctx->main_w.n = 0;   // (`half` has waited for all of it, half A of this call included)
    }
    for (int i = 0; i < a.nr; ++i) add(ctx->main_r, a.r[i]);
    for (int i = 0; i < a.nw; ++i) add(ctx->main_w, a.w[i]);
    return RCV_OK;
}

void rcv_split_done(rcv_ctx* ctx, const RcvRanges& b)
{
    ctx->half_busy = true;
    for (int i = 0; i < b.nr; ++i) add(ctx->half_r, b.r[i]);
    for (int i = 0; i < b.nw; ++i) add(ctx->half_w, b.w[i]);
}

// Every entry point but the split launch itself comes through here: whatever it enqueues on `stream` is ordered behind both halves of every
// earlier split call, and the half stream will wait for `stream` before it runs anything again (main_unknown).
int rcv_bind(rcv_ctx* ctx)
{
    RCV_TRY(rcv_bind_raw(ctx));
    RCV_TRY(rcv_join_half(ctx));
    ctx->main_unknown = true;
    return RCV_OK;
}

int rcv_launch_check(rcv_ctx*)
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? RCV_OK : RCV_ERR_DEVICE;
}

// Every host-side wait for the context's stream goes through here: the chained row filter's last launch gets its completion check
// enqueued first, and a fault that any check has raised comes back as RCV_ERR_DEVICE (rcv_filter_rows_mfma.hip: rcv_chain_flush / _poll).
int rcv_wait(rcv_ctx* ctx)
{
    const int jrc = rcv_join_half(ctx);   // (callers have come through rcv_bind: normally nothing left to join)
    const int frc = rcv_chain_flush(ctx);
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess || jrc < 0) {
        (void)hipGetLastError();
        if (ctx->half) (void)hipStreamSynchronize(ctx->half);
        for (rcv_ctx::ChainLane& l : ctx->fr_lane) {
            l.tickets_ready = false;   // whatever ran last may have left a counter set half-drawn: the next chained launch zeroes them all
            l.unchecked = false;
        }
        return RCV_ERR_DEVICE;
    }
    ctx->main_r.n = ctx->main_w.n = 0;
    RCV_TRY(frc);
    return rcv_chain_poll(ctx);
}

extern "C" int rcv_sync(rcv_ctx* ctx)
{
    RCV_TRY(rcv_bind(ctx));
    return rcv_wait(ctx);
}

extern "C" int rcv_ctx_device(const rcv_ctx* ctx) { return ctx ? ctx->device : RCV_ERR_ARG; }
// (the caller may enqueue its own work on the stream from now on: the context stops running calls as two halves on two streams)
extern "C" void* rcv_ctx_stream(const rcv_ctx* ctx)
{
    if (!ctx) return nullptr;
    rcv_ctx* c = const_cast<rcv_ctx*>(ctx);
    c->stream_exported = true;
    if (c->half_busy && hipSetDevice(c->device) == hipSuccess) (void)rcv_join_half(c);
    return (void*)ctx->stream;
}

extern "C" int rcv_malloc(rcv_ctx* ctx, size_t bytes, void** out)
{
    if (!out) return RCV_ERR_ARG;
    *out = nullptr;
    RCV_TRY(rcv_bind(ctx));
    if (bytes == 0) bytes = 16;
    RCV_HIP(hipMalloc(out, bytes));
    return RCV_OK;
}

extern "C" int rcv_free(rcv_ctx* ctx, void* p)
{
    RCV_TRY(rcv_bind(ctx));
    if (!p) return RCV_OK;
    RCV_HIP(hipStreamSynchronize(ctx->stream));
    RCV_HIP(hipFree(p));
    return RCV_OK;
}

extern "C" int rcv_upload(rcv_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    RCV_TRY(rcv_bind(ctx));
    if (bytes == 0) return RCV_OK;
    if (!dst || !src) return RCV_ERR_ARG;
    RCV_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return rcv_wait(ctx);
}

extern "C" int rcv_download(rcv_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    RCV_TRY(rcv_bind(ctx));
    if (bytes == 0) return RCV_OK;
    if (!dst || !src) return RCV_ERR_ARG;
    RCV_TRY(rcv_chain_flush(ctx));
    RCV_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return rcv_wait(ctx);
}

extern "C" int rcv_memset(rcv_ctx* ctx, void* dst, int value, size_t bytes)
{
    RCV_TRY(rcv_bind(ctx));
    if (bytes == 0) return RCV_OK;
    if (!dst) return RCV_ERR_ARG;
    RCV_HIP(hipMemsetAsync(dst, value, bytes, ctx->stream));
    return RCV_OK;
}

extern "C" int rcv_timer_start(rcv_ctx* ctx)
{
    RCV_TRY(rcv_bind(ctx));
    RCV_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return RCV_OK;
}

extern "C" int rcv_timer_stop(rcv_ctx* ctx, float* ms)
{
    if (!ms) return RCV_ERR_ARG;
    RCV_TRY(rcv_bind(ctx));
    RCV_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    RCV_HIP(hipEventSynchronize(ctx->ev1));
    RCV_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return rcv_chain_poll(ctx);   // (the timed window is not lengthened by a check launch; earlier launches have checked each other)
}

// ---- host-only helpers --------------------------------------------------------------------

static constexpr uint32_t fourcc(char a, char b, char c, char d)
{
    // rustcv-core/src/pixel_format.rs:10-12 : little-endian ASCII
    return (uint32_t)(uint8_t)a | ((uint32_t)(uint8_t)b << 8) | ((uint32_t)(uint8_t)c << 16) | ((uint32_t)(uint8_t)d << 24);
}

extern "C" int rcv_fourcc_to_code(uint32_t fcc, int* code)
{
    if (!code) return RCV_ERR_ARG;
    // rustcv/src/videoio/mod.rs:201-206 (YUYV, BGRA) ; rustcv-camera/src/pixel_format.rs:84-92 (RGB3, BGR4)
    if (fcc == fourcc('Y', 'U', 'Y', 'V')) { *code = RCV_YUYV2BGR; return RCV_OK; }
    if (fcc == fourcc('B', 'G', 'R', 'A') || fcc == fourcc('B', 'G', 'R', '4')) { *code = RCV_BGRA2BGR; return RCV_OK; }
    if (fcc == fourcc('R', 'G', 'B', '3')) { *code = RCV_RGB2BGR; return RCV_OK; }
    return RCV_ERR_UNSUPPORTED;
}

extern "C" int rcv_gaussian_taps_f32(int ksize, double sigma, float* taps)
{
    if (!taps || !(ksize & 1) || ksize < 3 || ksize > 31 || !(sigma > 0.0)) return RCV_ERR_ARG;
    double t[32], sum = 0.0;
    int r = ksize / 2;
    for (int i = 0; i < ksize; ++i) {
        double x = (double)(i - r);
        t[i] = exp(-(x * x) / (2.0 * sigma * sigma));
        sum += t[i];
    }
    for (int i = 0; i < ksize; ++i) taps[i] = (float)(t[i] / sum);
    return RCV_OK;
}

// ---- workspace ------------------------------------------------------------------------------
// Kernel-internal temporaries (unfused intermediates).  rcv_ws_reserve(total) first, then carve.

int rcv_ws_reserve(rcv_ctx* ctx, size_t total)
{
    ctx->ws_off = 0;
    if (total > ctx->ws_cap) {
        RCV_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->ws) RCV_HIP(hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_cap = 0;
        RCV_HIP(hipMalloc((void**)&ctx->ws, total));
        ctx->ws_cap = total;
    }
    return RCV_OK;
}

int rcv_side_reserve(rcv_ctx* ctx, size_t bytes, uint8_t** out)
{
    if (bytes > ctx->tmp2_cap) {
        RCV_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->tmp2) RCV_HIP(hipFree(ctx->tmp2));
        ctx->tmp2 = nullptr;
        ctx->tmp2_cap = 0;
        RCV_HIP(hipMalloc((void**)&ctx->tmp2, bytes));
        ctx->tmp2_cap = bytes;
    }
    *out = ctx->tmp2;
    return RCV_OK;
}

int rcv_ws_alloc(rcv_ctx* ctx, size_t bytes, uint8_t** out)
{
    size_t off = (ctx->ws_off + 255) & ~(size_t)255;
    if (!ctx->ws || off + bytes > ctx->ws_cap) return RCV_ERR_OOM;
    *out = ctx->ws + off;
    ctx->ws_off = off + bytes;
    return RCV_OK;
}

int rcv_upload_const(rcv_ctx* ctx, const void* host, size_t bytes, size_t offset)
{
    if (offset + bytes > 65536) return RCV_ERR_ARG;
    // pageable source: hipMemcpyAsync snapshots it before returning, and the copy is
    // ordered on the ctx stream ahead of the kernel that reads it.
    RCV_HIP(hipMemcpyAsync(ctx->kconst + offset, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return RCV_OK;
}

int rcv_const_table(rcv_ctx* ctx, const void* host, size_t bytes, size_t offset, const uint8_t** dev)
{
    RCV_TRY(rcv_upload_const(ctx, host, bytes, offset));
    *dev = ctx->kconst + offset;
    return RCV_OK;
}

// ---- views ----------------------------------------------------------------------------------

int rcv_view_strided(const rcv_mat* m, int want_depth, View* v)
{
    if (!m || !v) return RCV_ERR_ARG;
    if (m->rows < 0 || m->cols < 0) return RCV_ERR_ARG;
    if (m->depth != want_depth) return RCV_ERR_UNSUPPORTED;
    int esz = rcv_elem_size(m->depth);
    if (esz == 0 || m->channels == 0) return RCV_ERR_ARG;
    size_t rowb = (size_t)m->cols * m->channels * esz;
    if (m->rows > 0 && m->cols > 0) {
        if (!m->data) return RCV_ERR_ARG;
        if (m->step < rowb) return RCV_ERR_SIZE;
        if (m->step % esz) return RCV_ERR_ARG;
        size_t need = (size_t)(m->rows - 1) * m->step + rowb;
        if (m->cap < need) return RCV_ERR_SIZE;
    }
    v->p = (uint8_t*)m->data;
    v->step = m->step;
    v->fstride = 0;
    v->cap = m->cap;
    v->rows = m->rows;
    v->cols = m->cols;
    v->ch = m->channels;
    v->esz = esz;
    v->n = 1;
    return RCV_OK;
}

int rcv_view_batch(const rcv_batch* b, int want_depth, View* v)
{
    if (!b) return RCV_ERR_ARG;
    if (b->n < 0) return RCV_ERR_ARG;
    if (b->frame0.device != RCV_DEVICE) return RCV_ERR_ARG;
    RCV_TRY(rcv_view_strided(&b->frame0, want_depth, v));
    v->n = b->n;
    v->fstride = b->frame_stride;
    if (b->n > 1 && v->rows > 0 && v->cols > 0) {
        size_t need = (size_t)(v->rows - 1) * v->step + (size_t)v->cols * v->ch * v->esz;
        if (b->frame_stride < need) return RCV_ERR_SIZE;
    }
    return RCV_OK;
}

// ---- host staging ---------------------------------------------------------------------------
// A host Mat is mirrored whole (all `cap` bytes == Vec::len()) into a per-ctx, grow-only
// staging buffer, so the reference's length guards see exactly the caller's capacity.

int stage_begin(Stage* s, rcv_ctx* ctx)
{
    RCV_TRY(rcv_bind(ctx));
    s->ctx = ctx;
    s->count = 0;
    s->any_host = false;
    return RCV_OK;
}

int stage_in(Stage* s, const rcv_mat* m, bool upload, bool copy_back, rcv_mat** dev_out)
{
    if (!m || !dev_out) return RCV_ERR_ARG;
    if (s->count >= RCV_MAX_STAGE) return RCV_ERR_ARG;
    rcv_ctx* ctx = s->ctx;
    StagedMat* sm = &s->m[s->count];
    sm->host = m;
    sm->dev = *m;
    sm->copy_back = false;
    if (m->device == RCV_DEVICE) {
        if (s->any_host) return RCV_ERR_ARG; // all mats of one call live on the same side
    } else if (m->device == RCV_HOST) {
        if (s->count > 0 && !s->any_host) return RCV_ERR_ARG;
        s->any_host = true;
        size_t bytes = m->cap;
        if (bytes > 0 && !m->data) return RCV_ERR_ARG;
        int slot = s->count;
        if (ctx->stage_cap[slot] < bytes + 64) {
            RCV_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->stage_buf[slot]) RCV_HIP(hipFree(ctx->stage_buf[slot]));
            ctx->stage_buf[slot] = nullptr;
            ctx->stage_cap[slot] = 0;
            size_t want = bytes + 64 + (bytes >> 3);
            RCV_HIP(hipMalloc((void**)&ctx->stage_buf[slot], want));
            ctx->stage_cap[slot] = want;
        }
        if (upload && bytes)
            RCV_HIP(hipMemcpyAsync(ctx->stage_buf[slot], m->data, bytes, hipMemcpyHostToDevice, ctx->stream));
        sm->dev.data = ctx->stage_buf[slot];
        sm->dev.device = RCV_DEVICE;
        sm->copy_back = copy_back;
    } else {
        return RCV_ERR_ARG;
    }
    *dev_out = &sm->dev;
    s->count++;
    return RCV_OK;
}

int stage_finish(Stage* s, int rc)
{
    if (!s->any_host) return rc;
    rcv_ctx* ctx = s->ctx;
    if (rc >= 0) {
        for (int i = 0; i < s->count; ++i) {
            StagedMat* sm = &s->m[i];
            if (!sm->copy_back || sm->host->cap == 0) continue;
            RCV_HIP(hipMemcpyAsync(sm->host->data, sm->dev.data, sm->host->cap, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    const int wrc = rcv_wait(ctx);
    return rc < 0 ? rc : (wrc < 0 ? wrc : rc);
}
