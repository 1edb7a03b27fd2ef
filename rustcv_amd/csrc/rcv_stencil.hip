// rcv_stencil.hip -- GaussianBlur, filter2D (i8 / f32 weights) and Sobel: argument checking,
// dispatch, and the GENERIC kernels (any size, any step, any channel count, ksize up to 15).
// The generic kernels read taps straight from global memory through L1/L2 and resolve
// BORDER_REFLECT_101 per tap; they are the correctness path for shapes the tiled kernels
// (rcv_filter7_mfma.hip, rcv_stencil_tiled.hip) do not take.  None of these ops exists in the
// reference; semantics are SURVEY.md 8-A, restated in oracle/rcv_oracle.c.
// Compiled with -ffp-contract=off: every fused multiply-add below is an explicit fmaf().
#include "rcv_internal.h"
#include "rcv_kernels.h"
#include "rcv_device_utils.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

struct TapsI32 { int t[15]; int ksize; int D; };
struct TapsF32 { float t[31]; int ksize; };
struct KernI8 { int8_t k[225]; int ksize; int shift; };
struct KernF32 { float k[225]; int ksize; float delta; };

// one thread per output sample (byte column xb = x*ch + c)
#define SAMPLE_LOOP_BEGIN                                                                        \
    int y = blockIdx.y;                                                                          \
    const uint8_t* sf = s.p + (size_t)blockIdx.z * s.fstride;                                    \
    uint8_t* df = d.p + (size_t)blockIdx.z * d.fstride;                                          \
    int rowb = s.cols * s.ch;                                                                    \
    for (int xb = blockIdx.x * kBlock + threadIdx.x; xb < rowb; xb += gridDim.x * kBlock) {      \
        int x = xb / s.ch, c = xb - x * s.ch;
#define SAMPLE_LOOP_END }
// same, restricted to byte columns [xlo, xhi)
#define SAMPLE_RANGE_BEGIN(xlo, xhi)                                                             \
    int y = blockIdx.y;                                                                          \
    const uint8_t* sf = s.p + (size_t)blockIdx.z * s.fstride;                                    \
    uint8_t* df = d.p + (size_t)blockIdx.z * d.fstride;                                          \
    for (int xb = (xlo) + blockIdx.x * kBlock + threadIdx.x; xb < (xhi); xb += gridDim.x * kBlock) { \
        int x = xb / s.ch, c = xb - x * s.ch;

__global__ __launch_bounds__(kBlock) void k_gauss_int_generic(View s, View d, TapsI32 tp)
{
    int r = tp.ksize / 2;
    SAMPLE_LOOP_BEGIN
    int acc = 0;
    for (int ky = 0; ky < tp.ksize; ++ky) {
        const uint8_t* row = sf + (size_t)reflect101(y + ky - r, s.rows) * s.step;
        int h = 0;
        for (int kx = 0; kx < tp.ksize; ++kx) h += tp.t[kx] * (int)row[(size_t)reflect101(x + kx - r, s.cols) * s.ch + c];
        acc += tp.t[ky] * h;
    }
    df[(size_t)y * d.step + xb] = (uint8_t)((acc + tp.D / 2) / tp.D);
    SAMPLE_LOOP_END
}

__global__ __launch_bounds__(kBlock) void k_gauss_f32_generic(View s, View d, TapsF32 tp, int xlo, int xhi)
{
    int r = tp.ksize / 2;
    SAMPLE_RANGE_BEGIN(xlo, xhi)
    float acc = 0.0f;
    for (int ky = 0; ky < tp.ksize; ++ky) {
        const uint8_t* row = sf + (size_t)reflect101(y + ky - r, s.rows) * s.step;
        float h = 0.0f; // horizontal pass of that row: fmaf chain in tap order from 0
        for (int kx = 0; kx < tp.ksize; ++kx) h = fmaf(tp.t[kx], (float)row[(size_t)reflect101(x + kx - r, s.cols) * s.ch + c], h);
        acc = fmaf(tp.t[ky], h, acc); // vertical pass
    }
    float v = rintf(acc);
    df[(size_t)y * d.step + xb] = v < 0.0f ? 0 : (v > 255.0f ? 255 : (uint8_t)v);
    SAMPLE_LOOP_END
}

__global__ __launch_bounds__(kBlock) void k_filter_i8_generic(View s, View d, KernI8 kw)
{
    int r = kw.ksize / 2;
    int rnd = kw.shift > 0 ? (1 << (kw.shift - 1)) : 0;
    SAMPLE_LOOP_BEGIN
    int acc = 0;
    for (int ky = 0; ky < kw.ksize; ++ky) {
        const uint8_t* row = sf + (size_t)reflect101(y + ky - r, s.rows) * s.step;
        for (int kx = 0; kx < kw.ksize; ++kx)
            acc += (int)kw.k[ky * kw.ksize + kx] * (int)row[(size_t)reflect101(x + kx - r, s.cols) * s.ch + c];
    }
    df[(size_t)y * d.step + xb] = (uint8_t)rcv_ashr_sat1(acc + rnd, kw.shift); // arithmetic shift == floor
    SAMPLE_LOOP_END
}

__global__ __launch_bounds__(kBlock) void k_filter_f32_generic(View s, View d, KernF32 kw, int xlo, int xhi)
{
    int r = kw.ksize / 2;
    SAMPLE_RANGE_BEGIN(xlo, xhi)
    float acc = kw.delta;
    for (int ky = 0; ky < kw.ksize; ++ky) {
        const uint8_t* row = sf + (size_t)reflect101(y + ky - r, s.rows) * s.step;
        for (int kx = 0; kx < kw.ksize; ++kx)
            acc = fmaf(kw.k[ky * kw.ksize + kx], (float)row[(size_t)reflect101(x + kx - r, s.cols) * s.ch + c], acc);
    }
    float v = rintf(acc);
    df[(size_t)y * d.step + xb] = v < 0.0f ? 0 : (v > 255.0f ? 255 : (uint8_t)v);
    SAMPLE_LOOP_END
}

__global__ __launch_bounds__(kBlock) void k_sobel_generic(View s, View dx, View dy)
{
    int y = blockIdx.y;
    const uint8_t* sf = s.p + (size_t)blockIdx.z * s.fstride;
    const uint8_t* r0 = sf + (size_t)reflect101(y - 1, s.rows) * s.step;
    const uint8_t* r1 = sf + (size_t)y * s.step;
    const uint8_t* r2 = sf + (size_t)reflect101(y + 1, s.rows) * s.step;
    int16_t* ox = (int16_t*)(dx.p + (size_t)blockIdx.z * dx.fstride + (size_t)y * dx.step);
    int16_t* oy = (int16_t*)(dy.p + (size_t)blockIdx.z * dy.fstride + (size_t)y * dy.step);
    for (int x = blockIdx.x * kBlock + threadIdx.x; x < s.cols; x += gridDim.x * kBlock) {
        int xl = reflect101(x - 1, s.cols), xr = reflect101(x + 1, s.cols);
        int gx = ((int)r0[xr] - (int)r0[xl]) + 2 * ((int)r1[xr] - (int)r1[xl]) + ((int)r2[xr] - (int)r2[xl]);
        int gy = ((int)r2[xl] - (int)r0[xl]) + 2 * ((int)r2[x] - (int)r0[x]) + ((int)r2[xr] - (int)r0[xr]);
        ox[x] = (int16_t)gx;
        oy[x] = (int16_t)gy;
    }
}

inline dim3 sample_grid(const View& v)
{
    size_t rowb = (size_t)v.cols * v.ch;
    unsigned gx = (unsigned)((rowb + kBlock - 1) / kBlock);
    if (gx > 1024) gx = 1024;
    return dim3(gx, v.rows, v.n);
}

int check_pair(const rcv_batch* src, rcv_batch* dst, View* s, View* d)
{
    if (!src || !dst) return RCV_ERR_ARG;
    RCV_TRY(rcv_view_batch(src, RCV_8U, s));
    RCV_TRY(rcv_view_batch(dst, RCV_8U, d));
    if (s->rows != d->rows || s->cols != d->cols || s->ch != d->ch || s->n != d->n) return RCV_ERR_ARG;
    if (s->rows > 65535 || s->n > 65535) return RCV_ERR_UNSUPPORTED;
    if (s->p == d->p && s->rows > 0 && s->cols > 0) return RCV_ERR_ARG; // stencils are not in-place
    return RCV_OK;
}

} // namespace

// ------------------------------------------------------------------------------------------------

extern "C" int rcv_gaussian_blur_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst, int ksize, double sigma)
{
    RCV_TRY(rcv_bind(ctx));
    View s, d;
    RCV_TRY(check_pair(src, dst, &s, &d));
    if (sigma <= 0.0) {
        if (ksize != 3 && ksize != 5 && ksize != 7) return RCV_ERR_ARG;
    } else if (!(ksize & 1) || ksize < 3 || ksize > 31) {
        return RCV_ERR_ARG;
    }
    if (s.rows == 0 || s.cols == 0 || s.n == 0) return RCV_OK;
    if (sigma <= 0.0) {
        int rc = rcv_gauss_int_rows(ctx, s, d, ksize);   // small launches (one 1080p frame: a latency problem)
        if (rc != RCV_ERR_UNSUPPORTED) return rc;
        rc = rcv_gauss_int_tiled(ctx, s, d, ksize);
        if (rc != RCV_ERR_UNSUPPORTED) return rc;
        static const int t3[3] = {1, 2, 1}, t5[5] = {1, 4, 6, 4, 1}, t7[7] = {2, 7, 14, 18, 14, 7, 2};
        TapsI32 tp;
        tp.ksize = ksize;
        const int* t = ksize == 3 ? t3 : (ksize == 5 ? t5 : t7);
        tp.D = ksize == 3 ? 16 : (ksize == 5 ? 256 : 4096);
        rc = rcv_gauss_int_stream(ctx, s, d, t, ksize, ksize == 3 ? 4 : (ksize == 5 ? 8 : 12));
        if (rc != RCV_ERR_UNSUPPORTED) return rc;
        for (int i = 0; i < ksize; ++i) tp.t[i] = t[i];
        RCV_LAUNCH(k_gauss_int_generic, sample_grid(s), dim3(kBlock), 0, ctx->stream, s, d, tp);
    } else {
        TapsF32 tp;
        tp.ksize = ksize;
        RCV_TRY(rcv_gaussian_taps_f32(ksize, sigma, tp.t));
        int rc = rcv_gauss_f32_fast(ctx, s, d, tp.t, ksize);
        if (rc != RCV_ERR_UNSUPPORTED) return rc;
        RCV_LAUNCH(k_gauss_f32_generic, sample_grid(s), dim3(kBlock), 0, ctx->stream, s, d, tp, 0, s.cols * s.ch);
    }
    return rcv_launch_check(ctx);
}

extern "C" int rcv_filter2d_i8_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst, const int8_t* k, int ksize, int shift)
{
    View s, d;
    {   // large BGR batches on the chained kernel: two halves on the context's two streams, nothing joined per call (rcv_internal.h: rcv_ctx::half)
        if (ctx && src && dst && k && (ksize == 3 || ksize == 5 || ksize == 7) && shift >= 0 && shift <= 24 && src->n >= 16 && check_pair(src, dst, &s, &d) == RCV_OK) {
            const int rc = rcv_filter_i8_split(ctx, s, d, k, ksize, shift);
            if (rc != RCV_ERR_UNSUPPORTED) return rc;
        }
    }
    RCV_TRY(rcv_bind(ctx));
    RCV_TRY(check_pair(src, dst, &s, &d));
    if (!k || !(ksize & 1) || ksize < 1 || ksize > 15 || shift < 0 || shift > 24) return RCV_ERR_ARG;
    if (s.rows == 0 || s.cols == 0 || s.n == 0) return RCV_OK;
    int rc = rcv_filter_i8_fast(ctx, s, d, k, ksize, shift);
    if (rc != RCV_ERR_UNSUPPORTED) return rc;
    if (ksize <= 7 && (s.ch == 1 || s.ch == 3)) {   // rows only 4-byte aligned, odd widths: the streaming kernel in integer mode
        int16_t k16[49];
        for (int i = 0; i < ksize * ksize; ++i) k16[i] = k[i];
        rc = rcv_filter_i16_stream(ctx, s, d, k16, ksize, shift);
        if (rc != RCV_ERR_UNSUPPORTED) return rc;
    }
    KernI8 kw;
    kw.ksize = ksize;
    kw.shift = shift;
    for (int i = 0; i < ksize * ksize; ++i) kw.k[i] = k[i];
    RCV_LAUNCH(k_filter_i8_generic, sample_grid(s), dim3(kBlock), 0, ctx->stream, s, d, kw);
    return rcv_launch_check(ctx);
}

extern "C" int rcv_filter2d_i8_yuyv_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst, const int8_t* k, int ksize, int shift)
{
    RCV_TRY(rcv_bind(ctx));
    if (!src || !dst) return RCV_ERR_ARG;
    View s, d;
    RCV_TRY(rcv_view_batch(src, RCV_8U, &s));
    RCV_TRY(rcv_view_batch(dst, RCV_8U, &d));
    if (s.ch != 2 || d.ch != 3) return RCV_ERR_UNSUPPORTED;
    if (s.rows != d.rows || s.cols != d.cols || s.n != d.n || (s.cols & 1)) return RCV_ERR_ARG;
    if (!k || !(ksize & 1) || ksize < 1 || ksize > 15 || shift < 0 || shift > 24) return RCV_ERR_ARG;
    if (s.rows > 65535 || s.n > 65535) return RCV_ERR_UNSUPPORTED;
    if (s.rows == 0 || s.cols == 0 || s.n == 0) return RCV_OK;
    int rc = rcv_filter_i8_yuyv_fast(ctx, s, d, k, ksize, shift);
    if (rc != RCV_ERR_UNSUPPORTED) return rc;
    // unfused HIP path: convert into the workspace, then the ordinary filter
    const size_t tstep = ((size_t)s.cols * 3 + 15) & ~(size_t)15, tfs = tstep * s.rows;
    RCV_TRY(rcv_ws_reserve(ctx, tfs * s.n + 512));
    uint8_t* tmp;
    RCV_TRY(rcv_ws_alloc(ctx, tfs * s.n, &tmp));
    rcv_batch tb = *dst;
    tb.frame0.data = tmp;
    tb.frame0.cap = tfs;
    tb.frame0.step = tstep;
    tb.frame_stride = tfs;
    RCV_TRY(rcv_cvt_color_batch(ctx, RCV_YUYV2BGR_STRIDED, src, &tb));
    return rcv_filter2d_i8_batch(ctx, &tb, dst, k, ksize, shift);
}

extern "C" int rcv_filter2d_f32_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst, const float* k, int ksize, float delta)
{
    RCV_TRY(rcv_bind(ctx));
    View s, d;
    RCV_TRY(check_pair(src, dst, &s, &d));
    if (!k || !(ksize & 1) || ksize < 1 || ksize > 15) return RCV_ERR_ARG;
    if (s.rows == 0 || s.cols == 0 || s.n == 0) return RCV_OK;
    int rc = rcv_filter_f32_fast(ctx, s, d, k, ksize, delta);
    if (rc != RCV_ERR_UNSUPPORTED) return rc;
    KernF32 kw;
    kw.ksize = ksize;
    kw.delta = delta;
    for (int i = 0; i < ksize * ksize; ++i) kw.k[i] = k[i];
    RCV_LAUNCH(k_filter_f32_generic, sample_grid(s), dim3(kBlock), 0, ctx->stream, s, d, kw, 0, s.cols * s.ch);
    return rcv_launch_check(ctx);
}

extern "C" int rcv_sobel_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dx, rcv_batch* dy)
{
    // (as two halves on the context's two streams -- rcv_split_run -- this call measures +0.3 %: not split; tools/ab_split_ops.py)
    RCV_TRY(rcv_bind(ctx));
    if (!src || !dx || !dy) return RCV_ERR_ARG;
    View s, vx, vy;
    RCV_TRY(rcv_view_batch(src, RCV_8U, &s));
    RCV_TRY(rcv_view_batch(dx, RCV_16S, &vx));
    RCV_TRY(rcv_view_batch(dy, RCV_16S, &vy));
    if ((s.ch != 1 && s.ch != 3) || vx.ch != 1 || vy.ch != 1) return RCV_ERR_UNSUPPORTED;
    if (s.rows != vx.rows || s.cols != vx.cols || s.n != vx.n || s.rows != vy.rows || s.cols != vy.cols || s.n != vy.n)
        return RCV_ERR_ARG;
    if (s.rows > 65535 || s.n > 65535) return RCV_ERR_UNSUPPORTED;
    if (s.rows == 0 || s.cols == 0 || s.n == 0) return RCV_OK;
    int rc = rcv_sobel_tiled(ctx, s, vx, vy);   // 1 channel, or BGR: gradient of its gray conversion in one launch
    if (rc != RCV_ERR_UNSUPPORTED) return rc;
    if (s.ch == 3) {
        // unfused HIP path: gray into the workspace, then the ordinary Sobel
        const size_t tstep = ((size_t)s.cols + 15) & ~(size_t)15, tfs = tstep * s.rows;
        RCV_TRY(rcv_ws_reserve(ctx, tfs * s.n + 512));
        uint8_t* tmp;
        RCV_TRY(rcv_ws_alloc(ctx, tfs * s.n, &tmp));
        rcv_batch tb = *src;
        tb.frame0.data = tmp;
        tb.frame0.cap = tfs;
        tb.frame0.step = tstep;
        tb.frame0.channels = 1;
        tb.frame_stride = tfs;
        RCV_TRY(rcv_cvt_color_batch(ctx, RCV_BGR2GRAY, src, &tb));
        return rcv_sobel_batch(ctx, &tb, dx, dy);
    }
    RCV_LAUNCH(k_sobel_generic, sample_grid(s), dim3(kBlock), 0, ctx->stream, s, vx, vy);
    return rcv_launch_check(ctx);
}

// filter2D (i8 weights) -> BGR2GRAY -> Sobel.  One launch of the row-streaming MFMA kernel where the shape allows
// (rcv_filter_rows_mfma.hip, SOB instantiation); otherwise the two ordinary calls with the filtered image in the side buffer.
extern "C" int rcv_filter2d_i8_sobel_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dx, rcv_batch* dy, const int8_t* k, int ksize, int shift)
{
    // (as two halves on the context's two streams this call measures +6.7 %: its three waves per SIMD already cover the tail; not split)
    RCV_TRY(rcv_bind(ctx));
    if (!src || !dx || !dy) return RCV_ERR_ARG;
    View s, vx, vy;
    RCV_TRY(rcv_view_batch(src, RCV_8U, &s));
    RCV_TRY(rcv_view_batch(dx, RCV_16S, &vx));
    RCV_TRY(rcv_view_batch(dy, RCV_16S, &vy));
    if (s.ch != 3 || vx.ch != 1 || vy.ch != 1) return RCV_ERR_UNSUPPORTED;
    if (s.rows != vx.rows || s.cols != vx.cols || s.n != vx.n || s.rows != vy.rows || s.cols != vy.cols || s.n != vy.n) return RCV_ERR_ARG;
    if (!k || !(ksize & 1) || ksize < 1 || ksize > 15 || shift < 0 || shift > 24) return RCV_ERR_ARG;
    if (s.rows > 65535 || s.n > 65535) return RCV_ERR_UNSUPPORTED;
    if (s.rows == 0 || s.cols == 0 || s.n == 0) return RCV_OK;
    if (ksize == 3 || ksize == 5 || ksize == 7) {
        int16_t k16[49];
        for (int i = 0; i < ksize * ksize; ++i) k16[i] = k[i];
        const int rc = rcv_filter_i16_rows(ctx, s, s, k16, ksize, shift, 0, true, &vx, &vy);   // (`d` is not written: the source stands in)
        if (rc != RCV_ERR_UNSUPPORTED) return rc;
    }
    const size_t tstep = ((size_t)s.cols * 3 + 15) & ~(size_t)15, tfs = tstep * s.rows;
    uint8_t* tmp;
    RCV_TRY(rcv_side_reserve(ctx, tfs * s.n, &tmp));
    rcv_batch tb = *src;
    tb.frame0.data = tmp;
    tb.frame0.cap = tfs;
    tb.frame0.step = tstep;
    tb.frame0.device = 1;
    tb.frame_stride = tfs;
    RCV_TRY(rcv_filter2d_i8_batch(ctx, src, &tb, k, ksize, shift));
    return rcv_sobel_batch(ctx, &tb, dx, dy);
}

int rcv_filter_f32_generic_range(rcv_ctx* ctx, const View& s, const View& d, const float* k, int ksize, float delta, int xb_lo, int xb_hi)
{
    if (xb_hi <= xb_lo) return RCV_OK;
    KernF32 kw;
    kw.ksize = ksize;
    kw.delta = delta;
    for (int i = 0; i < ksize * ksize; ++i) kw.k[i] = k[i];
    dim3 grid((unsigned)((xb_hi - xb_lo + kBlock - 1) / kBlock), s.rows, s.n);
    RCV_LAUNCH(k_filter_f32_generic, grid, dim3(kBlock), 0, ctx->stream, s, d, kw, xb_lo, xb_hi);
    return rcv_launch_check(ctx);
}

int rcv_gauss_f32_generic_range(rcv_ctx* ctx, const View& s, const View& d, const float* taps, int ksize, int xb_lo, int xb_hi)
{
    if (xb_hi <= xb_lo) return RCV_OK;
    TapsF32 tp;
    tp.ksize = ksize;
    for (int i = 0; i < ksize; ++i) tp.t[i] = taps[i];
    dim3 grid((unsigned)((xb_hi - xb_lo + kBlock - 1) / kBlock), s.rows, s.n);
    RCV_LAUNCH(k_gauss_f32_generic, grid, dim3(kBlock), 0, ctx->stream, s, d, tp, xb_lo, xb_hi);
    return rcv_launch_check(ctx);
}

// ---- single-Mat forms (host or device mats) -------------------------------------------------------

#define RCV_UNARY_WRAPPER(call)                         \
    if (!src || !dst) return RCV_ERR_ARG;               \
    if (src->data == dst->data && src->data) return RCV_ERR_ARG; /* stencils are not in-place */ \
    Stage st;                                           \
    RCV_TRY(stage_begin(&st, ctx));                     \
    rcv_mat *ds, *dd;                                   \
    RCV_TRY(stage_in(&st, src, true, false, &ds));      \
    RCV_TRY(stage_in(&st, dst, true, true, &dd));       \
    rcv_batch bs = rcv_single(ds), bd = rcv_single(dd); \
    int rc = call;                                      \
    return stage_finish(&st, rc);

extern "C" int rcv_gaussian_blur(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dst, int ksize, double sigma)
{
    RCV_UNARY_WRAPPER(rcv_gaussian_blur_batch(ctx, &bs, &bd, ksize, sigma))
}

extern "C" int rcv_filter2d_i8(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dst, const int8_t* k, int ksize, int shift)
{
    RCV_UNARY_WRAPPER(rcv_filter2d_i8_batch(ctx, &bs, &bd, k, ksize, shift))
}

extern "C" int rcv_filter2d_i8_yuyv(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dst, const int8_t* k, int ksize, int shift)
{
    RCV_UNARY_WRAPPER(rcv_filter2d_i8_yuyv_batch(ctx, &bs, &bd, k, ksize, shift))
}

extern "C" int rcv_filter2d_f32(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dst, const float* k, int ksize, float delta)
{
    RCV_UNARY_WRAPPER(rcv_filter2d_f32_batch(ctx, &bs, &bd, k, ksize, delta))
}

extern "C" int rcv_filter2d_i8_sobel(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dx, rcv_mat* dy, const int8_t* k, int ksize, int shift)
{
    if (!src || !dx || !dy) return RCV_ERR_ARG;
    Stage st;
    RCV_TRY(stage_begin(&st, ctx));
    rcv_mat *ds, *d1, *d2;
    RCV_TRY(stage_in(&st, src, true, false, &ds));
    RCV_TRY(stage_in(&st, dx, true, true, &d1));
    RCV_TRY(stage_in(&st, dy, true, true, &d2));
    rcv_batch bs = rcv_single(ds), b1 = rcv_single(d1), b2 = rcv_single(d2);
    int rc = rcv_filter2d_i8_sobel_batch(ctx, &bs, &b1, &b2, k, ksize, shift);
    return stage_finish(&st, rc);
}

extern "C" int rcv_sobel(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dx, rcv_mat* dy)
{
    if (!src || !dx || !dy) return RCV_ERR_ARG;
    Stage st;
    RCV_TRY(stage_begin(&st, ctx));
    rcv_mat *ds, *d1, *d2;
    RCV_TRY(stage_in(&st, src, true, false, &ds));
    RCV_TRY(stage_in(&st, dx, true, true, &d1));
    RCV_TRY(stage_in(&st, dy, true, true, &d2));
    rcv_batch bs = rcv_single(ds), b1 = rcv_single(d1), b2 = rcv_single(d2);
    int rc = rcv_sobel_batch(ctx, &bs, &b1, &b2);
    return stage_finish(&st, rc);
}
