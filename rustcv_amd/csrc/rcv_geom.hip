// rcv_geom.hip -- bilinear resize and warpAffine (u8, channels 1/3/4).  Not in the reference
// (SURVEY.md F1); semantics SURVEY.md 8-A == oracle/rcv_oracle.c orc_resize / orc_warp_affine.
// f32 evaluation order is fixed and spelled out op by op; built with -ffp-contract=off.
#include "rcv_internal.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ uint8_t round_half_up_u8(float v)
{
    int iv = (int)floorf(v + 0.5f);
    return (uint8_t)min(max(iv, 0), 255);
}

// one thread per output pixel; rows on blockIdx.y, frames on blockIdx.z
template <int CH>
__global__ __launch_bounds__(kBlock) void k_resize(View s, View d, float scx, float scy)
{
    int y = blockIdx.y;
    const uint8_t* sf = s.p + (size_t)blockIdx.z * s.fstride;
    uint8_t* drow = d.p + (size_t)blockIdx.z * d.fstride + (size_t)y * d.step;
    float sy = ((float)y + 0.5f) * scy - 0.5f;
    sy = sy < 0.0f ? 0.0f : sy;
    sy = sy > (float)(s.rows - 1) ? (float)(s.rows - 1) : sy;
    int y0 = (int)floorf(sy);
    float fy = sy - (float)y0;
    int y1 = y0 + 1 < s.rows ? y0 + 1 : s.rows - 1;
    const uint8_t* ra = sf + (size_t)y0 * s.step;
    const uint8_t* rb = sf + (size_t)y1 * s.step;
    for (int x = blockIdx.x * kBlock + threadIdx.x; x < d.cols; x += gridDim.x * kBlock) {
        float sx = ((float)x + 0.5f) * scx - 0.5f;
        sx = sx < 0.0f ? 0.0f : sx;
        sx = sx > (float)(s.cols - 1) ? (float)(s.cols - 1) : sx;
        int x0 = (int)floorf(sx);
        float fx = sx - (float)x0;
        int x1 = x0 + 1 < s.cols ? x0 + 1 : s.cols - 1;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float p00 = (float)ra[(size_t)x0 * CH + c], p01 = (float)ra[(size_t)x1 * CH + c];
            float p10 = (float)rb[(size_t)x0 * CH + c], p11 = (float)rb[(size_t)x1 * CH + c];
            float top = fmaf(fx, p01 - p00, p00);
            float bot = fmaf(fx, p11 - p10, p10);
            float v = fmaf(fy, bot - top, top);
            drow[(size_t)x * CH + c] = round_half_up_u8(v);
        }
    }
}

struct Affine { float m[6]; };

template <int CH>
__global__ __launch_bounds__(kBlock) void k_warp_affine(View s, View d, Affine A)
{
    int y = blockIdx.y;
    const uint8_t* sf = s.p + (size_t)blockIdx.z * s.fstride;
    uint8_t* drow = d.p + (size_t)blockIdx.z * d.fstride + (size_t)y * d.step;
    float fyy = (float)y;
    for (int x = blockIdx.x * kBlock + threadIdx.x; x < d.cols; x += gridDim.x * kBlock) {
        float fxx = (float)x;
        float sx = fmaf(A.m[0], fxx, fmaf(A.m[1], fyy, A.m[2]));
        float sy = fmaf(A.m[3], fxx, fmaf(A.m[4], fyy, A.m[5]));
        uint8_t* o = drow + (size_t)x * CH;
        if (!(sx > -1.0f && sx < (float)s.cols && sy > -1.0f && sy < (float)s.rows)) {
#pragma unroll
            for (int c = 0; c < CH; ++c) o[c] = 0;
            continue;
        }
        float x0f = floorf(sx), y0f = floorf(sy);
        int x0 = (int)x0f, y0 = (int)y0f;
        float fx = sx - x0f, fy = sy - y0f;
        int x1 = x0 + 1, y1 = y0 + 1;
        bool vx0 = x0 >= 0, vx1 = x1 < s.cols, vy0 = y0 >= 0, vy1 = y1 < s.rows;
        const uint8_t* ra = sf + (size_t)(vy0 ? y0 : 0) * s.step;
        const uint8_t* rb = sf + (size_t)(vy1 ? y1 : 0) * s.step;
        size_t xa = (size_t)(vx0 ? x0 : 0) * CH, xb = (size_t)(vx1 ? x1 : 0) * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float p00 = (vx0 && vy0) ? (float)ra[xa + c] : 0.0f;
            float p01 = (vx1 && vy0) ? (float)ra[xb + c] : 0.0f;
            float p10 = (vx0 && vy1) ? (float)rb[xa + c] : 0.0f;
            float p11 = (vx1 && vy1) ? (float)rb[xb + c] : 0.0f;
            float top = fmaf(fx, p01 - p00, p00);
            float bot = fmaf(fx, p11 - p10, p10);
            float v = fmaf(fy, bot - top, top);
            o[c] = round_half_up_u8(v);
        }
    }
}

int check_geom(const rcv_batch* src, rcv_batch* dst, View* s, View* d)
{
    if (!src || !dst) return RCV_ERR_ARG;
    RCV_TRY(rcv_view_batch(src, RCV_8U, s));
    RCV_TRY(rcv_view_batch(dst, RCV_8U, d));
    if (s->ch != d->ch || s->n != d->n) return RCV_ERR_ARG;
    if (s->ch != 1 && s->ch != 3 && s->ch != 4) return RCV_ERR_UNSUPPORTED;
    if (d->rows > 65535 || d->n > 65535) return RCV_ERR_UNSUPPORTED;
    return RCV_OK;
}

inline dim3 px_grid(const View& d)
{
    unsigned gx = (unsigned)((d.cols + kBlock - 1) / kBlock);
    if (gx > 1024) gx = 1024;
    return dim3(gx, d.rows, d.n);
}

} // namespace

extern "C" int rcv_resize_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst)
{
    RCV_TRY(rcv_bind(ctx));
    View s, d;
    RCV_TRY(check_geom(src, dst, &s, &d));
    if (d.rows == 0 || d.cols == 0 || d.n == 0) return RCV_OK;
    if (s.rows == 0 || s.cols == 0) return RCV_ERR_ARG;
    float scx = (float)s.cols / (float)d.cols, scy = (float)s.rows / (float)d.rows;
    if (s.ch == 1) hipLaunchKernelGGL(k_resize<1>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, scx, scy);
    else if (s.ch == 3) hipLaunchKernelGGL(k_resize<3>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, scx, scy);
    else hipLaunchKernelGGL(k_resize<4>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, scx, scy);
    return rcv_launch_check(ctx);
}

extern "C" int rcv_warp_affine_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst, const float* M)
{
    RCV_TRY(rcv_bind(ctx));
    if (!M) return RCV_ERR_ARG;
    View s, d;
    RCV_TRY(check_geom(src, dst, &s, &d));
    if (d.rows == 0 || d.cols == 0 || d.n == 0) return RCV_OK;
    Affine A;
    for (int i = 0; i < 6; ++i) A.m[i] = M[i];
    if (s.ch == 1) hipLaunchKernelGGL(k_warp_affine<1>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, A);
    else if (s.ch == 3) hipLaunchKernelGGL(k_warp_affine<3>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, A);
    else hipLaunchKernelGGL(k_warp_affine<4>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, A);
    return rcv_launch_check(ctx);
}

extern "C" int rcv_resize(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dst)
{
    if (!src || !dst) return RCV_ERR_ARG;
    Stage st;
    RCV_TRY(stage_begin(&st, ctx));
    rcv_mat *ds, *dd;
    RCV_TRY(stage_in(&st, src, true, false, &ds));
    RCV_TRY(stage_in(&st, dst, true, true, &dd));
    rcv_batch bs = rcv_single(ds), bd = rcv_single(dd);
    return stage_finish(&st, rcv_resize_batch(ctx, &bs, &bd));
}

extern "C" int rcv_warp_affine(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dst, const float* M)
{
    if (!src || !dst) return RCV_ERR_ARG;
    Stage st;
    RCV_TRY(stage_begin(&st, ctx));
    rcv_mat *ds, *dd;
    RCV_TRY(stage_in(&st, src, true, false, &ds));
    RCV_TRY(stage_in(&st, dst, true, true, &dd));
    rcv_batch bs = rcv_single(ds), bd = rcv_single(dd);
    return stage_finish(&st, rcv_warp_affine_batch(ctx, &bs, &bd, M));
}
