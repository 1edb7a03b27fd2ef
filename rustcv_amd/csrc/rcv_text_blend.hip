// rcv_text_blend.hip -- the per-pixel half of put_text ("next" row f4).
//
// rustcv::imgproc::put_text (rustcv/src/imgproc/drawing.rs:123-163) lays a string out with rusttype, and for every
// positioned glyph calls `glyph.draw(|x, y, v| ...)`: the closure (:137-160) alpha-blends the colour into the Mat with
// coverage v at pixel (x + bb.min.x, y + bb.min.y), clipped to the Mat, storing truncated u8 values -- glyph after glyph,
// so where two boxes overlap the second blend reads what the first one stored.  Layout and rasterisation are third-
// party code on a font blob the checkout does not hold and stay on the host; the blend is per-pixel work
// on the Mat and runs here.
//
// One thread owns one Mat pixel of the union of the (clipped) glyph boxes and replays, in glyph order, every glyph
// whose box contains it; the pixel lives in registers between blends (u8 values, exactly what the reference reloads
// from the Mat), so the ordered semantics need neither atomics nor one launch per glyph.  The glyph loop is wave-
// uniform: the glyph table is read with scalar loads.  Arithmetic: the reference's three separately rounded f32
// operations per channel (__fmul_rn / __fsub_rn / __fadd_rn, never contracted) and Rust's `as u8` (saturating
// truncation, NaN -> 0).  A text line is a few hundred KB of coverage: the op is launch- and PCIe-latency bound,
// not an HBM-roofline row.
#include "rcv_internal.h"
#include <string.h>

namespace {

constexpr int kBlock = 64;
constexpr int kChunk = 1024;   // glyphs per launch (later chunks run after earlier ones: order preserved)

__device__ __forceinline__ uint32_t f32_as_u8(float v)
{
    // fmaxf(NaN, 0) == 0; the conversion truncates toward zero
    return (uint32_t)fminf(fmaxf(v, 0.0f), 255.0f);
}

__global__ __launch_bounds__(kBlock) void k_blend_glyphs(uint8_t* __restrict__ base, size_t fs, size_t step, int frames,
                                                        const rcv_glyph* __restrict__ gl, int n, const float* __restrict__ cov, int ux0,
                                                        int uy0, int ux1, int uy1, float cb, float cg, float cr)
{
    const int px = ux0 + (int)(blockIdx.x * kBlock + threadIdx.x);
    if (px >= ux1) return;
    for (int f = blockIdx.z; f < frames; f += gridDim.z)
        for (int py = uy0 + (int)blockIdx.y; py < uy1; py += gridDim.y) {
            uint8_t* p = base + (size_t)f * fs + (size_t)py * step + (size_t)px * 3;
            uint32_t b = 0, g = 0, r = 0;
            bool loaded = false;
            for (int i = 0; i < n; ++i) {
                const rcv_glyph G = gl[i];
                const long long dx = (long long)px - G.x, dy = (long long)py - G.y;
                if (dx >= 0 && dx < G.w && dy >= 0 && dy < G.h) {
                    if (!loaded) {
                        b = p[0], g = p[1], r = p[2];
                        loaded = true;
                    }
                    const float a = cov[G.offset + (unsigned long long)dy * (unsigned long long)G.w + (unsigned long long)dx];
                    const float inv = __fsub_rn(1.0f, a);
                    b = f32_as_u8(__fadd_rn(__fmul_rn(cb, a), __fmul_rn((float)b, inv)));
                    g = f32_as_u8(__fadd_rn(__fmul_rn(cg, a), __fmul_rn((float)g, inv)));
                    r = f32_as_u8(__fadd_rn(__fmul_rn(cr, a), __fmul_rn((float)r, inv)));
                }
            }
            if (loaded) {
                p[0] = (uint8_t)b;
                p[1] = (uint8_t)g;
                p[2] = (uint8_t)r;
            }
        }
}

// grow-only pinned staging for the glyph table + coverage; `pin_ev` marks the last H2D that read it
int pin_reserve(rcv_ctx* ctx, size_t bytes)
{
    if (!ctx->pin_ev) RCV_HIP(hipEventCreateWithFlags(&ctx->pin_ev, hipEventDisableTiming));
    else RCV_HIP(hipEventSynchronize(ctx->pin_ev));   // the previous call's upload has consumed the buffer
    if (bytes <= ctx->pin_cap) return RCV_OK;
    if (ctx->pin) RCV_HIP(hipHostFree(ctx->pin));
    ctx->pin = nullptr;
    ctx->pin_cap = 0;
    size_t cap = bytes + bytes / 2;
    RCV_HIP(hipHostMalloc((void**)&ctx->pin, cap, hipHostMallocDefault));
    ctx->pin_cap = cap;
    return RCV_OK;
}

} // namespace

extern "C" int rcv_blend_glyphs_batch(rcv_ctx* ctx, rcv_batch* mats, const rcv_glyph* glyphs, int32_t n_glyphs, const float* coverage,
                                      uint64_t n_coverage, uint8_t b, uint8_t g, uint8_t r)
{
    if (!mats || n_glyphs < 0 || (n_glyphs > 0 && !glyphs)) return RCV_ERR_ARG;
    RCV_TRY(rcv_bind(ctx));
    if (mats->frame0.device != RCV_DEVICE) return RCV_ERR_ARG;
    if (mats->frame0.channels != 3) return RCV_ERR_UNSUPPORTED;   // drawing.rs:132 hard-codes 3
    View m;
    RCV_TRY(rcv_view_batch(mats, RCV_8U, &m));
    // validate every box against the coverage array before anything is enqueued (the reference would index out of
    // bounds = panic; here: an error code and an untouched Mat)
    for (int i = 0; i < n_glyphs; ++i) {
        const rcv_glyph& G = glyphs[i];
        if (G.w < 0 || G.h < 0) return RCV_ERR_ARG;
        const uint64_t cnt = (uint64_t)G.w * (uint64_t)G.h;
        if (cnt && (!coverage || G.offset > n_coverage || cnt > n_coverage - G.offset)) return RCV_ERR_SIZE;
    }
    if (m.rows == 0 || m.cols == 0 || m.n == 0 || n_glyphs == 0) return RCV_OK;

    // keep only the boxes that meet the Mat, re-based onto a compacted coverage array holding just their values
    const size_t tbl_bytes = ((size_t)n_glyphs * sizeof(rcv_glyph) + 255) & ~(size_t)255;
    uint64_t used = 0;
    int kept = 0;
    for (int i = 0; i < n_glyphs; ++i) {
        const rcv_glyph& G = glyphs[i];
        if (G.w == 0 || G.h == 0) continue;
        if ((long long)G.x + G.w <= 0 || (long long)G.y + G.h <= 0 || G.x >= m.cols || G.y >= m.rows) continue;
        used += (uint64_t)G.w * (uint64_t)G.h;
        ++kept;
    }
    if (kept == 0) return RCV_OK;
    if (used > ((uint64_t)1 << 31)) return RCV_ERR_UNSUPPORTED;   // 8 GiB of coverage in one call
    const size_t cov_bytes = (size_t)used * sizeof(float), total = tbl_bytes + cov_bytes;
    RCV_TRY(pin_reserve(ctx, total));
    rcv_glyph* tbl = (rcv_glyph*)ctx->pin;
    float* cv = (float*)(ctx->pin + tbl_bytes);
    uint64_t off = 0;
    int k = 0;
    for (int i = 0; i < n_glyphs; ++i) {
        const rcv_glyph& G = glyphs[i];
        if (G.w == 0 || G.h == 0) continue;
        if ((long long)G.x + G.w <= 0 || (long long)G.y + G.h <= 0 || G.x >= m.cols || G.y >= m.rows) continue;
        const uint64_t cnt = (uint64_t)G.w * (uint64_t)G.h;
        memcpy(cv + off, coverage + G.offset, (size_t)cnt * sizeof(float));
        tbl[k] = G;
        tbl[k].offset = off;
        off += cnt;
        ++k;
    }
    RCV_TRY(rcv_ws_reserve(ctx, total + 256));
    uint8_t* dev = nullptr;
    RCV_TRY(rcv_ws_alloc(ctx, total, &dev));
    RCV_HIP(hipMemcpyAsync(dev, ctx->pin, total, hipMemcpyHostToDevice, ctx->stream));
    RCV_HIP(hipEventRecord(ctx->pin_ev, ctx->stream));
    const rcv_glyph* dtbl = (const rcv_glyph*)dev;
    const float* dcov = (const float*)(dev + tbl_bytes);

    for (int c0 = 0; c0 < kept; c0 += kChunk) {
        const int cn = kept - c0 < kChunk ? kept - c0 : kChunk;
        long long ux0 = m.cols, uy0 = m.rows, ux1 = 0, uy1 = 0;
        for (int i = c0; i < c0 + cn; ++i) {
            const rcv_glyph& G = tbl[i];
            const long long x0 = G.x > 0 ? G.x : 0, y0 = G.y > 0 ? G.y : 0;
            const long long x1 = (long long)G.x + G.w < m.cols ? (long long)G.x + G.w : m.cols;
            const long long y1 = (long long)G.y + G.h < m.rows ? (long long)G.y + G.h : m.rows;
            ux0 = x0 < ux0 ? x0 : ux0, uy0 = y0 < uy0 ? y0 : uy0;
            ux1 = x1 > ux1 ? x1 : ux1, uy1 = y1 > uy1 ? y1 : uy1;
        }
        const unsigned gy = (unsigned)(uy1 - uy0 < 65535 ? uy1 - uy0 : 65535), gz = (unsigned)(m.n < 65535 ? m.n : 65535);
        RCV_LAUNCH(k_blend_glyphs, dim3(cdiv((size_t)(ux1 - ux0), kBlock), gy, gz), dim3(kBlock), 0, ctx->stream, m.p, m.fstride,
                           m.step, m.n, dtbl + c0, cn, dcov, (int)ux0, (int)uy0, (int)ux1, (int)uy1, (float)b, (float)g, (float)r);
        RCV_TRY(rcv_launch_check(ctx));
    }
    return RCV_OK;
}

extern "C" int rcv_blend_glyphs(rcv_ctx* ctx, rcv_mat* mat, const rcv_glyph* glyphs, int32_t n_glyphs, const float* coverage,
                                uint64_t n_coverage, uint8_t b, uint8_t g, uint8_t r)
{
    if (!mat) return RCV_ERR_ARG;
    Stage st;
    RCV_TRY(stage_begin(&st, ctx));
    rcv_mat* dm;
    RCV_TRY(stage_in(&st, mat, true, true, &dm));
    rcv_batch bm = rcv_single(dm);
    int rc = rcv_blend_glyphs_batch(ctx, &bm, glyphs, n_glyphs, coverage, n_coverage, b, g, r);
    return stage_finish(&st, rc);
}
