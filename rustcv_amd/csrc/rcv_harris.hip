// rcv_harris.hip -- cornerHarris response, 3x3 NMS and the BGR->mask pipeline: argument
// checking, dispatch and the GENERIC (unfused, workspace-backed) kernels.  The fused
// single-launch pipeline lives in rcv_harris_fused.hip.  Not in the reference;
// semantics SURVEY.md 8-A == oracle/rcv_oracle.c.  Six separate IEEE f32 ops for the response,
// no contraction (-ffp-contract=off).
#include "rcv_internal.h"
#include "rcv_kernels.h"
#include <math.h>

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// ix/iy: packed i16 planes [n][rows][cols]
__global__ __launch_bounds__(kBlock) void k_sobel_packed(View s, int16_t* ix, int16_t* iy)
{
    int y = blockIdx.y;
    const uint8_t* sf = s.p + (size_t)blockIdx.z * s.fstride;
    const uint8_t* r0 = sf + (size_t)reflect101(y - 1, s.rows) * s.step;
    const uint8_t* r1 = sf + (size_t)y * s.step;
    const uint8_t* r2 = sf + (size_t)reflect101(y + 1, s.rows) * s.step;
    size_t base = ((size_t)blockIdx.z * s.rows + y) * s.cols;
    for (int x = blockIdx.x * kBlock + threadIdx.x; x < s.cols; x += gridDim.x * kBlock) {
        int xl = reflect101(x - 1, s.cols), xr = reflect101(x + 1, s.cols);
        ix[base + x] = (int16_t)(((int)r0[xr] - (int)r0[xl]) + 2 * ((int)r1[xr] - (int)r1[xl]) + ((int)r2[xr] - (int)r2[xl]));
        iy[base + x] = (int16_t)(((int)r2[xl] - (int)r0[xl]) + 2 * ((int)r2[x] - (int)r0[x]) + ((int)r2[xr] - (int)r0[xr]));
    }
}

__global__ __launch_bounds__(kBlock) void k_harris_resp(const int16_t* ix, const int16_t* iy, View r, int block, float s2, float k)
{
    int y = blockIdx.y, a = block / 2;
    size_t fbase = (size_t)blockIdx.z * r.rows * r.cols;
    float* out = (float*)(r.p + (size_t)blockIdx.z * r.fstride + (size_t)y * r.step);
    for (int x = blockIdx.x * kBlock + threadIdx.x; x < r.cols; x += gridDim.x * kBlock) {
        int sxx = 0, sxy = 0, syy = 0;
        for (int by = 0; by < block; ++by) {
            size_t rb = fbase + (size_t)reflect101(y + by - a, r.rows) * r.cols;
            for (int bx = 0; bx < block; ++bx) {
                int xx = reflect101(x + bx - a, r.cols);
                int gx = ix[rb + xx], gy = iy[rb + xx];
                sxx += gx * gx;
                sxy += gx * gy;
                syy += gy * gy;
            }
        }
        float fa = (float)sxx * s2, fb = (float)sxy * s2, fc = (float)syy * s2;
        float t1 = fa * fc, t2 = fb * fb, t3 = fa + fc;
        float t4 = k * t3;
        float t5 = t4 * t3;
        out[x] = (t1 - t2) - t5;
    }
}

__global__ __launch_bounds__(kBlock) void k_nms3x3(View r, View m, float thr)
{
    int y = blockIdx.y;
    const uint8_t* rf = r.p + (size_t)blockIdx.z * r.fstride;
    uint8_t* mrow = m.p + (size_t)blockIdx.z * m.fstride + (size_t)y * m.step;
    for (int x = blockIdx.x * kBlock + threadIdx.x; x < r.cols; x += gridDim.x * kBlock) {
        float v = *(const float*)(rf + (size_t)y * r.step + (size_t)x * 4);
        bool keep = v > thr;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                int yy = y + dy, xx = x + dx;
                if ((dx == 0 && dy == 0) || yy < 0 || yy >= r.rows || xx < 0 || xx >= r.cols) continue;
                float nb = *(const float*)(rf + (size_t)yy * r.step + (size_t)xx * 4);
                if (!(v >= nb)) keep = false;
            }
        mrow[x] = keep ? 255 : 0;
    }
}

// Streaming 3x3 NMS for 16-byte aligned f32 rows: one thread owns 4 adjacent pixels and walks down a row segment; every
// response row is read once (one 16-byte load per lane, the two outer neighbours by DPP wave shifts), its 3-wide row maximum
// serves the rows above and below, and the comparison `r >= all 8 neighbours` becomes r >= maximum(...).  The maxima are
// v_maximum3_f32 (IEEE-754-2019 maximum: a NaN operand gives NaN), so a NaN neighbour makes `r >= m` false exactly as the
// nine separate comparisons of k_nms3x3 / the oracle do; outside the image is -inf.
// RAG instantiation: any width >= 4 and response rows that are only 4-byte aligned (an odd width of a packed f32 image), mask
// rows of any alignment: dword-aligned 16-byte loads (same cost as aligned ones), the lane that holds the row's last, partial
// group takes its pixels from the clamped load by a per-lane shift and stores only its valid bytes; unaligned dword stores.
template <bool RAG>
__global__ __launch_bounds__(kBlock) void k_nms3x3_rows(View r, View m, float thr, int seg_rows, int gx, int gy, int nblocks, int blocks_per_xcd)
{
    const int lane = threadIdx.x & 63;
    // Block order (speed only): hardware places block b on XCD b % 8; with blocks_per_xcd > 0 each XCD works through its own
    // contiguous eighth of the (frame, row segment, column block) list, so that what ONE XCD has in flight is a compact address
    // range (DESIGN_HISTORY.md 6)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (blocks_per_xcd > 0) {
        const int t = (int)(blockIdx.x & 7) * blocks_per_xcd + (int)(blockIdx.x >> 3);
        if (t >= nblocks) return;
        bz = t / (gx * gy);
        const int rem = t - bz * gx * gy;
        by = rem / gx;
        bx = rem - by * gx;
    }
    const int x = (bx * kBlock + (int)threadIdx.x) * 4;
    const int ys = by * seg_rows, ye = min(r.rows, ys + seg_rows);
    const uint8_t* rf = r.p + (size_t)bz * r.fstride;
    uint8_t* mf = m.p + (size_t)bz * m.fstride;
    const bool live = x < r.cols;
    const int xc = min(x, r.cols - 4);
    const int shift = RAG ? max(x - xc, 0) : 0, nvalid = RAG ? min(max(r.cols - x, 0), 4) : 4;   // (RAG) the row's last group: `nvalid` pixels at offset `shift` of the load
    const float NEG = -INFINITY;
    // the pixel outside the wave's 256: lane 0 needs x-1, lane 63 needs x+4 (every other lane reads one fixed cached address)
    const int xe = lane == 0 ? x - 1 : x + 4;
    const bool e_ok = (lane == 0 || lane == 63) && xe >= 0 && xe < r.cols;
    const int xec = e_ok ? xe : 0;
    struct Row { float4 q; float e; };
    auto load = [&](int v) -> Row {
        const int vr = min(max(v, 0), r.rows - 1);
        const uint8_t* row = rf + (size_t)vr * r.step;
        Row w;
        if constexpr (RAG) {
            typedef float f4m __attribute__((ext_vector_type(4), aligned(4)));
            const f4m t = *(const f4m*)(row + (size_t)xc * 4);
            w.q = make_float4(t.x, t.y, t.z, t.w);
        } else {
            w.q = *(const float4*)(row + (size_t)xc * 4);
        }
        w.e = *(const float*)(row + (size_t)xec * 4);
        return w;
    };
    auto mx3 = [](float a, float b, float c) { return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c); };
    float rm3a[4], rm3b[4], lrb[4], rcb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rm3a[j] = rm3b[j] = lrb[j] = rcb[j] = NEG;
    Row cur = load(ys - 1);
    for (int v = ys - 1; v <= ye; ++v) {
        const Row nxt = load(v + 1);   // (rows past the image re-read the last row and are replaced by -inf below)
        const bool rowok = v >= 0 && v < r.rows;
        float q[4] = {cur.q.x, cur.q.y, cur.q.z, cur.q.w};
        float e = cur.e;
        if (RAG) {   // logical pixel j of the lane = loaded pixel j + shift; beyond the row: -inf
            const float l0 = q[0], l1 = q[1], l2 = q[2], l3 = q[3];
            q[0] = shift == 0 ? l0 : (shift == 1 ? l1 : (shift == 2 ? l2 : l3));
            q[1] = shift == 0 ? l1 : (shift == 1 ? l2 : l3);
            q[2] = shift == 0 ? l2 : l3;
#pragma unroll
            for (int j = 1; j < 4; ++j) q[j] = j < nvalid ? q[j] : NEG;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = (rowok && live) ? q[j] : NEG;
        e = (rowok && e_ok) ? e : NEG;
        float L = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(uint32_t, q[3]), 0x138, 0xf, 0xf, true));   // lane-1's last pixel
        float R = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(uint32_t, q[0]), 0x130, 0xf, 0xf, true));   // lane+1's first pixel
        if (lane == 0) L = e;
        if (lane == 63) R = e;
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lf = j ? q[j - 1] : L, rt = j < 3 ? q[j + 1] : R;
            const float rm3 = mx3(lf, q[j], rt), lr = __builtin_elementwise_maximum(lf, rt);
            const float m8 = mx3(rm3a[j], lrb[j], rm3);            // the 8 neighbours of row v-1's pixel
            const bool keep = rcb[j] > thr && rcb[j] >= m8;
            bits |= keep ? (0xffu << (8 * j)) : 0u;
            rm3a[j] = rm3b[j];
            rm3b[j] = rm3;
            lrb[j] = lr;
            rcb[j] = q[j];
        }
        const int c = v - 1;
        if (live && c >= ys && c < ye) {
            uint8_t* o = mf + (size_t)c * m.step + x;
            if constexpr (RAG) {
                typedef uint32_t u1m __attribute__((aligned(1)));
                if (nvalid == 4) *(u1m*)o = bits;
                else
                    for (int j = 0; j < nvalid; ++j) o[j] = (uint8_t)(bits >> (8 * j));
            } else {
                *(uint32_t*)o = bits;
            }
        }
        cur = nxt;
    }
}

inline dim3 px_grid(const View& d)
{
    unsigned gx = (unsigned)((d.cols + kBlock - 1) / kBlock);
    if (gx > 1024) gx = 1024;
    return dim3(gx, d.rows, d.n);
}

float harris_scale2(int block)
{
    double s = 1.0 / (4.0 * (double)block * 255.0); // 2^(aperture-1) * blockSize * 255, aperture = 3
    return (float)(s * s);
}

// Sobel planes of `src` (gray or BGR: the gradient of its gray conversion) as i16 images with 16-byte aligned rows in the two
// workspace buffers, then the register-window response kernel for any block size.  RCV_ERR_UNSUPPORTED when either kernel does
// not take the shape (the caller then runs the per-sample kernels on the same buffers).
size_t plane_step(int cols) { return rcv_harris_plane_step(cols); }
int harris_fast_planes(rcv_ctx* ctx, const View& src, const View* r, const View* m, int block, float k, float thr, uint8_t* wix, uint8_t* wiy)
{
    const View& o = r ? *r : *m;
    if ((r && !rcv_harris_resp_rows_ok(*r, block, true)) || (m && !rcv_harris_resp_rows_ok(*m, block, false)) || src.rows > 65535 ||
        (src.ch != 1 && src.ch != 3))
        return RCV_ERR_UNSUPPORTED;
    View vx;
    vx.p = wix + rcv_harris_plane_margin();   // column 0 (the kernel's margins lie either side of the row)
    vx.step = plane_step(o.cols);
    vx.fstride = vx.step * o.rows;
    vx.cap = vx.fstride;
    vx.rows = o.rows;
    vx.cols = o.cols;
    vx.ch = 1;
    vx.esz = 2;
    vx.n = o.n;
    View vy = vx;
    vy.p = wiy + rcv_harris_plane_margin();
    const int rc = rcv_sobel_tiled(ctx, src, vx, vy);
    if (rc != RCV_OK) return rc;
    return rcv_harris_resp_rows(ctx, vx, vy, r, m, block, k, thr);
}

// 3x3 NMS: the streaming kernel for 16-byte aligned response rows, the per-sample kernel otherwise
int nms_launch(rcv_ctx* ctx, const View& r, const View& m, float thr)
{
    if (r.cols >= 4 && (uintptr_t)r.p % 4 == 0 && r.step % 4 == 0 && (r.n <= 1 || r.fstride % 4 == 0)) {
        const bool aligned = r.cols % 4 == 0 && (uintptr_t)r.p % 16 == 0 && r.step % 16 == 0 && (r.n <= 1 || r.fstride % 16 == 0) &&
                             (uintptr_t)m.p % 4 == 0 && m.step % 4 == 0 && (m.n <= 1 || m.fstride % 4 == 0);
        // small launches (a few frames): segments sized so that every SIMD has a wave (rcv_plan_seg_rows)
        const int gx0 = ((r.cols + 3) / 4 + kBlock - 1) / kBlock;
        const int small = rcv_plan_seg_rows(r.rows, 4LL * gx0 * r.n, ctx->cu_count, 6, 8);
        // (launches that fill the GPU: 16-row segments -- round 3, tools/ablate_segs.py on 64 4K frames: 16 rows 0.509 ms, 32 rows 0.538,
        //  64 rows 0.554: with the XCD-contiguous block order short segments keep what one XCD reads at a time compact)
        int seg = small > 0 && small < 64 ? small : 16;
        dim3 grid((unsigned)gx0, (unsigned)((r.rows + seg - 1) / seg), r.n);
        const int gx = (int)grid.x, gy = (int)grid.y;
        const unsigned long long nb = (unsigned long long)grid.x * grid.y * grid.z;
        int bpx = 0;
        if (nb < (1ull << 30)) {
            bpx = (int)((nb + 7) / 8);
            grid = dim3((unsigned)bpx * 8);
        }
        if (aligned) RCV_LAUNCH(k_nms3x3_rows<false>, grid, dim3(kBlock), 0, ctx->stream, r, m, thr, seg, gx, gy, (int)nb, bpx);
        else RCV_LAUNCH(k_nms3x3_rows<true>, grid, dim3(kBlock), 0, ctx->stream, r, m, thr, seg, gx, gy, (int)nb, bpx);
        return rcv_launch_check(ctx);
    }
    RCV_LAUNCH(k_nms3x3, px_grid(r), dim3(kBlock), 0, ctx->stream, r, m, thr);
    return rcv_launch_check(ctx);
}

// gray (device View) -> resp using two packed i16 planes carved from the workspace
int harris_from_gray(rcv_ctx* ctx, const View& g, const View& r, int block, float k, uint8_t* wix, uint8_t* wiy)
{
    RCV_LAUNCH(k_sobel_packed, px_grid(g), dim3(kBlock), 0, ctx->stream, g, (int16_t*)wix, (int16_t*)wiy);
    RCV_LAUNCH(k_harris_resp, px_grid(r), dim3(kBlock), 0, ctx->stream, (const int16_t*)wix, (const int16_t*)wiy, r, block,
                       harris_scale2(block), k);
    return rcv_launch_check(ctx);
}

} // namespace

extern "C" int rcv_corner_harris_batch(rcv_ctx* ctx, const rcv_batch* gray, rcv_batch* resp, int block, float k)
{
    RCV_TRY(rcv_bind(ctx));
    if (!gray || !resp) return RCV_ERR_ARG;
    if (block < 1 || block > 7) return RCV_ERR_ARG;
    View g, r;
    RCV_TRY(rcv_view_batch(gray, RCV_8U, &g));
    RCV_TRY(rcv_view_batch(resp, RCV_32F, &r));
    if (g.ch != 1 || r.ch != 1) return RCV_ERR_UNSUPPORTED;
    if (g.rows != r.rows || g.cols != r.cols || g.n != r.n) return RCV_ERR_ARG;
    if (g.rows > 65535 || g.n > 65535) return RCV_ERR_UNSUPPORTED;
    if (g.rows == 0 || g.cols == 0 || g.n == 0) return RCV_OK;
    {   // block 2 on aligned shapes: the fused register-window kernel, response only
        const int rc = rcv_harris_fused(ctx, g, nullptr, &r, block, k, 0.0f);
        if (rc != RCV_ERR_UNSUPPORTED) return rc;
    }
    {   // any other block size on aligned shapes: one launch (gray -> Sobel -> window sums -> response)
        const int rc = rcv_harris_blocks_fused(ctx, g, &r, nullptr, block, k, 0.0f);
        if (rc != RCV_ERR_UNSUPPORTED) return rc;
    }
    size_t plane = (size_t)g.n * g.rows * plane_step(g.cols);   // (>= the packed layout of the per-sample kernels)
    RCV_TRY(rcv_ws_reserve(ctx, 2 * (plane + 256)));
    uint8_t *wix, *wiy;
    RCV_TRY(rcv_ws_alloc(ctx, plane, &wix));
    RCV_TRY(rcv_ws_alloc(ctx, plane, &wiy));
    {   // other block sizes: streaming Sobel into aligned planes + the register-window response kernel
        const int rc = harris_fast_planes(ctx, g, &r, nullptr, block, k, 0.0f, wix, wiy);
        if (rc != RCV_ERR_UNSUPPORTED) return rc;
    }
    return harris_from_gray(ctx, g, r, block, k, wix, wiy);
}

extern "C" int rcv_nms3x3_batch(rcv_ctx* ctx, const rcv_batch* resp, rcv_batch* mask, float thr)
{
    RCV_TRY(rcv_bind(ctx));
    if (!resp || !mask) return RCV_ERR_ARG;
    View r, m;
    RCV_TRY(rcv_view_batch(resp, RCV_32F, &r));
    RCV_TRY(rcv_view_batch(mask, RCV_8U, &m));
    if (r.ch != 1 || m.ch != 1) return RCV_ERR_UNSUPPORTED;
    if (m.rows != r.rows || m.cols != r.cols || m.n != r.n) return RCV_ERR_ARG;
    if (r.rows > 65535 || r.n > 65535) return RCV_ERR_UNSUPPORTED;
    if (r.rows == 0 || r.cols == 0 || r.n == 0) return RCV_OK;
    return nms_launch(ctx, r, m, thr);
}

extern "C" int rcv_harris_pipeline_batch(rcv_ctx* ctx, const rcv_batch* bgr, rcv_batch* mask, rcv_batch* resp,
                                         int block, float k, float thr)
{
    {   // large batches on the fused kernel: two halves on the context's two streams (rcv_internal.h: rcv_ctx::half)
        View s, m, r;
        if (ctx && bgr && mask && bgr->n >= 16 && block >= 1 && block <= 7 && rcv_knobs().harris_general <= 0 && rcv_view_batch(bgr, RCV_8U, &s) == RCV_OK &&
            rcv_view_batch(mask, RCV_8U, &m) == RCV_OK && (s.ch == 3 || s.ch == 2 || s.ch == 1) && m.ch == 1 && s.rows == m.rows && s.cols == m.cols && s.n == m.n &&
            !(s.ch == 2 && (s.cols & 1)) && s.rows > 0 && s.cols > 0 && s.rows <= 65535 && s.n <= 65535 &&
            (!resp || (rcv_view_batch(resp, RCV_32F, &r) == RCV_OK && r.ch == 1 && r.rows == s.rows && r.cols == s.cols && r.n == s.n))) {
            const int h = s.n / 2;
            RcvRanges a, b;
            a.read(s, 0, h); a.write(m, 0, h);
            b.read(s, h, s.n); b.write(m, h, s.n);
            if (resp) { a.write(r, 0, h); b.write(r, h, s.n); }
            const int rc = rcv_split_run(ctx, s.n, a, b, [&](int half) {
                const int f0 = half ? h : 0, f1 = half ? s.n : h;
                const View sv = rcv_view_frames(s, f0, f1), mv = rcv_view_frames(m, f0, f1), rv = resp ? rcv_view_frames(r, f0, f1) : View{};
                return rcv_harris_fused(ctx, sv, &mv, resp ? &rv : nullptr, block, k, thr);
            });
            if (rc != RCV_ERR_UNSUPPORTED) return rc;
        }
    }
    RCV_TRY(rcv_bind(ctx));
    if (!bgr || !mask) return RCV_ERR_ARG;
    if (block < 1 || block > 7) return RCV_ERR_ARG;
    View s, m, r;
    RCV_TRY(rcv_view_batch(bgr, RCV_8U, &s));
    RCV_TRY(rcv_view_batch(mask, RCV_8U, &m));
    if ((s.ch != 3 && s.ch != 2 && s.ch != 1) || m.ch != 1) return RCV_ERR_UNSUPPORTED;   // BGR, packed YUYV (2 channels), or gray
    if (s.rows != m.rows || s.cols != m.cols || s.n != m.n) return RCV_ERR_ARG;
    if (s.ch == 2 && (s.cols & 1)) return RCV_ERR_ARG;
    if (resp) {
        RCV_TRY(rcv_view_batch(resp, RCV_32F, &r));
        if (r.ch != 1) return RCV_ERR_UNSUPPORTED;
        if (r.rows != s.rows || r.cols != s.cols || r.n != s.n) return RCV_ERR_ARG;
    }
    if (s.rows > 65535 || s.n > 65535) return RCV_ERR_UNSUPPORTED;
    if (s.rows == 0 || s.cols == 0 || s.n == 0) return RCV_OK;
    if (rcv_knobs().harris_general > 0 && s.ch != 2) {   // (A/B and tests: the general-block kernel at blockSize 2 as well)
        const int rcg = rcv_harris_blocks_fused(ctx, s, resp ? &r : nullptr, &m, block, k, thr);
        if (rcg != RCV_ERR_UNSUPPORTED) return rcg;
    }
    int rc = rcv_harris_fused(ctx, s, &m, resp ? &r : nullptr, block, k, thr);
    if (rc != RCV_ERR_UNSUPPORTED) return rc;
    if (s.ch == 2) {
        // YUYV shapes the fused kernel does not take: convert into a side buffer (not the workspace, which the BGR pipeline
        // below may carve for itself), then run the BGR pipeline
        const size_t tstep = ((size_t)s.cols * 3 + 15) & ~(size_t)15, tfs = tstep * s.rows;
        uint8_t* tmp;
        RCV_TRY(rcv_side_reserve(ctx, tfs * s.n, &tmp));
        rcv_batch tb = *bgr;
        tb.frame0.data = tmp;
        tb.frame0.cap = tfs;
        tb.frame0.step = tstep;
        tb.frame0.channels = 3;
        tb.frame_stride = tfs;
        RCV_TRY(rcv_cvt_color_batch(ctx, RCV_YUYV2BGR_STRIDED, bgr, &tb));
        return rcv_harris_pipeline_batch(ctx, &tb, mask, resp, block, k, thr);
    }
    {   // any other block size on aligned shapes: one launch (gray conversion, Sobel, window sums, response, NMS)
        const int rc3 = rcv_harris_blocks_fused(ctx, s, resp ? &r : nullptr, &m, block, k, thr);
        if (rc3 != RCV_ERR_UNSUPPORTED) return rc3;
    }
    // other shapes / block sizes: Ix, Iy (and the response when the caller does not want it, and the gray image for the
    // per-sample kernels) live in the workspace
    const size_t npx = (size_t)s.n * s.rows * s.cols;
    const size_t plane = (size_t)s.n * s.rows * plane_step(s.cols);        // i16 plane with margins and 16-byte aligned rows (>= the packed layout)
    const size_t rstep = ((size_t)s.cols * 4 + 15) & ~(size_t)15;          // response rows likewise
    RCV_TRY(rcv_ws_reserve(ctx, npx + 2 * plane + (resp ? 0 : rstep * s.rows * s.n) + 5 * 256));
    uint8_t *wg, *wix, *wiy, *wr = nullptr;
    RCV_TRY(rcv_ws_alloc(ctx, npx, &wg));   // (only the per-sample path of a BGR source uses it)
    RCV_TRY(rcv_ws_alloc(ctx, plane, &wix));
    RCV_TRY(rcv_ws_alloc(ctx, plane, &wiy));
    // streaming kernels: Sobel straight from the source (a BGR source: the gradient of its gray conversion, no gray image), then
    // ONE register-window kernel for the response of any block size and its 3x3 NMS (no response image unless the caller wants it)
    {
        const int rc2 = harris_fast_planes(ctx, s, resp ? &r : nullptr, &m, block, k, thr, wix, wiy);
        if (rc2 != RCV_ERR_UNSUPPORTED) return rc2;
    }
    if (!resp) {
        RCV_TRY(rcv_ws_alloc(ctx, rstep * s.rows * s.n, &wr));
        r = s;
        r.p = wr;
        r.step = rstep;
        r.fstride = rstep * s.rows;
        r.cap = r.fstride;
        r.ch = 1;
        r.esz = 4;
    }
    rcv_batch gb;
    gb.frame0.data = wg;
    gb.frame0.cap = (size_t)s.rows * s.cols;
    gb.frame0.step = (size_t)s.cols;
    gb.frame0.rows = s.rows;
    gb.frame0.cols = s.cols;
    gb.frame0.channels = 1;
    gb.frame0.depth = RCV_8U;
    gb.frame0.device = RCV_DEVICE;
    gb.frame0.reserved = 0;
    gb.frame_stride = (size_t)s.rows * s.cols;
    gb.n = s.n;
    gb.reserved = 0;
    View g;
    if (s.ch == 1) g = s;   // the source is the gray image
    else {
        RCV_TRY(rcv_cvt_color_batch(ctx, RCV_BGR2GRAY, bgr, &gb));
        RCV_TRY(rcv_view_batch(&gb, RCV_8U, &g));
    }
    RCV_TRY(harris_from_gray(ctx, g, r, block, k, wix, wiy));
    return nms_launch(ctx, r, m, thr);
}

// ---- single-Mat forms -----------------------------------------------------------------------------

extern "C" int rcv_corner_harris(rcv_ctx* ctx, const rcv_mat* gray, rcv_mat* resp, int block, float k)
{
    if (!gray || !resp) return RCV_ERR_ARG;
    Stage st;
    RCV_TRY(stage_begin(&st, ctx));
    rcv_mat *ds, *dd;
    RCV_TRY(stage_in(&st, gray, true, false, &ds));
    RCV_TRY(stage_in(&st, resp, true, true, &dd));
    rcv_batch bs = rcv_single(ds), bd = rcv_single(dd);
    return stage_finish(&st, rcv_corner_harris_batch(ctx, &bs, &bd, block, k));
}

extern "C" int rcv_nms3x3(rcv_ctx* ctx, const rcv_mat* resp, rcv_mat* mask, float thr)
{
    if (!resp || !mask) return RCV_ERR_ARG;
    Stage st;
    RCV_TRY(stage_begin(&st, ctx));
    rcv_mat *ds, *dd;
    RCV_TRY(stage_in(&st, resp, true, false, &ds));
    RCV_TRY(stage_in(&st, mask, true, true, &dd));
    rcv_batch bs = rcv_single(ds), bd = rcv_single(dd);
    return stage_finish(&st, rcv_nms3x3_batch(ctx, &bs, &bd, thr));
}

extern "C" int rcv_harris_pipeline(rcv_ctx* ctx, const rcv_mat* bgr, rcv_mat* mask, rcv_mat* resp, int block, float k, float thr)
{
    if (!bgr || !mask) return RCV_ERR_ARG;
    Stage st;
    RCV_TRY(stage_begin(&st, ctx));
    rcv_mat *ds, *dm, *dr = nullptr;
    RCV_TRY(stage_in(&st, bgr, true, false, &ds));
    RCV_TRY(stage_in(&st, mask, true, true, &dm));
    if (resp) RCV_TRY(stage_in(&st, resp, true, true, &dr));
    rcv_batch bs = rcv_single(ds), bm = rcv_single(dm), br;
    if (resp) br = rcv_single(dr);
    return stage_finish(&st, rcv_harris_pipeline_batch(ctx, &bs, &bm, resp ? &br : nullptr, block, k, thr));
}
