/*
 * rcv_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see rcv_oracle.h).
 *
 * Build:  gcc -O2 -std=c11 -ffp-contract=off -fopenmp -shared -fPIC rcv_oracle.c -o liboracle.so -lm
 * Loops are written for obviousness, not speed; the OpenMP pragmas only split
 * independent output rows and never change any per-pixel result.
 */
#include "rcv_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;
int orc_threads(void) { return g_threads; }
void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n < 1) n = 1;
    g_threads = n;
    omp_set_num_threads(n);
#else
    (void)n;
    g_threads = 1;
#endif
}

/* ========================================================================== */
/* (A) reference restatements                                                 */
/* ========================================================================== */

/* rustcv/src/videoio/mod.rs:373-382 (facade) == rustcv-camera/src/decode.rs:226-228 (twin) */
static inline uint8_t clamp_u8(int32_t v) { return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v); }

/* rustcv/src/videoio/mod.rs:344-371.  The frame is a FLAT array of w*h/2
 * macropixels [Y0 U Y1 V]; row stride is ignored; an odd w*h leaves the last
 * pixel untouched.  `>>` on i32 is arithmetic in Rust; C's >> on negative int
 * is implementation-defined, gcc/clang make it arithmetic -- and every negative
 * result clamps to 0 anyway.
 * Guards: facade (:345-348) returns silently iff src is short; twin
 * (decode.rs:160-167) checks pairs*4 / pairs*6 on both buffers.               */
int orc_yuyv_to_bgr(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len,
                    size_t width, size_t height, int variant)
{
    size_t pairs = width * height / 2;
    if (variant == 0) {
        if (src_len < width * height * 2) return 0;
        if (dst_len < pairs * 6) return 0; /* reference would panic on the index; we refuse */
    } else {
        if (src_len < pairs * 4 || dst_len < pairs * 6) return 0;
    }
    for (size_t i = 0; i < pairs; ++i) {
        const uint8_t* s = src + i * 4;
        uint8_t* d = dst + i * 6;
        int32_t y0 = s[0], u = (int32_t)s[1] - 128, y1 = s[2], v = (int32_t)s[3] - 128;
        int32_t c0 = y0 - 16, c1 = y1 - 16;
        d[0] = clamp_u8((298 * c0 + 516 * u + 128) >> 8);
        d[1] = clamp_u8((298 * c0 - 100 * u - 208 * v + 128) >> 8);
        d[2] = clamp_u8((298 * c0 + 409 * v + 128) >> 8);
        d[3] = clamp_u8((298 * c1 + 516 * u + 128) >> 8);
        d[4] = clamp_u8((298 * c1 - 100 * u - 208 * v + 128) >> 8);
        d[5] = clamp_u8((298 * c1 + 409 * v + 128) >> 8);
    }
    return 1;
}

/* rustcv/src/videoio/mod.rs:385-399 (variant 0: both-length guard, then w*h
 * pixels) ; rustcv-camera/src/decode.rs:200-207 (variant 1: zip to the shorter
 * of src/4 and dst/3 whole chunks, width/height unused)                        */
int orc_bgra_to_bgr(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len,
                    size_t width, size_t height, int variant)
{
    size_t n;
    if (variant == 0) {
        n = width * height;
        if (src_len < n * 4 || dst_len < n * 3) return 0;
    } else {
        size_t a = src_len / 4, b = dst_len / 3;
        n = a < b ? a : b;
    }
    for (size_t i = 0; i < n; ++i) {
        dst[3 * i + 0] = src[4 * i + 0];
        dst[3 * i + 1] = src[4 * i + 1];
        dst[3 * i + 2] = src[4 * i + 2];
    }
    return 1;
}

/* rustcv-camera/src/decode.rs:213-219: chunks_exact(3) zipped -> min whole pixels */
void orc_rgb_to_bgr(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len)
{
    size_t a = src_len / 3, b = dst_len / 3, n = a < b ? a : b;
    for (size_t i = 0; i < n; ++i) {
        dst[3 * i + 0] = src[3 * i + 2];
        dst[3 * i + 1] = src[3 * i + 1];
        dst[3 * i + 2] = src[3 * i + 0];
    }
}

/* rustcv/src/imgproc/drawing.rs:80-87.  (r as usize)/(c as usize) of a negative
 * i32 wraps to a huge usize in Rust; the multiply would overflow-panic in debug
 * and wrap in release.  Negative r/c can only arise from thickness > rect size;
 * we model the release build: 64-bit wrapping arithmetic, then the len guard. */
static inline void set_pixel(uint8_t* data, size_t len, size_t step, int32_t r, int32_t c,
                             uint8_t b, uint8_t g, uint8_t rr)
{
    size_t idx = (size_t)(int64_t)r * step + (size_t)(int64_t)c * 3u;
    if (idx + 2 < len && idx + 2 >= 2) { /* second term: idx+2 did not wrap past 0 */
        data[idx] = b;
        data[idx + 1] = g;
        data[idx + 2] = rr;
    }
}

/* rustcv/src/imgproc/drawing.rs:67-106 */
void orc_rectangle(uint8_t* data, size_t data_len, int32_t rows, int32_t cols, size_t step,
                   int32_t x, int32_t y, int32_t w, int32_t h,
                   uint8_t b, uint8_t g, uint8_t r, int32_t thickness)
{
    int32_t x_min = x > 0 ? x : 0;
    int32_t y_min = y > 0 ? y : 0;
    /* rect.x + rect.width: i32 add (wraps in a Rust release build, panics in debug) */
    int32_t xe = (int32_t)((uint32_t)x + (uint32_t)w), ye = (int32_t)((uint32_t)y + (uint32_t)h);
    int32_t x_max = xe < cols ? xe : cols;
    int32_t y_max = ye < rows ? ye : rows;
    if (x_min >= x_max || y_min >= y_max) return;
    for (int32_t c = x_min; c < x_max; ++c)
        for (int32_t t = 0; t < thickness; ++t) {
            set_pixel(data, data_len, step, y_min + t, c, b, g, r);
            set_pixel(data, data_len, step, y_max - 1 - t, c, b, g, r);
        }
    for (int32_t rr = y_min; rr < y_max; ++rr)
        for (int32_t t = 0; t < thickness; ++t) {
            set_pixel(data, data_len, step, rr, x_min + t, b, g, r);
            set_pixel(data, data_len, step, rr, x_max - 1 - t, b, g, r);
        }
}

/* rustcv/src/highgui/mod.rs:125-141 mat_to_u32_buffer: chunks_exact(3) of the WHOLE data vector (step ignored)
 * zipped with a zero-initialised rows*cols u32 buffer; pixel = (r << 16) | (g << 8) | b                        */
void orc_bgr_to_u32(const uint8_t* src, size_t src_len, uint32_t* dst, size_t pixel_count)
{
    size_t n = src_len / 3 < pixel_count ? src_len / 3 : pixel_count;
    for (size_t i = 0; i < pixel_count; ++i) dst[i] = 0;
    for (size_t i = 0; i < n; ++i)
        dst[i] = ((uint32_t)src[3 * i + 2] << 16) | ((uint32_t)src[3 * i + 1] << 8) | (uint32_t)src[3 * i];
}

/* rustcv/src/imgcodecs/mod.rs:51-63 (imwrite): for each row, row_bytes(r) (honours step), B,G,R -> R,G,B, packed */
void orc_bgr_to_rgb_rows(const uint8_t* src, size_t sstep, uint8_t* dst, int rows, int cols)
{
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const uint8_t* p = src + (size_t)y * sstep + (size_t)x * 3;
            uint8_t* d = dst + ((size_t)y * cols + x) * 3;
            d[0] = p[2];
            d[1] = p[1];
            d[2] = p[0];
        }
}

/* "next" row f2: stride-aware capture formats.  Arithmetic = the reference's BT.601 integer formula
 * (videoio/mod.rs:356-363); NV12 chroma sampling as in rustcv-backend-msmf/examples/camera_view/convert.rs:46-86
 * (uv row = row/2, uv col = col/2, same stride for both planes).  uyvy = 0: [Y0 U Y1 V], 1: [U Y0 V Y1].
 * Whole macropixels only: an odd last column is left untouched.                                                   */
static inline void yuv_px(int y, int u, int v, uint8_t* d)
{
    int c = 298 * (y - 16) + 128;
    d[0] = clamp_u8((c + 516 * u) >> 8);
    d[1] = clamp_u8((c - 100 * u - 208 * v) >> 8);
    d[2] = clamp_u8((c + 409 * v) >> 8);
}
void orc_yuv422_to_bgr_strided(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int rows, int cols, int uyvy)
{
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x + 1 < cols; x += 2) {
            const uint8_t* p = src + (size_t)y * sstep + (size_t)x * 2;
            int y0 = uyvy ? p[1] : p[0], u = (uyvy ? p[0] : p[1]) - 128, y1 = uyvy ? p[3] : p[2], v = (uyvy ? p[2] : p[3]) - 128;
            yuv_px(y0, u, v, dst + (size_t)y * dstep + (size_t)x * 3);
            yuv_px(y1, u, v, dst + (size_t)y * dstep + (size_t)x * 3 + 3);
        }
}
int orc_nv12_to_bgr(const uint8_t* src, size_t src_len, size_t sstep, uint8_t* dst, size_t dstep, int rows, int cols)
{
    size_t ysz = sstep * (size_t)rows, uvsz = sstep * (size_t)((rows + 1) / 2);
    if (src_len < ysz + uvsz) return 0; /* convert.rs:56-58: silent return */
    const uint8_t* uv = src + ysz;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const uint8_t* q = uv + (size_t)(y / 2) * sstep + (size_t)(x / 2) * 2;
            yuv_px(src[(size_t)y * sstep + x], q[0] - 128, q[1] - 128, dst + (size_t)y * dstep + (size_t)x * 3);
        }
    return 1;
}

/* Rust `f32 as u8`: truncation toward zero, saturating, NaN -> 0 */
static inline uint8_t f32_as_u8(float v)
{
    if (!(v > 0.0f)) return 0; /* negatives, -0, NaN */
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

/* rustcv/src/imgproc/drawing.rs:137-160 -- the per-pixel half of put_text: the closure `glyph.draw` calls for EVERY
 * pixel (x, y, v) of one positioned glyph's pixel bounding box, v = coverage.  (Layout and rasterisation are rusttype's,
 * a third-party crate working on a font blob the checkout does not hold: out of scope)  Separate f32
 * multiply / subtract / add roundings, no fused operation (rustc does not contract), truncating store after EVERY
 * glyph -- so overlapping boxes compose in glyph order.  No len guard in the reference (it would panic): the caller
 * guarantees (rows-1)*step + cols*3 <= len. */
void orc_blend_glyph(uint8_t* data, int32_t rows, int32_t cols, size_t step, int32_t min_x, int32_t min_y, int32_t w, int32_t h,
                     const float* cov, uint8_t cb, uint8_t cg, uint8_t cr)
{
    for (int32_t y = 0; y < h; ++y)
        for (int32_t x = 0; x < w; ++x) {
            /* `x as i32 + bounding_box.min.x` (:140-141) */
            int32_t px = (int32_t)((uint32_t)x + (uint32_t)min_x), py = (int32_t)((uint32_t)y + (uint32_t)min_y);
            if (px >= 0 && px < cols && py >= 0 && py < rows) {
                size_t idx = (size_t)py * step + (size_t)px * 3u;
                float alpha = cov[(size_t)y * (size_t)w + (size_t)x];
                float b_old = (float)data[idx], g_old = (float)data[idx + 1], r_old = (float)data[idx + 2];
                float inv = 1.0f - alpha;
                float b_new = ((float)cb * alpha) + (b_old * inv);
                float g_new = ((float)cg * alpha) + (g_old * inv);
                float r_new = ((float)cr * alpha) + (r_old * inv);
                data[idx] = f32_as_u8(b_new);
                data[idx + 1] = f32_as_u8(g_new);
                data[idx + 2] = f32_as_u8(r_new);
            }
        }
}

/* ========================================================================== */
/* (B) build-defined ops, SURVEY.md 8-A                                       */
/* ========================================================================== */

/* BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba ; iterated for tiny n */
int orc_reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * n - 2 - i;
    }
    return i;
}

/* byte offset of reflected pixel (i - r) for i in [0, n + 2r): hoists the border map out of the tap loops */
static int* reflect_table(int n, int r, int ch)
{
    int* t = (int*)malloc((size_t)(n + 2 * r) * sizeof(int));
    if (t) for (int i = 0; i < n + 2 * r; ++i) t[i] = orc_reflect101(i - r, n) * ch;
    return t;
}

static inline uint8_t sat_u8_i(int32_t v) { return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v); }

/* g = (1868 B + 9617 G + 4899 R + 8192) >> 14 */
void orc_bgr2gray(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int rows, int cols)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        const uint8_t* s = src + (size_t)y * sstep;
        uint8_t* d = dst + (size_t)y * dstep;
        for (int x = 0; x < cols; ++x)
            d[x] = (uint8_t)((1868 * s[3 * x] + 9617 * s[3 * x + 1] + 4899 * s[3 * x + 2] + 8192) >> 14);
    }
}

void orc_gaussian_taps_f32(int ksize, double sigma, float* taps)
{
    double t[64], sum = 0.0;
    int r = ksize / 2;
    for (int i = 0; i < ksize; ++i) {
        double x = (double)(i - r);
        t[i] = exp(-(x * x) / (2.0 * sigma * sigma));
        sum += t[i];
    }
    for (int i = 0; i < ksize; ++i) taps[i] = (float)(t[i] / sum);
}

int orc_gaussian_blur(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep,
                      int rows, int cols, int ch, int ksize, double sigma)
{
    if (rows <= 0 || cols <= 0) return 0;
    int r = ksize / 2;
    if (sigma <= 0.0) {
        static const int t3[3] = {1, 2, 1}, t5[5] = {1, 4, 6, 4, 1}, t7[7] = {2, 7, 14, 18, 14, 7, 2};
        const int* t;
        int D;
        if (ksize == 3) { t = t3; D = 16; }
        else if (ksize == 5) { t = t5; D = 256; }
        else if (ksize == 7) { t = t7; D = 4096; }
        else return -2;
#pragma omp parallel for schedule(static)
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols; ++x)
                for (int c = 0; c < ch; ++c) {
                    int32_t acc = 0;
                    for (int ky = 0; ky < ksize; ++ky) {
                        const uint8_t* s = src + (size_t)orc_reflect101(y + ky - r, rows) * sstep;
                        for (int kx = 0; kx < ksize; ++kx)
                            acc += t[ky] * t[kx] * (int32_t)s[(size_t)orc_reflect101(x + kx - r, cols) * ch + c];
                    }
                    dst[(size_t)y * dstep + (size_t)x * ch + c] = (uint8_t)((acc + D / 2) / D);
                }
        return 0;
    }
    if (!(ksize & 1) || ksize < 3 || ksize > 31) return -2;
    float taps[32];
    orc_gaussian_taps_f32(ksize, sigma, taps);
    /* horizontal pass into f32, then vertical; each an fmaf chain in tap order from 0 */
    float* tmp = (float*)malloc((size_t)rows * cols * ch * sizeof(float));
    if (!tmp) return -5;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        const uint8_t* s = src + (size_t)y * sstep;
        for (int x = 0; x < cols; ++x)
            for (int c = 0; c < ch; ++c) {
                float acc = 0.0f;
                for (int kx = 0; kx < ksize; ++kx)
                    acc = fmaf(taps[kx], (float)s[(size_t)orc_reflect101(x + kx - r, cols) * ch + c], acc);
                tmp[((size_t)y * cols + x) * ch + c] = acc;
            }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
            for (int c = 0; c < ch; ++c) {
                float acc = 0.0f;
                for (int ky = 0; ky < ksize; ++ky)
                    acc = fmaf(taps[ky], tmp[((size_t)orc_reflect101(y + ky - r, rows) * cols + x) * ch + c], acc);
                float v = rintf(acc);
                dst[(size_t)y * dstep + (size_t)x * ch + c] = v < 0.0f ? 0 : (v > 255.0f ? 255 : (uint8_t)v);
            }
    free(tmp);
    return 0;
}

/* out = sat_u8((sum k*p + (1<<(shift-1))) >> shift), arithmetic shift; shift==0: no rounding term */
int orc_filter2d_i8(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep,
                    int rows, int cols, int ch, const int8_t* k, int ksize, int shift)
{
    if (!(ksize & 1) || ksize < 1 || ksize > 15 || shift < 0 || shift > 24) return -2;
    int r = ksize / 2;
    int32_t rnd = shift > 0 ? (1 << (shift - 1)) : 0;
    int* xm = reflect_table(cols, r, ch);
    if (!xm) return -5;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        const uint8_t* rp[15];
        for (int ky = 0; ky < ksize; ++ky) rp[ky] = src + (size_t)orc_reflect101(y + ky - r, rows) * sstep;
        for (int x = 0; x < cols; ++x)
            for (int c = 0; c < ch; ++c) {
                int32_t acc = 0;
                for (int ky = 0; ky < ksize; ++ky)
                    for (int kx = 0; kx < ksize; ++kx)
                        acc += (int32_t)k[ky * ksize + kx] * (int32_t)rp[ky][xm[x + kx] + c];
                int32_t v = acc + rnd;
                v = v >= 0 ? (v >> shift) : -((-v + ((1 << shift) - 1)) >> shift); /* floor division == arithmetic shift */
                dst[(size_t)y * dstep + (size_t)x * ch + c] = sat_u8_i(v);
            }
    }
    free(xm);
    return 0;
}

/* acc = delta; for ky, for kx: acc = fmaf(k, (float)p, acc); out = sat_u8(rintf(acc)) */
int orc_filter2d_f32(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep,
                     int rows, int cols, int ch, const float* k, int ksize, float delta)
{
    if (!(ksize & 1) || ksize < 1 || ksize > 15) return -2;
    int r = ksize / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
            for (int c = 0; c < ch; ++c) {
                float acc = delta;
                for (int ky = 0; ky < ksize; ++ky) {
                    const uint8_t* s = src + (size_t)orc_reflect101(y + ky - r, rows) * sstep;
                    for (int kx = 0; kx < ksize; ++kx)
                        acc = fmaf(k[ky * ksize + kx], (float)s[(size_t)orc_reflect101(x + kx - r, cols) * ch + c], acc);
                }
                float v = rintf(acc);
                dst[(size_t)y * dstep + (size_t)x * ch + c] = v < 0.0f ? 0 : (v > 255.0f ? 255 : (uint8_t)v);
            }
    return 0;
}

/* dx = [-1 0 1; -2 0 2; -1 0 1], dy = transpose */
void orc_sobel(const uint8_t* src, size_t sstep, int16_t* dx, size_t dxstep,
               int16_t* dy, size_t dystep, int rows, int cols)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        const uint8_t* r0 = src + (size_t)orc_reflect101(y - 1, rows) * sstep;
        const uint8_t* r1 = src + (size_t)y * sstep;
        const uint8_t* r2 = src + (size_t)orc_reflect101(y + 1, rows) * sstep;
        int16_t* ox = (int16_t*)((uint8_t*)dx + (size_t)y * dxstep);
        int16_t* oy = (int16_t*)((uint8_t*)dy + (size_t)y * dystep);
        for (int x = 0; x < cols; ++x) {
            int xl = orc_reflect101(x - 1, cols), xr = orc_reflect101(x + 1, cols);
            int gx = (r0[xr] - r0[xl]) + 2 * (r1[xr] - r1[xl]) + (r2[xr] - r2[xl]);
            int gy = (r2[xl] - r0[xl]) + 2 * (r2[x] - r0[x]) + (r2[xr] - r0[xr]);
            ox[x] = (int16_t)gx;
            oy[x] = (int16_t)gy;
        }
    }
}

static inline uint8_t round_half_up_u8(float v)
{
    int iv = (int)floorf(v + 0.5f);
    return sat_u8_i(iv);
}

/* bilinear, half-pixel centres, clamp to image; f32 op order fixed:
 *   scale = (float)in / (float)out            (one f32 division)
 *   s     = ((float)d + 0.5f) * scale - 0.5f  (separate mul, sub)
 *   s     = min(max(s, 0), in-1) ; i0 = (int)floorf(s) ; f = s - i0 ; i1 = min(i0+1, in-1)
 *   top = fmaf(fx, p01-p00, p00) ; bot = fmaf(fx, p11-p10, p10) ; v = fmaf(fy, bot-top, top)
 *   out = sat_u8((int)floorf(v + 0.5f))                                         */
void orc_resize(const uint8_t* src, size_t sstep, int srows, int scols,
                uint8_t* dst, size_t dstep, int drows, int dcols, int ch)
{
    float scx = (float)scols / (float)dcols, scy = (float)srows / (float)drows;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y) {
        float sy = ((float)y + 0.5f) * scy - 0.5f;
        if (sy < 0.0f) sy = 0.0f;
        if (sy > (float)(srows - 1)) sy = (float)(srows - 1);
        int y0 = (int)floorf(sy);
        float fy = sy - (float)y0;
        int y1 = y0 + 1 < srows ? y0 + 1 : srows - 1;
        const uint8_t* ra = src + (size_t)y0 * sstep;
        const uint8_t* rb = src + (size_t)y1 * sstep;
        for (int x = 0; x < dcols; ++x) {
            float sx = ((float)x + 0.5f) * scx - 0.5f;
            if (sx < 0.0f) sx = 0.0f;
            if (sx > (float)(scols - 1)) sx = (float)(scols - 1);
            int x0 = (int)floorf(sx);
            float fx = sx - (float)x0;
            int x1 = x0 + 1 < scols ? x0 + 1 : scols - 1;
            for (int c = 0; c < ch; ++c) {
                float p00 = ra[(size_t)x0 * ch + c], p01 = ra[(size_t)x1 * ch + c];
                float p10 = rb[(size_t)x0 * ch + c], p11 = rb[(size_t)x1 * ch + c];
                float top = fmaf(fx, p01 - p00, p00);
                float bot = fmaf(fx, p11 - p10, p10);
                float v = fmaf(fy, bot - top, top);
                dst[(size_t)y * dstep + (size_t)x * ch + c] = round_half_up_u8(v);
            }
        }
    }
}

/* M maps dst -> src; constant border 0; each of the four taps is 0 when outside */
void orc_warp_affine(const uint8_t* src, size_t sstep, int srows, int scols,
                     uint8_t* dst, size_t dstep, int drows, int dcols, int ch, const float* M)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            float fxx = (float)x, fyy = (float)y;
            float sx = fmaf(M[0], fxx, fmaf(M[1], fyy, M[2]));
            float sy = fmaf(M[3], fxx, fmaf(M[4], fyy, M[5]));
            uint8_t* d = dst + (size_t)y * dstep + (size_t)x * ch;
            if (!(sx > -1.0f && sx < (float)scols && sy > -1.0f && sy < (float)srows)) {
                for (int c = 0; c < ch; ++c) d[c] = 0;
                continue;
            }
            float x0f = floorf(sx), y0f = floorf(sy);
            int x0 = (int)x0f, y0 = (int)y0f;
            float fx = sx - x0f, fy = sy - y0f;
            int x1 = x0 + 1, y1 = y0 + 1;
            int vx0 = x0 >= 0, vx1 = x1 < scols, vy0 = y0 >= 0, vy1 = y1 < srows;
            for (int c = 0; c < ch; ++c) {
                float p00 = (vx0 && vy0) ? (float)src[(size_t)y0 * sstep + (size_t)x0 * ch + c] : 0.0f;
                float p01 = (vx1 && vy0) ? (float)src[(size_t)y0 * sstep + (size_t)x1 * ch + c] : 0.0f;
                float p10 = (vx0 && vy1) ? (float)src[(size_t)y1 * sstep + (size_t)x0 * ch + c] : 0.0f;
                float p11 = (vx1 && vy1) ? (float)src[(size_t)y1 * sstep + (size_t)x1 * ch + c] : 0.0f;
                float top = fmaf(fx, p01 - p00, p00);
                float bot = fmaf(fx, p11 - p10, p10);
                float v = fmaf(fy, bot - top, top);
                d[c] = round_half_up_u8(v);
            }
        }
}

/* RCV_32F images ("warp_affine (u8/f32 ...)"; the Harris response map): the same sampling rules and the same
 * f32 operations in the same order as the u8 functions above -- top = fmaf(fx, p01 - p00, p00), bot likewise, v = fmaf(fy, bot -
 * top, top) -- with f32 taps and the UNROUNDED v as the result.  Steps in bytes.  north_star's 1-ULP clause applies to these. */
void orc_resize_f32(const float* src, size_t sstep, int srows, int scols,
                    float* dst, size_t dstep, int drows, int dcols, int ch)
{
    float scx = (float)scols / (float)dcols, scy = (float)srows / (float)drows;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y) {
        float sy = ((float)y + 0.5f) * scy - 0.5f;
        if (sy < 0.0f) sy = 0.0f;
        if (sy > (float)(srows - 1)) sy = (float)(srows - 1);
        int y0 = (int)floorf(sy);
        float fy = sy - (float)y0;
        int y1 = y0 + 1 < srows ? y0 + 1 : srows - 1;
        const float* ra = (const float*)((const uint8_t*)src + (size_t)y0 * sstep);
        const float* rb = (const float*)((const uint8_t*)src + (size_t)y1 * sstep);
        float* out = (float*)((uint8_t*)dst + (size_t)y * dstep);
        for (int x = 0; x < dcols; ++x) {
            float sx = ((float)x + 0.5f) * scx - 0.5f;
            if (sx < 0.0f) sx = 0.0f;
            if (sx > (float)(scols - 1)) sx = (float)(scols - 1);
            int x0 = (int)floorf(sx);
            float fx = sx - (float)x0;
            int x1 = x0 + 1 < scols ? x0 + 1 : scols - 1;
            for (int c = 0; c < ch; ++c) {
                float p00 = ra[(size_t)x0 * ch + c], p01 = ra[(size_t)x1 * ch + c];
                float p10 = rb[(size_t)x0 * ch + c], p11 = rb[(size_t)x1 * ch + c];
                float top = fmaf(fx, p01 - p00, p00);
                float bot = fmaf(fx, p11 - p10, p10);
                out[(size_t)x * ch + c] = fmaf(fy, bot - top, top);
            }
        }
    }
}

void orc_warp_affine_f32(const float* src, size_t sstep, int srows, int scols,
                         float* dst, size_t dstep, int drows, int dcols, int ch, const float* M)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            float fxx = (float)x, fyy = (float)y;
            float sx = fmaf(M[0], fxx, fmaf(M[1], fyy, M[2]));
            float sy = fmaf(M[3], fxx, fmaf(M[4], fyy, M[5]));
            float* d = (float*)((uint8_t*)dst + (size_t)y * dstep) + (size_t)x * ch;
            if (!(sx > -1.0f && sx < (float)scols && sy > -1.0f && sy < (float)srows)) {
                for (int c = 0; c < ch; ++c) d[c] = 0.0f;
                continue;
            }
            float x0f = floorf(sx), y0f = floorf(sy);
            int x0 = (int)x0f, y0 = (int)y0f;
            float fx = sx - x0f, fy = sy - y0f;
            int x1 = x0 + 1, y1 = y0 + 1;
            int vx0 = x0 >= 0, vx1 = x1 < scols, vy0 = y0 >= 0, vy1 = y1 < srows;
            const float* ra = (const float*)((const uint8_t*)src + (size_t)(vy0 ? y0 : 0) * sstep);
            const float* rb = (const float*)((const uint8_t*)src + (size_t)(vy1 ? y1 : 0) * sstep);
            for (int c = 0; c < ch; ++c) {
                float p00 = (vx0 && vy0) ? ra[(size_t)x0 * ch + c] : 0.0f;
                float p01 = (vx1 && vy0) ? ra[(size_t)x1 * ch + c] : 0.0f;
                float p10 = (vx0 && vy1) ? rb[(size_t)x0 * ch + c] : 0.0f;
                float p11 = (vx1 && vy1) ? rb[(size_t)x1 * ch + c] : 0.0f;
                float top = fmaf(fx, p01 - p00, p00);
                float bot = fmaf(fx, p11 - p10, p10);
                d[c] = fmaf(fy, bot - top, top);
            }
        }
}

/* Harris response from integer Sobel; box sums exact i32; six separate f32 ops */
static void harris_from_sobel(const int16_t* ix, const int16_t* iy, float* resp, size_t rstep,
                              int rows, int cols, int block, float k)
{
    double s = 1.0 / (4.0 * (double)block * 255.0); /* 2^(aperture-1) with aperture=3 */
    float s2 = (float)(s * s);
    int a = block / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        float* out = (float*)((uint8_t*)resp + (size_t)y * rstep);
        for (int x = 0; x < cols; ++x) {
            int32_t sxx = 0, sxy = 0, syy = 0;
            for (int by = 0; by < block; ++by) {
                int yy = orc_reflect101(y + by - a, rows);
                for (int bx = 0; bx < block; ++bx) {
                    int xx = orc_reflect101(x + bx - a, cols);
                    int32_t gx = ix[(size_t)yy * cols + xx], gy = iy[(size_t)yy * cols + xx];
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
}

int orc_corner_harris(const uint8_t* gray, size_t sstep, float* resp, size_t rstep,
                      int rows, int cols, int block, float k)
{
    if (block < 1 || block > 7 || rows <= 0 || cols <= 0) return -2;
    int16_t* ix = (int16_t*)malloc((size_t)rows * cols * 2);
    int16_t* iy = (int16_t*)malloc((size_t)rows * cols * 2);
    if (!ix || !iy) { free(ix); free(iy); return -5; }
    orc_sobel(gray, sstep, ix, (size_t)cols * 2, iy, (size_t)cols * 2, rows, cols);
    harris_from_sobel(ix, iy, resp, rstep, rows, cols, block, k);
    free(ix);
    free(iy);
    return 0;
}

/* mask = 255 iff r > thr && r >= all 8 neighbours (outside image = -inf) */
void orc_nms3x3(const float* resp, size_t rstep, uint8_t* mask, size_t mstep,
                int rows, int cols, float thr)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float r = *(const float*)((const uint8_t*)resp + (size_t)y * rstep + (size_t)x * 4);
            int keep = r > thr;
            for (int dy = -1; dy <= 1 && keep; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int yy = y + dy, xx = x + dx;
                    if ((dx == 0 && dy == 0) || yy < 0 || yy >= rows || xx < 0 || xx >= cols) continue;
                    float n = *(const float*)((const uint8_t*)resp + (size_t)yy * rstep + (size_t)xx * 4);
                    if (!(r >= n)) { keep = 0; break; }
                }
            mask[(size_t)y * mstep + x] = keep ? 255 : 0;
        }
}

int orc_harris_pipeline(const uint8_t* bgr, size_t sstep, uint8_t* mask, size_t mstep,
                        float* resp, size_t rstep, int rows, int cols, int block, float k, float thr)
{
    if (rows <= 0 || cols <= 0) return -2;
    uint8_t* gray = (uint8_t*)malloc((size_t)rows * cols);
    float* r = resp;
    size_t rs = rstep;
    if (!r) { r = (float*)malloc((size_t)rows * cols * 4); rs = (size_t)cols * 4; }
    if (!gray || !r) { free(gray); if (!resp) free(r); return -5; }
    orc_bgr2gray(bgr, sstep, gray, (size_t)cols, rows, cols);
    int rc = orc_corner_harris(gray, (size_t)cols, r, rs, rows, cols, block, k);
    if (rc == 0) orc_nms3x3(r, rs, mask, mstep, rows, cols, thr);
    free(gray);
    if (!resp) free(r);
    return rc;
}

/* ========================================================================== */
/* synthetic frames                                          */
/* ========================================================================== */

uint64_t orc_splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline uint8_t synth_noise(uint64_t seed, uint64_t frame, int y, int x, int c)
{
    uint64_t ctr = (frame << 40) + ((uint64_t)y << 20) + ((uint64_t)x << 2) + (uint64_t)c;
    return (uint8_t)(orc_splitmix64(seed ^ ctr) >> 56);
}

static inline uint8_t synth_sample(int family, uint64_t seed, uint64_t frame, int rows, int cols,
                                   int y, int x, int c)
{
    uint8_t n = synth_noise(seed, frame, y, x, c);
    if (family == 0) return n;
    int q = n >> 2;
    int ramp = (int)(((int64_t)x * 96) / cols);
    int chk = (((x >> 6) + (y >> 6)) & 1) ? 64 : 0;
    int mx = cols - 200 > 1 ? cols - 200 : 1, my = rows - 200 > 1 ? rows - 200 : 1;
    int sx0 = (int)((frame * 37u) % (uint64_t)mx), sy0 = (int)((frame * 23u) % (uint64_t)my);
    if (x >= sx0 && x < sx0 + 200 && y >= sy0 && y < sy0 + 200) return (uint8_t)(255 - q);
    return (uint8_t)(q + ramp + chk);
}

void orc_synth_frame(uint8_t* dst, size_t step, int rows, int cols, int ch,
                     int family, uint64_t seed, uint64_t frame)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
            for (int c = 0; c < ch; ++c)
                dst[(size_t)y * step + (size_t)x * ch + c] = synth_sample(family, seed, frame, rows, cols, y, x, c);
}

void orc_synth_yuyv(uint8_t* dst, size_t step, int rows, int cols, uint64_t seed, uint64_t frame)
{
    uint64_t seed2 = seed ^ 0x9E3779B97F4A7C15ull;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            uint8_t* d = dst + (size_t)y * step + (size_t)x * 2;
            d[0] = synth_noise(seed, frame, y, x, 0);
            d[1] = synth_noise(seed2, frame, y, x >> 1, (x & 1) ? 3 : 1);
        }
}

void orc_bench_kernel7(int8_t* k49)
{
    for (int i = 0; i < 49; ++i)
        k49[i] = (int8_t)((int)((orc_splitmix64(0xF117E2Dull ^ (uint64_t)i) >> 40) % 17u) - 8);
}
