/*
 * rcv_oracle.h -- CPU oracle for the rustcv imgproc per-pixel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle.so, and only as the checker / the reported CPU baseline.
 * librustcv_hip.so never links or calls it.
 *
 * Two groups of functions:
 *
 *  (A) restatements of reference code (file:line relative to /root/reference):
 *        orc_yuyv_to_bgr   rustcv/src/videoio/mod.rs:344-382, twin rustcv-camera/src/decode.rs:160-191
 *        orc_bgra_to_bgr   rustcv/src/videoio/mod.rs:385-399, twin rustcv-camera/src/decode.rs:200-207
 *        orc_rgb_to_bgr    rustcv-camera/src/decode.rs:213-219
 *        orc_rectangle     rustcv/src/imgproc/drawing.rs:67-106
 *        orc_blend_glyph   rustcv/src/imgproc/drawing.rs:137-160 (put_text's per-pixel closure)
 *      PINNING: the reference (Rust, no rustc/cargo in this image) cannot be
 *      built or run here.  It holds three tests for this path
 *      (rustcv-camera/src/decode.rs:234-273): two inequality checks for
 *      yuyv_to_bgr and one exact vector for rgb_to_bgr -- all three are
 *      replayed in tests/test_oracle.py.  bgra_to_bgr, rectangle and the
 *      put_text blend have no reference test or fixture: for those three the
 *      oracle is a line-by-line restatement with PARITY UNPINNED.
 *
 *  (B) build-defined ops that do NOT exist in the reference (SURVEY.md F1,
 *      spec in SURVEY.md 8-A): bgr2gray, gaussian_blur, filter2d_i8/f32, sobel,
 *      resize, warp_affine, corner_harris, nms3x3, harris_pipeline, and the
 *      synthetic frame generator.  PARITY UNPINNED against the reference (there
 *      is nothing to pin to); cross-checked against scipy integer arithmetic in
 *      tests/test_oracle.py.
 *
 * All images are row-major, pixel-interleaved, with an explicit byte `step`.
 * f32 code is compiled with -ffp-contract=off; every fused multiply-add that the
 * spec wants is written as an explicit fmaf().
 */
#ifndef RCV_ORACLE_H
#define RCV_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- (A) reference restatements ------------------------------------------ */

/* returns 1 if the conversion ran, 0 if the reference's length guard made it a
 * silent no-op.  variant 0 = facade guard (src only), 1 = twin guard (src+dst) */
int orc_yuyv_to_bgr(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len,
                    size_t width, size_t height, int variant);
int orc_bgra_to_bgr(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len,
                    size_t width, size_t height, int variant);
void orc_rgb_to_bgr(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len);
void orc_rectangle(uint8_t* data, size_t data_len, int32_t rows, int32_t cols, size_t step,
                   int32_t x, int32_t y, int32_t w, int32_t h,
                   uint8_t b, uint8_t g, uint8_t r, int32_t thickness);

/* "next" rows f2 / f4 */
void orc_bgr_to_u32(const uint8_t* src, size_t src_len, uint32_t* dst, size_t pixel_count);      /* highgui/mod.rs:125-141 */
void orc_bgr_to_rgb_rows(const uint8_t* src, size_t sstep, uint8_t* dst, int rows, int cols);    /* imgcodecs/mod.rs:51-63 */
void orc_yuv422_to_bgr_strided(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int rows, int cols, int uyvy);
int orc_nv12_to_bgr(const uint8_t* src, size_t src_len, size_t sstep, uint8_t* dst, size_t dstep, int rows, int cols);
/* put_text's blend closure for one rasterised glyph box, drawing.rs:137-160 (no reference test: PARITY UNPINNED) */
void orc_blend_glyph(uint8_t* data, int32_t rows, int32_t cols, size_t step, int32_t min_x, int32_t min_y, int32_t w, int32_t h,
                     const float* cov, uint8_t cb, uint8_t cg, uint8_t cr);

/* ---- (B) build-defined ops -------------------------------- */

int orc_reflect101(int i, int n);

void orc_bgr2gray(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int rows, int cols);

/* ksize in {3,5,7} when sigma<=0 (integer taps); odd ksize<=31 when sigma>0 */
int orc_gaussian_blur(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep,
                      int rows, int cols, int ch, int ksize, double sigma);
/* host helper shared with nobody: f32 taps for sigma>0 (f64 normalise, cast once) */
void orc_gaussian_taps_f32(int ksize, double sigma, float* taps);

int orc_filter2d_i8(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep,
                    int rows, int cols, int ch, const int8_t* k, int ksize, int shift);
int orc_filter2d_f32(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep,
                     int rows, int cols, int ch, const float* k, int ksize, float delta);

void orc_sobel(const uint8_t* src, size_t sstep, int16_t* dx, size_t dxstep,
               int16_t* dy, size_t dystep, int rows, int cols);

void orc_resize(const uint8_t* src, size_t sstep, int srows, int scols,
                uint8_t* dst, size_t dstep, int drows, int dcols, int ch);

void orc_warp_affine(const uint8_t* src, size_t sstep, int srows, int scols,
                     uint8_t* dst, size_t dstep, int drows, int dcols, int ch, const float* M);

/* RCV_32F images: the same rules and f32 operation order, f32 taps, the unrounded interpolated value as the result */
void orc_resize_f32(const float* src, size_t sstep, int srows, int scols,
                    float* dst, size_t dstep, int drows, int dcols, int ch);
void orc_warp_affine_f32(const float* src, size_t sstep, int srows, int scols,
                         float* dst, size_t dstep, int drows, int dcols, int ch, const float* M);

int orc_corner_harris(const uint8_t* gray, size_t sstep, float* resp, size_t rstep,
                      int rows, int cols, int block, float k);
void orc_nms3x3(const float* resp, size_t rstep, uint8_t* mask, size_t mstep,
                int rows, int cols, float thr);
/* BGR -> gray -> sobel -> harris response -> nms mask; resp may be NULL */
int orc_harris_pipeline(const uint8_t* bgr, size_t sstep, uint8_t* mask, size_t mstep,
                        float* resp, size_t rstep, int rows, int cols, int block, float k, float thr);

/* ---- synthetic frames (replaces the empty rustcv-simulation stub) ---------- */
uint64_t orc_splitmix64(uint64_t z);
/* family 0 = noise, 1 = scene.  ch = 1,3,4 interleaved u8 */
void orc_synth_frame(uint8_t* dst, size_t step, int rows, int cols, int ch,
                     int family, uint64_t seed, uint64_t frame);
/* packed YUYV (2 B/px): Y from stream `seed`, U/V from stream seed^GOLDEN */
void orc_synth_yuyv(uint8_t* dst, size_t step, int rows, int cols, uint64_t seed, uint64_t frame);
/* the config-3 7x7 integer kernel */
void orc_bench_kernel7(int8_t* k49);

/* number of OpenMP threads the stencil loops will use (1 if built without) */
int orc_threads(void);
void orc_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
