// rcv_kernels.h -- internal dispatch hooks of the shape-specialised (tiled / MFMA) kernels.
// Each returns RCV_ERR_UNSUPPORTED when it does not take the shape; the caller then falls
// through to the generic HIP kernel for that op (never to a CPU path).
#pragma once
#include "rcv_internal.h"

int rcv_gauss_int_rows(rcv_ctx* ctx, const View& s, const View& d, int ksize);    // rcv_gauss_rows.hip: small launches, ksize 3 / 5
int rcv_gauss_int_tiled(rcv_ctx* ctx, const View& s, const View& d, int ksize);
int rcv_filter_i8_fast(rcv_ctx* ctx, const View& s, const View& d, const int8_t* k, int ksize, int shift);
int rcv_filter_i16_rows(rcv_ctx* ctx, const View& s, const View& d, const int16_t* k, int ksize, int shift, int src_yuyv = 0, bool any_size = false,
                        const View* gx = nullptr, const View* gy = nullptr);   // rcv_filter_rows_mfma.hip, |k| <= 511
int rcv_filter_i16_fast(rcv_ctx* ctx, const View& s, const View& d, const int16_t* k, int ksize, int shift, int src_yuyv);  // |k| <= 511
int rcv_filter_i8_yuyv_fast(rcv_ctx* ctx, const View& s, const View& d, const int8_t* k, int ksize, int shift);
int rcv_sobel_tiled(rcv_ctx* ctx, const View& s, const View& dx, const View& dy);
int rcv_filter_i16_gray(rcv_ctx* ctx, const View& s, const View& d, const int16_t* k, int ksize, int shift);   // rcv_filter_gray_dot4.hip
// src: BGR (3 ch), packed YUYV (2 ch) or gray (1 ch); mask may be null (response only: needs resp)
int rcv_harris_fused(rcv_ctx* ctx, const View& src, const View* mask, const View* resp, int block, float k, float thr);
// cornerHarris response for any block 1..7 from i16 Sobel planes (rcv_harris_blocks.hip)
size_t rcv_harris_plane_step(int cols);   // row step of the i16 planes the kernel reads (8 pixels of margin left, 16 right)
size_t rcv_harris_plane_margin();         // bytes in front of column 0
bool rcv_harris_resp_rows_ok(const View& out, int block, bool is_resp);
// r: f32 response (may be null), m: u8 NMS mask with threshold thr (may be null)
int rcv_harris_resp_rows(rcv_ctx* ctx, const View& ix, const View& iy, const View* r, const View* m, int block, float k, float thr);
// the same with gray conversion and Sobel in front of the window: one launch, aligned shapes, BGR or gray source
int rcv_harris_blocks_fused(rcv_ctx* ctx, const View& s, const View* r, const View* m, int block, float k, float thr);
int rcv_filter_f32_fast(rcv_ctx* ctx, const View& s, const View& d, const float* k, int ksize, float delta);
int rcv_gauss_f32_fast(rcv_ctx* ctx, const View& s, const View& d, const float* taps, int ksize);
// integer filters on the streaming f32 kernel (exact): shapes the strip kernel does not take
int rcv_filter_i16_stream(rcv_ctx* ctx, const View& s, const View& d, const int16_t* k, int ksize, int shift);
int rcv_gauss_int_stream(rcv_ctx* ctx, const View& s, const View& d, const int* taps, int ksize, int shift);
// generic kernels restricted to the byte columns [xb_lo, xb_hi) of every row (edge fix-up of the streaming kernels)
int rcv_filter_f32_generic_range(rcv_ctx* ctx, const View& s, const View& d, const float* k, int ksize, float delta, int xb_lo, int xb_hi);
int rcv_gauss_f32_generic_range(rcv_ctx* ctx, const View& s, const View& d, const float* taps, int ksize, int xb_lo, int xb_hi);
// the chained filter2D of 16+ BGR frames as two halves on the context's two streams (rcv_filter_rows_mfma.hip); RCV_ERR_UNSUPPORTED: not taken, nothing enqueued.
// Called INSTEAD of rcv_bind by the entry point.
int rcv_filter_i8_split(rcv_ctx* ctx, const View& s, const View& d, const int8_t* k, int ksize, int shift);
