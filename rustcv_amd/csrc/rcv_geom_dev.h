// rcv_geom_dev.h -- device helpers shared by the geometry kernels (rcv_geom.hip: resize / warpAffine; rcv_warp_resize.hip: the
// fused warpAffine -> down-scale launches): the byte-granular reference formulation of one warped pixel, the packed-f32 bilinear
// sample of a BGR pixel from its tap dwords, the XCD-aware tile order.  f32 evaluation order is fixed and spelled out op by op
// (SURVEY.md 8-A == oracle/rcv_oracle.c orc_resize / orc_warp_affine); built with -ffp-contract=off.
#pragma once
#include "rcv_internal.h"
#include <math.h>
#include <string.h>

namespace {

__device__ __forceinline__ uint8_t round_half_up_u8(float v)
{
    int iv = (int)floorf(v + 0.5f);
    return (uint8_t)min(max(iv, 0), 255);
}

// Two horizontally adjacent BGR taps (6 bytes at byte offset 3*x0 of a row) fetched with ONE 8-byte load instead of
// six byte loads.  The load window is clamped into the row (rows are >= 8 bytes here), then shifted into place;
// x0 may be -1 (left tap outside: its bytes are garbage and masked by the caller).
__device__ __forceinline__ uint64_t load_taps6(const uint8_t* row, int x0, int rowbytes)
{
    const int off = 3 * x0;
    const int offc = min(max(off, 0), rowbytes - 8);
    uint64_t raw;
    __builtin_memcpy(&raw, row + offc, 8);   // unaligned 8-byte global load
    const int sh = (off - offc) * 8;         // -24 .. +56 bits
    return sh >= 0 ? (raw >> sh) : (raw << (-sh));
}

struct Affine { float m[6]; };

// One output pixel, any channel count, byte-granular taps: the reference formulation every fast path below must match.
template <int CH>
__device__ __forceinline__ void warp_px(const uint8_t* sf, const View& s, const Affine& A, float fxx, float fyy, uint8_t* o)
{
    float sx = fmaf(A.m[0], fxx, fmaf(A.m[1], fyy, A.m[2]));
    float sy = fmaf(A.m[3], fxx, fmaf(A.m[4], fyy, A.m[5]));
    if (!(sx > -1.0f && sx < (float)s.cols && sy > -1.0f && sy < (float)s.rows)) {
#pragma unroll
        for (int c = 0; c < CH; ++c) o[c] = 0;
        return;
    }
    float x0f = floorf(sx), y0f = floorf(sy);
    int x0 = (int)x0f, y0 = (int)y0f;
    float fx = sx - x0f, fy = sy - y0f;
    int x1 = x0 + 1, y1 = y0 + 1;
    bool vx0 = x0 >= 0, vx1 = x1 < s.cols, vy0 = y0 >= 0, vy1 = y1 < s.rows;
    const uint8_t* ra = sf + (size_t)(vy0 ? y0 : 0) * s.step;
    const uint8_t* rb = sf + (size_t)(vy1 ? y1 : 0) * s.step;
    size_t xa = (size_t)(vx0 ? x0 : 0) * CH, xb = (size_t)(vx1 ? x1 : 0) * CH;
    uint64_t ta = 0, tb = 0;
    const bool wide = CH == 3 && s.cols >= 3;
    if (wide) {
        ta = load_taps6(ra, x0, s.cols * 3);
        tb = load_taps6(rb, x0, s.cols * 3);
    }
    // four channels on 4-byte aligned rows: a tap is one dword (taps outside the source read a clamped position and count as 0)
    const bool quad = CH == 4 && ((((uintptr_t)ra | (uintptr_t)rb) & 3) == 0);
    uint32_t q00 = 0, q01 = 0, q10 = 0, q11 = 0;
    if (quad) {
        q00 = (vx0 && vy0) ? *(const uint32_t*)(ra + xa) : 0u; q01 = (vx1 && vy0) ? *(const uint32_t*)(ra + xb) : 0u;
        q10 = (vx0 && vy1) ? *(const uint32_t*)(rb + xa) : 0u; q11 = (vx1 && vy1) ? *(const uint32_t*)(rb + xb) : 0u;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        float p00, p01, p10, p11;
        if (quad) {
            p00 = (float)((q00 >> (8 * c)) & 0xff); p01 = (float)((q01 >> (8 * c)) & 0xff);
            p10 = (float)((q10 >> (8 * c)) & 0xff); p11 = (float)((q11 >> (8 * c)) & 0xff);
        } else if (wide) {
            p00 = (vx0 && vy0) ? (float)(uint32_t)((ta >> (8 * c)) & 0xff) : 0.0f;
            p01 = (vx1 && vy0) ? (float)(uint32_t)((ta >> (24 + 8 * c)) & 0xff) : 0.0f;
            p10 = (vx0 && vy1) ? (float)(uint32_t)((tb >> (8 * c)) & 0xff) : 0.0f;
            p11 = (vx1 && vy1) ? (float)(uint32_t)((tb >> (24 + 8 * c)) & 0xff) : 0.0f;
        } else {
            p00 = (vx0 && vy0) ? (float)ra[xa + c] : 0.0f;
            p01 = (vx1 && vy0) ? (float)ra[xb + c] : 0.0f;
            p10 = (vx0 && vy1) ? (float)rb[xa + c] : 0.0f;
            p11 = (vx1 && vy1) ? (float)rb[xb + c] : 0.0f;
        }
        float top = fmaf(fx, p01 - p00, p00);
        float bot = fmaf(fx, p11 - p10, p10);
        float v = fmaf(fy, bot - top, top);
        o[c] = round_half_up_u8(v);
    }
}

typedef float f2 __attribute__((ext_vector_type(2)));

// byte N of a dword -> f32 in one instruction (the compiler otherwise mixes shifts, masks and integer subtracts in)
template <int N>
__device__ __forceinline__ float ub(uint32_t v)
{
    float f;
    if constexpr (N == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(v));
    else if constexpr (N == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(v));
    else if constexpr (N == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(v));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(v));
    return f;
}

// Bilinear sample of one BGR pixel from its two tap dwords per row {b0 g0 r0 b1 | g1 r1 . .} (upper row a, lower row b),
// packed {b, g, r, 0}.  The f32 operations and their order are those of warp_px<3> / resize_px<3>: per channel
// top = fma(fx, p01 - p00, p00), bot = fma(fx, p11 - p10, p10), v = fma(fy, bot - top, top), floor(v + 0.5).  Channels 0 and 1
// ride in packed-f32 pairs, channel 2 pairs its top and bottom row (v_pk_add / v_pk_fma: two IEEE operations per
// instruction, bit-identical to the scalar ops).  All taps must be valid (interior), so the result is an exact integer in
// [0, 255] and v_cvt_pk_u8_f32 converts, saturates and packs it.
// PIN = true pins each byte conversion to one v_cvt_f32_ubyteN (fewer instructions: -2 % in the 8-row warp / 4-row resize
// kernels); the fused down-scale kernel, which interleaves four pixels, schedules better with the compiler's own choice.
// d = fma({w, w}, a, b) with w = the low (HI = 0) or the high (HI = 1) half of the register pair wp: op_sel broadcasts the
// half, so the {fx, fy} pair the coordinate arithmetic leaves behind feeds all four lerps without a v_mov to duplicate it
template <int HI>
__device__ __forceinline__ f2 pk_fma_bc(f2 wp, f2 a, f2 b)
{
    f2 d;
    if constexpr (HI == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(wp), "v"(a), "v"(b));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(wp), "v"(a), "v"(b));
    return d;
}

// {floor(b), floor(g), floor(r), 0} of three values in [0, 256): v_cvt_u32_f32 truncates (= floor for these non-negative values)
// and its SDWA form writes the low byte of the result straight into byte 0 / 1 / 2 of the destination, the other bytes kept --
// three instructions for what v_floor_f32 + v_cvt_pk_u8_f32 need six (the interpolated value + 0.5 lies in [0.5, 255.5]: every
// lerp result is between its two end points, so neither the saturation nor the rounding of v_cvt_pk_u8_f32 is ever used)
__device__ __forceinline__ uint32_t pack_floor3(float b, float g, float r)
{
    uint32_t d;
    asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD" : "=v"(d) : "v"(b));
    asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(g));
    asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(r));
    return d;
}

// floor of a value in [0, 256) as an integer: one v_cvt_u32_f32 (truncation) instead of v_floor_f32 + v_cvt_i32_f32
__device__ __forceinline__ uint32_t trunc_u32(float v)
{
    uint32_t d;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(d) : "v"(v));
    return d;
}

template <bool PIN>
__device__ __forceinline__ uint32_t bilerp_bgr(uint32_t alo, uint32_t ahi, uint32_t blo, uint32_t bhi, f2 fxy)
{
    f2 a0, a1, b0, b1, c0, c1;
    if constexpr (PIN) {
        a0 = f2{ub<0>(alo), ub<1>(alo)}; a1 = f2{ub<3>(alo), ub<0>(ahi)};
        b0 = f2{ub<0>(blo), ub<1>(blo)}; b1 = f2{ub<3>(blo), ub<0>(bhi)};
        c0 = f2{ub<2>(alo), ub<2>(blo)}; c1 = f2{ub<1>(ahi), ub<1>(bhi)};
    } else {
        a0 = f2{(float)(alo & 0xff), (float)((alo >> 8) & 0xff)}; a1 = f2{(float)(alo >> 24), (float)(ahi & 0xff)};
        b0 = f2{(float)(blo & 0xff), (float)((blo >> 8) & 0xff)}; b1 = f2{(float)(blo >> 24), (float)(bhi & 0xff)};
        c0 = f2{(float)((alo >> 16) & 0xff), (float)((blo >> 16) & 0xff)}; c1 = f2{(float)((ahi >> 8) & 0xff), (float)((bhi >> 8) & 0xff)};
    }
    const f2 half2 = {0.5f, 0.5f};
    const f2 top = pk_fma_bc<0>(fxy, a1 - a0, a0);
    const f2 bot = pk_fma_bc<0>(fxy, b1 - b0, b0);
    const f2 tb2 = pk_fma_bc<0>(fxy, c1 - c0, c0);
    const f2 v01 = pk_fma_bc<1>(fxy, bot - top, top) + half2;
    const float v2 = fmaf(fxy.y, tb2.y - tb2.x, tb2.x) + 0.5f;
    return pack_floor3(v01.x, v01.y, v2);
}   // output rows per thread: fewer, longer-lived workgroups and 16 tap loads in flight per lane

// Tile order of the LDS-staged warp kernels: hardware places block b on XCD b % 8.  With tiles_per_xcd > 0 every XCD works through
// its own contiguous run of the tile list, so that tiles whose patches overlap (the bounding box of a rotated tile is ~1.5x the
// tile; a one-channel tile row is half a 128-byte line) run on the same L2 shortly after one another.  The list order is (frame
// group, strip, tile row, tile column inside the strip): vertical strips of `strip` tile columns walked row by row, so that the
// ~100 tiles an XCD has in flight form a block whose patches overlap on all four sides inside that L2; strip = 0: plain raster.
__device__ __forceinline__ bool wl_tile(int tiles_per_xcd, int strip, int gx, int gy, int ntiles, int& bx, int& by, int& bz)
{
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
    if (tiles_per_xcd <= 0) return true;
    const int t = (int)(blockIdx.x & 7) * tiles_per_xcd + (int)(blockIdx.x >> 3);
    if (t >= ntiles) return false;
    bz = t / (gx * gy);
    const int rem = t - bz * gx * gy;
    if (strip > 0) {   // (the last strip may be narrower)
        const int per = strip * gy, sidx = rem / per, r2 = rem - sidx * per, w = min(strip, gx - sidx * strip);
        by = r2 / w;
        bx = sidx * strip + r2 - by * w;
    } else {
        by = rem / gx;
        bx = rem - by * gx;
    }
    return true;
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

} // namespace
