// rcv_geom.hip -- bilinear resize and warpAffine (u8 and f32, channels 1/3/4).  Not in the reference
// (SURVEY.md F1); semantics SURVEY.md 8-A == oracle/rcv_oracle.c orc_resize / orc_warp_affine.
// f32 evaluation order is fixed and spelled out op by op; built with -ffp-contract=off.
#include "rcv_geom_dev.h"

namespace {

constexpr int kBlock = 256;

// Exact integer down-scale by S in {2,4} (both axes), 3 channels: with half-pixel centres the bilinear sample point
// falls exactly between the centre 2x2 pixels of each SxS block with weights 1/2, and the general f32 path
// (round-half-up) reduces to (a+b+c+d+2)>>2 -- verified against the general oracle in tests.  One thread makes 4
// output pixels from two 12*S-byte source runs (16-byte vector loads) and stores 12 bytes.
template <int S>
__global__ __launch_bounds__(kBlock) void k_resize_box(View s, View d)
{
    const int y = blockIdx.y;
    const int t = blockIdx.x * kBlock + threadIdx.x;      // group of 4 output pixels
    if (4 * t >= d.cols) return;
    const uint8_t* sf = s.p + (size_t)blockIdx.z * s.fstride;
    const uint8_t* ra = sf + (size_t)(S * y + S / 2 - 1) * s.step + (size_t)t * 12 * S;
    const uint8_t* rb = ra + s.step;
    uint32_t a[3 * S], b[3 * S];
    if constexpr (S == 4) {   // 48-byte runs, 16-byte aligned
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint4 va = ((const uint4*)ra)[i], vb = ((const uint4*)rb)[i];
            a[4 * i] = va.x; a[4 * i + 1] = va.y; a[4 * i + 2] = va.z; a[4 * i + 3] = va.w;
            b[4 * i] = vb.x; b[4 * i + 1] = vb.y; b[4 * i + 2] = vb.z; b[4 * i + 3] = vb.w;
        }
    } else {                  // 24-byte runs, 8-byte aligned
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint2 va = ((const uint2*)ra)[i], vb = ((const uint2*)rb)[i];
            a[2 * i] = va.x; a[2 * i + 1] = va.y;
            b[2 * i] = vb.x; b[2 * i + 1] = vb.y;
        }
    }
    uint32_t o[3] = {0, 0, 0};
#pragma unroll
    for (int px = 0; px < 4; ++px)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int k0 = 3 * (S * px + S / 2 - 1) + c, k1 = k0 + 3;   // byte index of the two centre pixels in the run
            const uint32_t s4 = ((a[k0 >> 2] >> ((k0 & 3) * 8)) & 0xff) + ((a[k1 >> 2] >> ((k1 & 3) * 8)) & 0xff) +
                                ((b[k0 >> 2] >> ((k0 & 3) * 8)) & 0xff) + ((b[k1 >> 2] >> ((k1 & 3) * 8)) & 0xff);
            const int ob = 3 * px + c;
            o[ob >> 2] |= ((s4 + 2) >> 2) << ((ob & 3) * 8);
        }
    struct U3 { uint32_t a, b, c; };
    *(U3*)(d.p + (size_t)blockIdx.z * d.fstride + (size_t)y * d.step + (size_t)t * 12) = U3{o[0], o[1], o[2]};
}

// one thread per output pixel; rows on blockIdx.y, frames on blockIdx.z
// One output pixel of the bilinear resize, any channel count: the formulation every fast path must match.
template <int CH>
__device__ __forceinline__ void resize_px(const uint8_t* ra, const uint8_t* rb, const View& s, float scx, float fy, int x, uint8_t* o)
{
    float sx = ((float)x + 0.5f) * scx - 0.5f;
    sx = sx < 0.0f ? 0.0f : sx;
    sx = sx > (float)(s.cols - 1) ? (float)(s.cols - 1) : sx;
    int x0 = (int)floorf(sx);
    float fx = sx - (float)x0;
    int x1 = x0 + 1 < s.cols ? x0 + 1 : s.cols - 1;
    uint64_t ta = 0, tb = 0;
    const bool wide = CH == 3 && s.cols >= 3;
    if (wide) {
        ta = load_taps6(ra, x0, s.cols * 3);
        tb = load_taps6(rb, x0, s.cols * 3);
        if (x1 == x0) {  // right edge: the second tap is the first one again
            ta = (ta & 0xffffffull) | ((ta & 0xffffffull) << 24);
            tb = (tb & 0xffffffull) | ((tb & 0xffffffull) << 24);
        }
    }
    // four channels on 4-byte aligned rows: a tap is one dword (16 byte loads -> 4 dword loads per pixel)
    const bool quad = CH == 4 && ((((uintptr_t)ra | (uintptr_t)rb) & 3) == 0);
    uint32_t q00 = 0, q01 = 0, q10 = 0, q11 = 0;
    if (quad) {
        q00 = ((const uint32_t*)ra)[x0]; q01 = ((const uint32_t*)ra)[x1];
        q10 = ((const uint32_t*)rb)[x0]; q11 = ((const uint32_t*)rb)[x1];
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        float p00, p01, p10, p11;
        if (quad) {
            p00 = (float)((q00 >> (8 * c)) & 0xff); p01 = (float)((q01 >> (8 * c)) & 0xff);
            p10 = (float)((q10 >> (8 * c)) & 0xff); p11 = (float)((q11 >> (8 * c)) & 0xff);
        } else if (wide) {
            p00 = (float)(uint32_t)((ta >> (8 * c)) & 0xff);
            p01 = (float)(uint32_t)((ta >> (24 + 8 * c)) & 0xff);
            p10 = (float)(uint32_t)((tb >> (8 * c)) & 0xff);
            p11 = (float)(uint32_t)((tb >> (24 + 8 * c)) & 0xff);
        } else {
            p00 = (float)ra[(size_t)x0 * CH + c], p01 = (float)ra[(size_t)x1 * CH + c];
            p10 = (float)rb[(size_t)x0 * CH + c], p11 = (float)rb[(size_t)x1 * CH + c];
        }
        float top = fmaf(fx, p01 - p00, p00);
        float bot = fmaf(fx, p11 - p10, p10);
        float v = fmaf(fy, bot - top, top);
        o[c] = round_half_up_u8(v);
    }
}

// source rows and vertical weight of output row y (uniform per row)
__device__ __forceinline__ void resize_row(const View& s, float scy, int y, int& y0, int& y1, float& fy)
{
    float sy = ((float)y + 0.5f) * scy - 0.5f;
    sy = sy < 0.0f ? 0.0f : sy;
    sy = sy > (float)(s.rows - 1) ? (float)(s.rows - 1) : sy;
    y0 = (int)floorf(sy);
    fy = sy - (float)y0;
    y1 = y0 + 1 < s.rows ? y0 + 1 : s.rows - 1;
}

template <int CH>
__global__ __launch_bounds__(kBlock) void k_resize(View s, View d, float scx, float scy)
{
    int y = blockIdx.y;
    const uint8_t* sf = s.p + (size_t)blockIdx.z * s.fstride;
    uint8_t* drow = d.p + (size_t)blockIdx.z * d.fstride + (size_t)y * d.step;
    int y0, y1;
    float fy;
    resize_row(s, scy, y, y0, y1, fy);
    const uint8_t* ra = sf + (size_t)y0 * s.step;
    const uint8_t* rb = sf + (size_t)y1 * s.step;
    for (int x = blockIdx.x * kBlock + threadIdx.x; x < d.cols; x += gridDim.x * kBlock) {
        if (CH == 4 && ((uintptr_t)drow & 3) == 0) {   // the pixel as one dword store
            uint8_t o[4];
            resize_px<CH>(ra, rb, s, scx, fy, x, o);
            ((uint32_t*)drow)[x] = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3 % CH] << 24);
        } else resize_px<CH>(ra, rb, s, scx, fy, x, drow + (size_t)x * CH);
    }
}

template <int CH>
__global__ __launch_bounds__(kBlock) void k_warp_affine(View s, View d, Affine A)
{
    int y = blockIdx.y;
    const uint8_t* sf = s.p + (size_t)blockIdx.z * s.fstride;
    uint8_t* drow = d.p + (size_t)blockIdx.z * d.fstride + (size_t)y * d.step;
    float fyy = (float)y;
    for (int x = blockIdx.x * kBlock + threadIdx.x; x < d.cols; x += gridDim.x * kBlock) {
        if (CH == 4 && ((uintptr_t)drow & 3) == 0) {   // the pixel as one dword store
            uint8_t o[4];
            warp_px<CH>(sf, s, A, (float)x, fyy, o);
            ((uint32_t*)drow)[x] = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3 % CH] << 24);
        } else warp_px<CH>(sf, s, A, (float)x, fyy, drow + (size_t)x * CH);
    }
}

// BGR fast path: one thread per output pixel (adjacent lanes -> adjacent source taps -> L1-friendly), byte -> float by
// v_cvt_f32_ubyteN straight from the tap dwords, invalid taps zeroed once per pixel instead of once per channel, and
// the three result bytes of four neighbouring lanes gathered with DPP so that every fourth lane stores 12 bytes.
// Same f32 operations in the same order as k_warp_affine<3>.
constexpr int kWarpRows = 8;
#ifndef RCV_WARP_WW
#define RCV_WARP_WW 64
#endif
#ifndef RCV_WARP_TW
#define RCV_WARP_TW 256
#endif
constexpr int kWarpWW = RCV_WARP_WW, kWarpTW = RCV_WARP_TW;   // wave / workgroup width in pixels

// 4 x 4 transpose of dwords inside every quad of lanes: on return lane 4q+i holds in a[j] what lane 4q+j held in a[i].
// Two butterfly stages (lane distance 1, then 2), each a select between a register and a quad-permuted neighbour register:
// 8 VALU instructions.  The kernels below use it to turn "row r of pixels, one per lane" into "4 pixels of one row per
// lane", so that ONE full-wave 12-byte store covers four output rows instead of four quarter-wave stores.
__device__ __forceinline__ void quad_transpose4(uint32_t (&a)[4], int lane)
{
    const bool odd = lane & 1, hi = lane & 2;
    const uint32_t x1 = __builtin_amdgcn_update_dpp(0u, a[1], 0xB1, 0xf, 0xf, true), x0 = __builtin_amdgcn_update_dpp(0u, a[0], 0xB1, 0xf, 0xf, true);
    const uint32_t x3 = __builtin_amdgcn_update_dpp(0u, a[3], 0xB1, 0xf, 0xf, true), x2 = __builtin_amdgcn_update_dpp(0u, a[2], 0xB1, 0xf, 0xf, true);
    const uint32_t b0 = odd ? x1 : a[0], b1 = odd ? a[1] : x0, b2 = odd ? x3 : a[2], b3 = odd ? a[3] : x2;
    const uint32_t y2 = __builtin_amdgcn_update_dpp(0u, b2, 0x4E, 0xf, 0xf, true), y0 = __builtin_amdgcn_update_dpp(0u, b0, 0x4E, 0xf, 0xf, true);
    const uint32_t y3 = __builtin_amdgcn_update_dpp(0u, b3, 0x4E, 0xf, 0xf, true), y1 = __builtin_amdgcn_update_dpp(0u, b1, 0x4E, 0xf, 0xf, true);
    a[0] = hi ? y2 : b0;
    a[2] = hi ? b2 : y0;
    a[1] = hi ? y3 : b1;
    a[3] = hi ? b3 : y1;
}

// The 12 bytes {a, b, c} of a quad of BGR pixels to a RAGGED destination: rows that are not 4-byte aligned (an odd width of a
// packed image: step = cols * 3) and the row's last quad, which holds npx < 4 pixels when the width is not a multiple of 4.
__device__ __forceinline__ void store_quad_ragged(uint8_t* q, uint32_t a, uint32_t b, uint32_t c, int npx)
{
    if (npx >= 4) {
        typedef uint32_t u3m __attribute__((ext_vector_type(3), aligned(1)));
        *(u3m*)q = u3m{a, b, c};
    } else {
        const uint32_t w[3] = {a, b, c};
#pragma unroll
        for (int i = 0; i < 9; ++i)
            if (i < 3 * npx) q[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
    }
}

// One wave's share of k_warp_affine_bgr / of the tiles k_warp_affine_lds<3> cannot stage: column x, rows ybase .. ybase + 7 of
// frames f0 .. f1 - 1.  The map is the same for every frame of a batch, so the source coordinates, tap offsets and lerp weights
// of the thread's 8 pixels -- a quarter of the interior path's arithmetic -- are computed once and reused for every frame.
// The tap loads are unconditional (clamped tap windows): a branch around them would make the compiler wait vmcnt(0) row by row.
__device__ __forceinline__ void warp_bgr_wave(const View& s, const View& d, const Affine& A, int f0, int f1, int x, int ybase)
{
    const int rowbytes = s.cols * 3;
    const float fxx = (float)min(x, d.cols - 1);
    const bool ragd = (d.cols & 3) || ((uintptr_t)d.p & 3) || (d.step & 3) || (d.fstride & 3);   // ragged destination (uniform)

    // ---- interior fast path (wave-uniform) ----
    // The border version below spends ~160 VALU ops per pixel and two unaligned 8-byte loads per row.  sx and sy are
    // monotonic in the row index for a fixed lane (fmaf rounds monotonically), so testing the first and the last row of the
    // thread bounds all eight.  If every lane of the wave keeps all four taps and the whole 12-byte aligned tap window
    // inside the source (0 <= sx < cols-3, 0 <= sy < rows-1) the validity masks, the clamps, the funnel shifts and the final
    // saturation are all no-ops: 54 VALU ops per pixel for the first frame of a group and 41 for the others, same f32
    // operations in the same order.
    {
        const float fy0 = (float)min(ybase, d.rows - 1), fy1 = (float)min(ybase + kWarpRows - 1, d.rows - 1);
        const float xa = fmaf(A.m[0], fxx, fmaf(A.m[1], fy0, A.m[2])), xb = fmaf(A.m[0], fxx, fmaf(A.m[1], fy1, A.m[2]));
        const float ya = fmaf(A.m[3], fxx, fmaf(A.m[4], fy0, A.m[5])), yb = fmaf(A.m[3], fxx, fmaf(A.m[4], fy1, A.m[5]));
        const float xl = (float)(s.cols - 3), yl = (float)(s.rows - 1);   // x0 <= cols-4: the 12-byte window below ends inside the row
        const bool inter = fminf(xa, xb) >= 0.0f && fmaxf(xa, xb) < xl && fminf(ya, yb) >= 0.0f && fmaxf(ya, yb) < yl;   // NaN -> false
        // (every frame of the batch 4-byte aligned: base and frame stride)
        const bool small = ((uintptr_t)s.p & 3) == 0 && (s.fstride & 3) == 0 && (s.step & 3) == 0 && s.step < (1u << 24) && s.rows < (1 << 24) &&
                           (unsigned long long)s.rows * s.step < (1ull << 32) && d.step < (1u << 24) && d.rows < (1 << 24) &&
                           (unsigned long long)d.rows * d.step < (1ull << 32);
        // all eight rows of every lane outside the source on the same side: constant border, nothing to load
        const bool outside = !(fmaxf(xa, xb) > -1.0f) || !(fminf(xa, xb) < (float)s.cols) || !(fmaxf(ya, yb) > -1.0f) || !(fminf(ya, yb) < (float)s.rows);
        if (__all(outside)) {
            if ((threadIdx.x & 3) == 0 && x < d.cols) {
                struct U3 { uint32_t a, b, c; };
                for (int f = f0; f < f1; ++f) {
                    uint8_t* dfr = d.p + (size_t)f * d.fstride;
#pragma unroll
                    for (int r = 0; r < kWarpRows; ++r)
                        if (ybase + r < d.rows) {
                            uint8_t* q = dfr + (size_t)(ybase + r) * d.step + (size_t)x * 3;
                            if (ragd) store_quad_ragged(q, 0u, 0u, 0u, d.cols - x);
                            else *(U3*)q = U3{0u, 0u, 0u};
                        }
                }
            }
            return;
        }
        if (small && __all(inter)) {
            const unsigned sstep = (unsigned)s.step;
            // Tap pair = 6 bytes at byte 3*x0 of a row.  An UNALIGNED 8-byte load costs ~40 cycles per wave instruction
            // (the address path serialises misaligned lanes; measured, same for ds_read_b64), so the taps are fetched as
            // the three ALIGNED dwords that contain them and shifted into place with v_alignbyte.
            struct U3 { uint32_t a, b, c; };
            f2 fxy[kWarpRows];
            unsigned sh[kWarpRows], oa[kWarpRows];
#pragma unroll
            for (int r = 0; r < kWarpRows; ++r) {
                const float fyy = (float)min(ybase + r, d.rows - 1);
                // (sx, sy) as one packed pair: fmaf(m0, x, fmaf(m1, y, m2)) and fmaf(m3, x, fmaf(m4, y, m5))
                const f2 sxy = __builtin_elementwise_fma(f2{A.m[0], A.m[3]}, f2{fxx, fxx},
                                                         __builtin_elementwise_fma(f2{A.m[1], A.m[4]}, f2{fyy, fyy}, f2{A.m[2], A.m[5]}));
                // sx, sy >= 0 here: the float -> int conversion (truncation) IS the floor, and v_fract_f32 returns the exact
                // sx - floor(sx) the specification states (the difference is representable; no clamp can trigger below 2^23)
                fxy[r] = f2{__builtin_amdgcn_fractf(sxy.x), __builtin_amdgcn_fractf(sxy.y)};
                const unsigned x0 = (unsigned)(int)sxy.x, y0 = (unsigned)(int)sxy.y;
                const unsigned off = __umul24(y0, sstep) + 3u * x0;   // rows start 4-byte aligned (checked by the caller)
                sh[r] = off & 3u;
                oa[r] = off & ~3u;
            }
            const int lane = threadIdx.x & 63;
            const int xq = x & ~3, yi = ybase + (lane & 3);
            unsigned so[kWarpRows / 4];
#pragma unroll
            for (int h = 0; h < kWarpRows / 4; ++h) so[h] = __umul24((unsigned)(yi + 4 * h), (unsigned)d.step) + 3u * (unsigned)xq;
            // Frames are software-pipelined in halves of four rows: while one half's 12 lerps run, the other half's eight
            // tap loads and the next frame's are in flight (the tap registers are reloaded right after their last use), so
            // a wave always has 8-16 loads outstanding instead of alternating between waiting and computing.
            typedef const __attribute__((address_space(1))) uint8_t* cgp;
            typedef __attribute__((address_space(1))) uint8_t* gp;
            typedef uint32_t u3v __attribute__((ext_vector_type(3)));
            typedef __attribute__((address_space(1))) u3v gU3;
            u3v ta[kWarpRows], tb[kWarpRows];
            auto load_half = [&](int h, int f) {
                // uniform frame base pinned to SGPRs: the loads take the scalar-base + 32-bit-offset form
                cgp sf = (cgp)(s.p + (size_t)f * s.fstride);
                asm("" : "+s"(sf));
#pragma unroll
                for (int r = 4 * h; r < 4 * h + 4; ++r) {
                    // (the offsets are made opaque here so that their zero-extension is not hoisted out of the frame loop as
                    //  64-bit values -- instruction selection then no longer sees base + zext(offset) and builds 64-bit
                    //  vector addresses with one v_lshl_add_u64 per load)
                    unsigned o0 = oa[r], o1 = oa[r] + sstep;
                    asm("" : "+v"(o0), "+v"(o1));
                    ta[r] = *(const gU3*)(sf + o0);
                    tb[r] = *(const gU3*)(sf + o1);
                }
            };
            auto finish_half = [&](int h, int f) {
                gp dfr = (gp)(d.p + (size_t)f * d.fstride);
                asm("" : "+s"(dfr));
                uint32_t t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * h + i;
                    const uint32_t alo = __builtin_amdgcn_alignbyte(ta[r].y, ta[r].x, sh[r]), ahi = __builtin_amdgcn_alignbyte(ta[r].z, ta[r].y, sh[r]);
                    const uint32_t blo = __builtin_amdgcn_alignbyte(tb[r].y, tb[r].x, sh[r]), bhi = __builtin_amdgcn_alignbyte(tb[r].z, tb[r].y, sh[r]);
                    t[i] = bilerp_bgr<true>(alo, ahi, blo, bhi, fxy[r]);
                }
                // quad transpose, then lane 4q+i stores the 12 bytes of pixels 4q..4q+3 of row 4h+i
                quad_transpose4(t, lane);
                if (xq < d.cols && yi + 4 * h < d.rows) {
                    unsigned o = so[h];
                    asm("" : "+v"(o));
                    // 4 x {b g r 0} -> 12 bytes with three byte permutes
                    const u3v val = {__builtin_amdgcn_perm(t[1], t[0], 0x04020100u), __builtin_amdgcn_perm(t[2], t[1], 0x05040201u), __builtin_amdgcn_perm(t[3], t[2], 0x06050402u)};
                    typedef uint32_t u3m __attribute__((ext_vector_type(3), aligned(1)));
                    typedef __attribute__((address_space(1))) u3m gU3m;
                    if (ragd && d.cols - xq < 4) store_quad_ragged(d.p + (size_t)f * d.fstride + o, val.x, val.y, val.z, d.cols - xq);
                    else *(gU3m*)(dfr + o) = val;
                }
            };
            load_half(0, f0);
            load_half(1, f0);
            for (int f = f0; f < f1 - 1; ++f) {
                finish_half(0, f);
                load_half(0, f + 1);
                finish_half(1, f);
                load_half(1, f + 1);
            }
            finish_half(0, f1 - 1);
            finish_half(1, f1 - 1);
            return;
        }
    }

    for (int f = f0; f < f1; ++f) {
    const uint8_t* sf = s.p + (size_t)f * s.fstride;
    uint8_t* dfr = d.p + (size_t)f * d.fstride;
    uint2 ta[kWarpRows], tb[kWarpRows];
    float fx[kWarpRows], fy[kWarpRows];
    int sh[kWarpRows];          // bit shift that puts the tap pair at bit 0 of the 8-byte window
    uint32_t ok[kWarpRows];     // bit0 inside, bit1 vx0, bit2 vx1, bit3 vy0, bit4 vy1
#pragma unroll
    for (int r = 0; r < kWarpRows; ++r) {
        const float fyy = (float)min(ybase + r, d.rows - 1);
        const float sx = fmaf(A.m[0], fxx, fmaf(A.m[1], fyy, A.m[2]));
        const float sy = fmaf(A.m[3], fxx, fmaf(A.m[4], fyy, A.m[5]));
        const bool inside = sx > -1.0f && sx < (float)s.cols && sy > -1.0f && sy < (float)s.rows;
        const float x0f = floorf(inside ? sx : 0.0f), y0f = floorf(inside ? sy : 0.0f);
        const int x0 = (int)x0f, y0 = (int)y0f;
        fx[r] = (inside ? sx : 0.0f) - x0f;
        fy[r] = (inside ? sy : 0.0f) - y0f;
        const bool vx0 = x0 >= 0, vx1 = x0 + 1 < s.cols, vy0 = y0 >= 0, vy1 = y0 + 1 < s.rows;
        ok[r] = (inside ? 1u : 0u) | (vx0 ? 2u : 0u) | (vx1 ? 4u : 0u) | (vy0 ? 8u : 0u) | (vy1 ? 16u : 0u);
        const int off = 3 * x0, offc = min(max(off, 0), rowbytes - 8);
        sh[r] = (off - offc) * 8;
        const uint8_t* ra = sf + (size_t)max(y0, 0) * s.step + offc;
        const uint8_t* rb = sf + (size_t)min(y0 + 1, s.rows - 1) * s.step + offc;
        __builtin_memcpy(&ta[r], ra, 8);
        __builtin_memcpy(&tb[r], rb, 8);
    }
#pragma unroll
    for (int r = 0; r < kWarpRows; ++r) {
        uint64_t a64 = ((uint64_t)ta[r].y << 32) | ta[r].x, b64 = ((uint64_t)tb[r].y << 32) | tb[r].x;
        if (sh[r] != 0) {   // only at the left / right source edge
            a64 = sh[r] > 0 ? (a64 >> sh[r]) : (a64 << -sh[r]);
            b64 = sh[r] > 0 ? (b64 >> sh[r]) : (b64 << -sh[r]);
        }
        uint32_t alo = (uint32_t)a64, ahi = (uint32_t)(a64 >> 32), blo = (uint32_t)b64, bhi = (uint32_t)(b64 >> 32);
        const bool in = ok[r] & 1, vx0 = ok[r] & 2, vx1 = ok[r] & 4, vy0 = ok[r] & 8, vy1 = ok[r] & 16;
        // zero the taps that fall outside the source (constant border 0)
        const uint32_t m0 = vx0 ? 0x00ffffffu : 0u, m1l = vx1 ? 0xff000000u : 0u, m1h = vx1 ? 0x0000ffffu : 0u;
        alo &= vy0 ? (m0 | m1l) : 0u;
        ahi &= vy0 ? m1h : 0u;
        blo &= vy1 ? (m0 | m1l) : 0u;
        bhi &= vy1 ? m1h : 0u;
        const float p00[3] = {(float)(alo & 0xff), (float)((alo >> 8) & 0xff), (float)((alo >> 16) & 0xff)};
        const float p01[3] = {(float)(alo >> 24), (float)(ahi & 0xff), (float)((ahi >> 8) & 0xff)};
        const float p10[3] = {(float)(blo & 0xff), (float)((blo >> 8) & 0xff), (float)((blo >> 16) & 0xff)};
        const float p11[3] = {(float)(blo >> 24), (float)(bhi & 0xff), (float)((bhi >> 8) & 0xff)};
        uint32_t px = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float top = fmaf(fx[r], p01[c] - p00[c], p00[c]);
            const float bot = fmaf(fx[r], p11[c] - p10[c], p10[c]);
            const float v = fmaf(fy[r], bot - top, top);
            px |= (uint32_t)round_half_up_u8(v) << (8 * c);
        }
        px = in ? px : 0u;
        // lanes 4q..4q+3 -> 12 bytes stored by lane 4q  (row_shl:n brings lane+n's value; quads stay inside a DPP row of 16)
        const uint32_t p1 = __builtin_amdgcn_update_dpp(0u, px, 0x101, 0xf, 0xf, true);
        const uint32_t p2 = __builtin_amdgcn_update_dpp(0u, px, 0x102, 0xf, 0xf, true);
        const uint32_t p3 = __builtin_amdgcn_update_dpp(0u, px, 0x103, 0xf, 0xf, true);
        // (stores may be conditional here: no load is outstanding any more)
        if ((threadIdx.x & 3) == 0 && x < d.cols && ybase + r < d.rows) {
            struct U3 { uint32_t a, b, c; };
            uint8_t* q = dfr + (size_t)(ybase + r) * d.step + (size_t)x * 3;
            if (ragd) store_quad_ragged(q, px | (p1 << 24), (p1 >> 8) | (p2 << 16), (p2 >> 16) | (p3 << 8), d.cols - x);
            else *(U3*)q = U3{px | (p1 << 24), (p1 >> 8) | (p2 << 16), (p2 >> 16) | (p3 << 8)};
        }
    }
    }
}

// blockIdx.z = a GROUP of fpg consecutive frames (see warp_bgr_wave).
// (Two tile orders that keep vertically neighbouring tiles on one XCD's L2 were measured against the plain order: 1-10 % slower.)
__global__ __launch_bounds__(kBlock) void k_warp_affine_bgr(View s, View d, Affine A, int fpg)
{
    // lane -> pixel: a wave covers kWarpWW columns x (64 / kWarpWW) bands of kWarpRows rows, a workgroup kWarpTW columns
    const int wlane = threadIdx.x & 63, wwave = threadIdx.x >> 6;
    const int x = blockIdx.x * kWarpTW + (wwave % (kWarpTW / kWarpWW)) * kWarpWW + (wlane % kWarpWW);   // d.cols % 4 == 0: quads never straddle the row end
    const int ybase = ((int)blockIdx.y * (kBlock / kWarpTW) + (wwave / (kWarpTW / kWarpWW)) * (64 / kWarpWW) + wlane / kWarpWW) * kWarpRows;
    const int f0 = (int)blockIdx.z * fpg;
    warp_bgr_wave(s, d, A, f0, min(f0 + fpg, d.n), x, ybase);
}

// ---- warpAffine BGR through an LDS-staged source patch ------------------------------------------------------------------
// k_warp_affine_bgr is bound by its tap gathers: every pixel costs two 12-byte vector loads whose 64 lanes spread over many
// cache lines, and the texture-address path spends ~32 cycles on each such wave instruction whatever its width (4 / 8 / 12 /
// 16-byte tap loads time the same; DESIGN_HISTORY.md 4) -- 512 of them per wave and frame against ~330 cycles of arithmetic.  Here a
// workgroup (4 waves, an output tile of 64 columns x 32 rows: near-square, so the rotated source patch is only ~1.4x the
// tile) copies the patch into LDS with coalesced 12-byte loads of 4 pixels each -- three per thread and frame at 7 degrees instead
// of sixteen gathers -- unpacked to one dword per pixel {b g r x}: a pixel's two taps of a row are then two consecutive dwords
// (one ds_read2_b32, no byte alignment work), and with a row pitch chosen by the host for the matrix (warp_lds_plan) the 32
// lanes of a read land on 32 different banks.  Two LDS buffers: the next frame's patch is in flight while this frame's 8
// pixels per thread are computed, one barrier per frame.  pitch / prow / cpr: bytes per staged row (a multiple of 16), staged
// rows and 4-pixel chunks per row; a tile whose patch does not fit, touches the source border or lies outside runs
// warp_bgr_wave (direct gathers) instead.  Same f32 operations in the same order as every other path.
constexpr int kWlTW = 64, kWlTH = 32, kWlMaxG = 6;

// bilerp_bgr on TWO unpacked pixels (p[k] = {p00, p01, p10, p11} of pixel k, each {b g r x}: the upper and the lower tap pair)
// whose weights arrive as the pairs fxp = {fx of pixel 0, fx of pixel 1}, fyp likewise (round 4).
// Channels b, g of a pixel ride in one packed pair as in bilerp_bgr (the pixel's weight broadcast out of its half of fxp / fyp);
// channel r -- which bilerp_bgr finishes with three unpacked operations -- pairs with the OTHER pixel's channel r through
// the whole chain: 3.5 instead of 5 instructions per pixel for that channel, the same f32 operations in the same order.
__device__ __forceinline__ void bilerp_bgrx_pair(const uint32_t (&p)[2][4], f2 fxp, f2 fyp, uint32_t& o0, uint32_t& o1)
{
    const f2 half2 = {0.5f, 0.5f};
    f2 v01[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const f2 a0 = {ub<0>(p[k][0]), ub<1>(p[k][0])}, a1 = {ub<0>(p[k][1]), ub<1>(p[k][1])};
        const f2 b0 = {ub<0>(p[k][2]), ub<1>(p[k][2])}, b1 = {ub<0>(p[k][3]), ub<1>(p[k][3])};
        const f2 top = k ? pk_fma_bc<1>(fxp, a1 - a0, a0) : pk_fma_bc<0>(fxp, a1 - a0, a0);
        const f2 bot = k ? pk_fma_bc<1>(fxp, b1 - b0, b0) : pk_fma_bc<0>(fxp, b1 - b0, b0);
        v01[k] = (k ? pk_fma_bc<1>(fyp, bot - top, top) : pk_fma_bc<0>(fyp, bot - top, top)) + half2;
    }
    const f2 c00 = {ub<2>(p[0][0]), ub<2>(p[1][0])}, c01 = {ub<2>(p[0][1]), ub<2>(p[1][1])};
    const f2 c10 = {ub<2>(p[0][2]), ub<2>(p[1][2])}, c11 = {ub<2>(p[0][3]), ub<2>(p[1][3])};
    const f2 topc = __builtin_elementwise_fma(fxp, c01 - c00, c00);
    const f2 botc = __builtin_elementwise_fma(fxp, c11 - c10, c10);
    const f2 vc = __builtin_elementwise_fma(fyp, botc - topc, topc) + half2;
    o0 = pack_floor3(v01[0].x, v01[0].y, vc.x);
    o1 = pack_floor3(v01[1].x, v01[1].y, vc.y);
}

// quad_transpose4 in 8 instead of 16 VALU instructions (round 4): v_cndmask_b32 is a VOP2 instruction, so its first source
// takes the DPP quad permutation itself -- d = vcc ? own : neighbour's -- and the v_mov_b32_dpp in front of every select goes.
// The four lane masks (even / odd lane, lower / upper pair of a quad) are wave constants in SGPR pairs.  s_nop 1 in front:
// a DPP source needs two wait states after the VALU write of that register, and the compiler's hazard recogniser does not
// look inside an asm block (inside the block every DPP source was written at least three instructions earlier).
template <int N> struct IntC { static constexpr int value = N; };
struct QuadMasks { uint64_t even, odd, lo, hi; };
__device__ __forceinline__ QuadMasks quad_masks() { return QuadMasks{0x5555555555555555ull, 0xaaaaaaaaaaaaaaaaull, 0x3333333333333333ull, 0xccccccccccccccccull}; }
__device__ __forceinline__ void quad_transpose4_dpp(uint32_t (&a)[4], const QuadMasks& m)
{
    uint32_t b0, b1, b2, b3, c0, c1, c2, c3;
    asm("s_nop 1\n\t"
        "s_mov_b64 vcc, %12\n\t"
        "v_cndmask_b32_dpp %0, %9, %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"     // even ? a0 : a1 of lane ^ 1
        "v_cndmask_b32_dpp %2, %11, %10, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   // even ? a2 : a3 of lane ^ 1
        "s_mov_b64 vcc, %13\n\t"
        "v_cndmask_b32_dpp %1, %8, %9, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"     // odd ? a1 : a0 of lane ^ 1
        "v_cndmask_b32_dpp %3, %10, %11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   // odd ? a3 : a2 of lane ^ 1
        "s_mov_b64 vcc, %14\n\t"
        "v_cndmask_b32_dpp %4, %2, %0, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"     // lower pair ? b0 : b2 of lane ^ 2
        "s_nop 0\n\t"
        "v_cndmask_b32_dpp %5, %3, %1, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"     // lower pair ? b1 : b3 of lane ^ 2
        "s_mov_b64 vcc, %15\n\t"
        "v_cndmask_b32_dpp %6, %0, %2, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"     // upper pair ? b2 : b0 of lane ^ 2
        "v_cndmask_b32_dpp %7, %1, %3, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"           // upper pair ? b3 : b1 of lane ^ 2
        : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "s"(m.even), "s"(m.odd), "s"(m.lo), "s"(m.hi)
        : "vcc");
    a[0] = c0; a[1] = c1; a[2] = c2; a[3] = c3;
}

// RAGS: source rows of any alignment (an odd width of a packed image): a chunk's 12 bytes are fetched as the 16 aligned bytes
// that contain them and shifted into place with v_alignbyte when they are written to LDS.
// CH = 1 (one-channel images): the same tiles, patch geometry and LDS layout (one dword per patch pixel).  A chunk's 4 pixels are
// fetched as the 8 aligned bytes that contain pixels x .. x+4 and written as the four dwords {g(x) g(x+1) . .}: ONE ds_read2_b32
// returns the two tap pairs of a pixel.  The lerp is k_warp_affine_gray's (same f32 operations, same order).
__device__ __forceinline__ void warp_gray_frame(const View& s, const View& d, const Affine& A, const int frame, const int x, const int ybase);

template <int CH, bool RAGS>
__global__ __launch_bounds__(kBlock) void k_warp_affine_lds(View s, View d, Affine A, int fpg, int pitch, int prow, int cpr, int gx, int gy, int ntiles,
                                                            int tiles_per_xcd, int strip)
{
    constexpr bool AL = RAGS || CH == 1;   // chunks are fetched as aligned dwords and shifted into place
    extern __shared__ __attribute__((aligned(16))) uint8_t wl_lds[];
    int bx, by, bz;
    if (!wl_tile(tiles_per_xcd, strip, gx, gy, ntiles, bx, by, bz)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = bx * kWlTW + lane, ybase = by * kWlTH + wave * kWarpRows;
    const int f0 = bz * fpg, f1 = min(f0 + fpg, d.n);
    // the tile's source patch: sx and sy are monotonic in x and in y (fmaf rounds monotonically), so the four corners of the
    // (clamped) tile bound every lane's coordinates
    const float cx0 = (float)min(bx * kWlTW, d.cols - 1), cx1 = (float)min(bx * kWlTW + kWlTW - 1, d.cols - 1);
    const float cy0 = (float)min(by * kWlTH, d.rows - 1), cy1 = (float)min(by * kWlTH + kWlTH - 1, d.rows - 1);
    const float t0 = fmaf(A.m[1], cy0, A.m[2]), t1 = fmaf(A.m[1], cy1, A.m[2]), u0 = fmaf(A.m[4], cy0, A.m[5]), u1 = fmaf(A.m[4], cy1, A.m[5]);
    const float xa = fmaf(A.m[0], cx0, t0), xb = fmaf(A.m[0], cx1, t0), xc = fmaf(A.m[0], cx0, t1), xd = fmaf(A.m[0], cx1, t1);
    const float ya = fmaf(A.m[3], cx0, u0), yb = fmaf(A.m[3], cx1, u0), yc = fmaf(A.m[3], cx0, u1), yd = fmaf(A.m[3], cx1, u1);
    const float xmin = fminf(fminf(xa, xb), fminf(xc, xd)), xmax = fmaxf(fmaxf(xa, xb), fmaxf(xc, xd));
    const float ymin = fminf(fminf(ya, yb), fminf(yc, yd)), ymax = fmaxf(fmaxf(ya, yb), fmaxf(yc, yd));
    // INSIDE: every tap of the tile inside the source, one more row of slack below (the last 12-byte chunk of a patch row may read
    // past the row's end into the next row).  Round 4, aligned BGR sources whose width is a multiple of 4: a tile whose patch
    // leaves the source is staged as well -- a 4-pixel chunk is then wholly inside or wholly outside the source, outside chunks
    // are staged as zeros, and a tap of value 0 is exactly the specification's constant border (orc_warp_affine: every tap outside
    // the source contributes 0.0f; a sample with all four taps outside is 0 whatever its weights).  The per-pixel gather code these
    // tiles used to run made them 8 % of the tiles and 12 % (one channel: 25 %) of the launch's time, most of it as its tail.
    // Common to both: 32-bit in-frame offsets, 4-byte aligned rows in every frame.
    const bool inside = xmin >= 0.0f && xmax < (float)(s.cols - 1) && ymin >= 0.0f && ymax < (float)(s.rows - 2);   // NaN -> false
    const bool edge_ok = CH == 3 && !RAGS && (s.cols & 3) == 0 && xmin > -1.0e6f && xmax < 1.0e6f && ymin > -1.0e6f && ymax < 1.0e6f;   // NaN -> false
    bool ok = (inside || edge_ok) &&
              (AL || (((uintptr_t)s.p & 3) == 0 && (s.fstride & 3) == 0 && (s.step & 3) == 0)) && s.step >= 16 && s.step < (1u << 24) && s.rows < (1 << 24) &&
              (unsigned long long)s.rows * s.step < (1ull << 32) && d.step < (1u << 24) && d.rows < (1 << 24) && (unsigned long long)d.rows * d.step < (1ull << 32);
    int ix0 = 0, iy0 = 0;
    if (ok) {
        ix0 = (int)floorf(xmin) & ~3;   // patch columns start at a multiple of 4 pixels = 12 bytes: 4-byte aligned chunk loads
        iy0 = (int)floorf(ymin);
        ok = (int)floorf(xmax) + 2 - ix0 <= 4 * cpr && (int)floorf(ymax) + 2 - iy0 <= prow;
    }
    const bool border = __builtin_amdgcn_readfirstlane((int)!inside) != 0;   // (uniform) the patch leaves the source: some chunks are zeros
    if (!__builtin_amdgcn_readfirstlane((int)ok)) {   // (the same value in every lane of the workgroup)
        if constexpr (CH == 3) warp_bgr_wave(s, d, A, f0, f1, x, ybase);
        else
            for (int f = f0; f < f1; ++f) warp_gray_frame(s, d, A, f, x, ybase);
        return;
    }
    ix0 = __builtin_amdgcn_readfirstlane(ix0);
    iy0 = __builtin_amdgcn_readfirstlane(iy0);

    // ---- frame-invariant per-thread state: lerp weights and the LDS offset of each pixel's upper-left tap ----
    // (round 4) the weights of rows 2k and 2k+1 as the pairs fxp[k] = {fx, fx'}, fyp[k] = {fy, fy'}: a pixel's b / g lerps
    // broadcast its half, the r lerps of the two pixels share packed operations (bilerp_bgrx_pair); the LDS offsets for BOTH
    // patch buffers (the frame loop is unrolled by two, so neither the buffer base nor a select is added per frame)
    const bool ragd = (d.cols & 3) || ((uintptr_t)d.p & 3) || (d.step & 3) || (d.fstride & 3);   // ragged destination (uniform)
    const float fxx = (float)min(x, d.cols - 1);
    const unsigned bufbytes = (unsigned)(pitch * prow);
    typedef __attribute__((address_space(3))) uint8_t* lp;
    typedef const __attribute__((address_space(3))) uint32_t* lcu;
    const unsigned lds0 = (unsigned)(uintptr_t)(lp)wl_lds;   // (the LDS base folded into every precomputed offset: no add per access)
    f2 fxp[kWarpRows / 2], fyp[kWarpRows / 2];
    unsigned la[2][kWarpRows], lb[2][kWarpRows];   // upper / lower tap row
#pragma unroll
    for (int r = 0; r < kWarpRows; ++r) {
        const float fyy = (float)min(ybase + r, d.rows - 1);
        const f2 sxy = __builtin_elementwise_fma(f2{A.m[0], A.m[3]}, f2{fxx, fxx}, __builtin_elementwise_fma(f2{A.m[1], A.m[4]}, f2{fyy, fyy}, f2{A.m[2], A.m[5]}));
        const f2 fl = __builtin_elementwise_floor(sxy);
        fxp[r >> 1][r & 1] = sxy.x - fl.x;   // (the specification's sx - floor(sx): exact; also right of / below zero)
        fyp[r >> 1][r & 1] = sxy.y - fl.y;
        la[0][r] = lds0 + __umul24((unsigned)((int)fl.y - iy0), (unsigned)pitch) + 4u * (unsigned)((int)fl.x - ix0);
        la[1][r] = la[0][r] + bufbytes;
        lb[0][r] = la[0][r] + (unsigned)pitch;
        lb[1][r] = la[1][r] + (unsigned)pitch;
        asm volatile("" : "+v"(la[0][r]), "+v"(la[1][r]), "+v"(lb[0][r]), "+v"(lb[1][r]));   // (registers, not sums re-formed at every use)
    }
    const int xq = x & ~3, yi = ybase + (lane & 3);
    unsigned so[kWarpRows / 4];
#pragma unroll
    for (int h = 0; h < kWarpRows / 4; ++h) so[h] = __umul24((unsigned)(yi + 4 * h), (unsigned)d.step) + (unsigned)CH * (unsigned)xq;
    // ---- staging plan: chunk c = 4 pixels (12 source bytes -> 16 LDS bytes); thread t copies chunks t, t + 256, ... ----
    const int nchunks = prow * cpr;
    const unsigned cpr_magic = (1u << 20) / (unsigned)cpr + 1u;   // c / cpr == (c * cpr_magic) >> 20 for every c < 1536 and cpr <= 755 (here cpr * prow <= 1536, prow >= 3): one division instead of one per chunk slot             // <= kWlMaxG * 256 (host)
    // (a frame's last row ends at (rows - 1) * step + cols * CH: a padded LAST row need not be allocated -- rcv_view guarantees no more)
    const unsigned frame_lim = ((unsigned)(s.rows - 1) * (unsigned)s.step + (unsigned)(s.cols * CH) - (CH == 1 ? 8u : (RAGS ? 16u : 12u))) & (AL ? ~0u : ~3u);
    unsigned goff[kWlMaxG], loff[2][kWlMaxG];
    unsigned gzero = 0;   // bit g: chunk slot g lies outside the source (border tiles: staged as zeros)
    bool gval[kWlMaxG];
#pragma unroll
    for (int g = 0; g < kWlMaxG; ++g) {
        const int c = (int)threadIdx.x + kBlock * g;
        gval[g] = c < nchunks;
        const int row = gval[g] ? (int)(((unsigned)c * cpr_magic) >> 20) : 0, col = gval[g] ? c - row * cpr : 0;   // (other threads re-read the patch's first chunk: a cache hit)
        // rows below the source and a chunk past the frame's end are read from a clamped position: no tap lies in them
        const int ry = iy0 + row, px = ix0 + 4 * col;   // (border tiles: either may lie outside the source)
        gzero |= (unsigned)(ry < 0 || ry >= s.rows || px < 0 || px + 4 > s.cols) << g;
        goff[g] = min(__umul24((unsigned)min(max(ry, 0), s.rows - 1), (unsigned)s.step) + (unsigned)(CH * max(px, 0)), frame_lim);
        loff[0][g] = lds0 + (unsigned)(row * pitch + 16 * col);
        loff[1][g] = loff[0][g] + bufbytes;
        asm volatile("" : "+v"(loff[0][g]), "+v"(loff[1][g]));
    }
    const int ng = (nchunks + kBlock - 1) / kBlock;
    typedef const __attribute__((address_space(1))) uint8_t* cgp;
    typedef __attribute__((address_space(1))) uint8_t* gp;
    typedef uint32_t u3v __attribute__((ext_vector_type(3)));
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) u3v gU3;
    typedef __attribute__((address_space(1))) u4v gU4;
    typedef __attribute__((address_space(3))) u4v* lU4;
    u4v G[kWlMaxG];          // (RAGS / one channel: the aligned bytes that contain the chunk)
    u3v G3[kWlMaxG];         // (aligned BGR: the chunk's 12 bytes)
    unsigned gmis[kWlMaxG];  // RAGS: byte position of the chunk inside them
    auto gload = [&](int f) {
        const uint8_t* fb = s.p + (size_t)f * s.fstride;
        const unsigned fmis = AL ? (unsigned)((uintptr_t)fb & 3) : 0u;   // the frame base, aligned down: offsets stay non-negative
        cgp sf = (cgp)(fb - fmis);
        asm("" : "+s"(sf));
#pragma unroll
        for (int g = 0; g < kWlMaxG; ++g)
            if (g < ng) {   // uniform
                if constexpr (CH == 1) {
                    unsigned o = goff[g] + fmis;
                    asm("" : "+v"(o));
                    typedef uint32_t u2v_ __attribute__((ext_vector_type(2)));
                    typedef __attribute__((address_space(1))) u2v_ gU2;
                    gmis[g] = o & 3u;
                    const u2v_ t = *(const gU2*)(sf + (o & ~3u));
                    G[g] = u4v{t.x, t.y, 0u, 0u};
                } else if constexpr (RAGS) {
                    unsigned o = goff[g] + fmis;
                    asm("" : "+v"(o));
                    gmis[g] = o & 3u;
                    G[g] = *(const gU4*)(sf + (o & ~3u));
                } else {
                    // (the in-place barrier keeps the zero-extension of the offset in this block: scalar frame base + 32-bit
                    //  thread offset is then ONE instruction with no address arithmetic and no copy)
                    asm volatile("" : "+v"(goff[g]));
                    G3[g] = *(const gU3*)(sf + goff[g]);
                }
            }
    };
    const QuadMasks qm = quad_masks();
    // the patch of the frame whose chunks G / G3 hold goes to LDS buffer B
    auto stage = [&](auto Bc) {
        constexpr int B = decltype(Bc)::value;
        if constexpr (CH == 3 && !RAGS) {
            if (border) {   // (uniform; the volatile asm keeps it a branch: if-converted it is three selects per chunk on EVERY tile)
                asm volatile("; tile whose patch leaves the source: chunks outside are zeros");
#pragma unroll
                for (int g = 0; g < kWlMaxG; ++g)
                    if ((gzero >> g) & 1u) G3[g] = u3v{0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int g = 0; g < kWlMaxG; ++g)
            if (gval[g]) {
                if constexpr (CH == 1) {   // pixels x .. x+4 -> four dwords {g(x+i) g(x+i+1) . .}
                    const uint32_t e0 = __builtin_amdgcn_alignbyte(G[g].y, G[g].x, gmis[g]), e1 = __builtin_amdgcn_alignbyte(0u, G[g].y, gmis[g]);
                    *(lU4)(uintptr_t)(loff[B][g]) = u4v{e0, __builtin_amdgcn_alignbyte(e1, e0, 1), __builtin_amdgcn_alignbyte(e1, e0, 2), __builtin_amdgcn_alignbyte(e1, e0, 3)};
                    continue;
                }
                // 12 bytes {b g r b | g r b g | r b g r} -> four {b g r x} dwords
                uint32_t c0, c1, c2;
                if constexpr (RAGS) {
                    c0 = __builtin_amdgcn_alignbyte(G[g].y, G[g].x, gmis[g]);
                    c1 = __builtin_amdgcn_alignbyte(G[g].z, G[g].y, gmis[g]);
                    c2 = __builtin_amdgcn_alignbyte(G[g].w, G[g].z, gmis[g]);
                } else {
                    c0 = G3[g].x; c1 = G3[g].y; c2 = G3[g].z;
                }
                *(lU4)(uintptr_t)(loff[B][g]) = u4v{c0, __builtin_amdgcn_alignbyte(c1, c0, 3), __builtin_amdgcn_alignbyte(c2, c1, 2), c2 >> 8};
            }
    };
    // the tile of frame f from LDS buffer B.  INNER (a tile that lies inside an aligned destination: every lane stores) has NO
    // branch around its two stores -- see the loop below.
    auto compute = [&](const int f, auto Bc, auto Ic) {
        constexpr int B = decltype(Bc)::value;
        constexpr bool INNER = decltype(Ic)::value != 0;
        gp dfr = (gp)(d.p + (size_t)f * d.fstride);
        asm("" : "+s"(dfr));
#pragma unroll
        for (int h = 0; h < kWarpRows / 4; ++h) {
            uint32_t t[4];
            if constexpr (CH == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * h + i;
                    const uint32_t a = *(lcu)(uintptr_t)(la[B][r]), b = *(lcu)(uintptr_t)(lb[B][r]);
                    const f2 p0 = {ub<0>(a), ub<0>(b)}, p1 = {ub<1>(a), ub<1>(b)};
                    const f2 tb2 = (r & 1) ? pk_fma_bc<1>(fxp[r >> 1], p1 - p0, p0) : pk_fma_bc<0>(fxp[r >> 1], p1 - p0, p0);
                    t[i] = trunc_u32(fmaf(fyp[r >> 1][r & 1], tb2.y - tb2.x, tb2.x) + 0.5f);   // every tap inside the source: an integer in [0, 255]
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = 4 * h + 2 * i;
                    uint32_t p[2][4];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const lcu pa = (lcu)(uintptr_t)(la[B][r + k]), pb = (lcu)(uintptr_t)(lb[B][r + k]);
                        p[k][0] = pa[0]; p[k][1] = pa[1]; p[k][2] = pb[0]; p[k][3] = pb[1];
                    }
                    bilerp_bgrx_pair(p, fxp[r >> 1], fyp[r >> 1], t[2 * i], t[2 * i + 1]);
                }
            }
            // quad transpose, then lane 4q+i stores the 12 bytes of pixels 4q..4q+3 of row 4h+i
            // (the same transpose through LDS -- four dword writes and one ds_read_b128 per wave instead of 16 VALU
            //  instructions -- timed the same: 1.655 against 1.645 ms)
            quad_transpose4_dpp(t, qm);
            if constexpr (CH == 1) {
                if (INNER || (xq < d.cols && yi + 4 * h < d.rows)) {   // lane 4q+i: pixels 4q .. 4q+3 of row 4h+i as one dword
                    typedef uint32_t u1m __attribute__((aligned(1)));
                    const uint32_t v = t[0] | (t[1] << 8) | (t[2] << 16) | (t[3] << 24);
                    if constexpr (INNER) {
                        typedef __attribute__((address_space(1))) u1m gU1m;
                        asm volatile("" : "+v"(so[h]));
                        *(gU1m*)(dfr + so[h]) = v;
                    } else {
                        uint8_t* q = d.p + (size_t)f * d.fstride + so[h];
                        if (d.cols - xq >= 4) *(u1m*)q = v;
                        else
                            for (int j = 0; j < d.cols - xq; ++j) q[j] = (uint8_t)(v >> (8 * j));
                    }
                }
                continue;
            }
            const u3v val = {__builtin_amdgcn_perm(t[1], t[0], 0x04020100u), __builtin_amdgcn_perm(t[2], t[1], 0x05040201u), __builtin_amdgcn_perm(t[3], t[2], 0x06050402u)};
            // (one store instruction for aligned and for byte-aligned rows -- the hardware takes either; only the row's last,
            //  partial quad of a width that is not a multiple of 4 goes out byte by byte)
            typedef uint32_t u3m __attribute__((ext_vector_type(3), aligned(1)));
            typedef __attribute__((address_space(1))) u3m gU3m;
            if constexpr (INNER) {
                asm volatile("" : "+v"(so[h]));   // (as goff above)
                *(gU3m*)(dfr + so[h]) = val;
            } else if (xq < d.cols && yi + 4 * h < d.rows) {
                asm volatile("" : "+v"(so[h]));
                if (ragd && d.cols - xq < 4) store_quad_ragged(d.p + (size_t)f * d.fstride + so[h], val.x, val.y, val.z, d.cols - xq);
                else *(gU3m*)(dfr + so[h]) = val;
            }
        }
    };
    // The frame loop, rotated: compute(f) | stage(f + 1) | barrier | loads of f + 2.  The chunk registers are next touched in
    // stage(f + 1), and between their loads and that point every path issues the same vector-memory instructions -- the two
    // output stores of compute(f), unconditional for INNER tiles -- so the compiler's wait there is vmcnt(2): the wave waits for
    // its chunk loads, NOT for the stores it has just issued.  With a branch around the stores (or the wait at the head of the
    // loop, where the prologue's path has no stores) that wait is vmcnt(0) and every wave sits out the write latency of its own
    // stores once per frame: 1.65 -> 1.22 ms with the stores removed, against 1.13 ms for the arithmetic alone.
    // One barrier per frame: stage(f + 1) overwrites the buffer compute(f - 1) read, which every wave left before barrier f.
    auto run = [&](auto Ic) {
        gload(f0);
        stage(IntC<0>{});
        __syncthreads();
        if (f0 + 1 < f1) gload(f0 + 1);
        for (int f = f0;; f += 2) {
            compute(f, IntC<0>{}, Ic);
            if (f + 1 >= f1) break;
            stage(IntC<1>{});
            __syncthreads();
            if (f + 2 < f1) gload(f + 2);
            compute(f + 1, IntC<1>{}, Ic);
            if (f + 2 >= f1) break;
            stage(IntC<0>{});
            __syncthreads();
            if (f + 3 < f1) gload(f + 3);
        }
    };
    const bool inner = !ragd && bx * kWlTW + kWlTW <= d.cols && by * kWlTH + kWlTH <= d.rows;   // (uniform)
    if (inner) run(IntC<1>{});
    else run(IntC<0>{});
}

// ---- one-channel warpAffine, FOUR FRAMES per LDS pass (round 3) ---------------------------------------------------------------
// k_warp_affine_lds<1> stages one frame at a time: per frame and thread 3.5 chunk loads, 3.5 ds_write_b128, 16 ds_read_b32 and a
// barrier for 8 cheap pixels (10 VALU each) -- the LDS pipe and the barrier, not the arithmetic, set its pace (1.06 ms for 32 8K
// frames, 24 % of the HBM roofline).  A one-channel patch pixel fills one byte of its LDS dword; this kernel fills all four with
// the SAME pixel of four consecutive frames {f0 f1 f2 f3}: a chunk's four dwords (one per frame, 4 pixels each) go through a
// 4 x 4 byte transpose (8 v_perm) into one ds_write_b128, one ds_read_b64 returns a tap pair of all four frames, the coordinates /
// weights are shared anyway, the lerps of two frames ride in one packed-f32 operation, the results of four frames go through ONE
// quad transpose (as dwords {f0 f1 f2 f3}) and a byte transpose back.  A quarter of the LDS instructions and barriers per frame,
// the same f32 operations per pixel in the same order.  Aligned sources only (base, step, frame stride multiples of 4: a chunk is
// one aligned dword); everything else stays on k_warp_affine_lds<1>.
__device__ __forceinline__ void bytes4x4_transpose(uint32_t (&v)[4])
{
    const uint32_t a = __builtin_amdgcn_perm(v[1], v[0], 0x05010400u), b = __builtin_amdgcn_perm(v[1], v[0], 0x07030602u);   // {v0.0 v1.0 v0.1 v1.1}, {v0.2 v1.2 v0.3 v1.3}
    const uint32_t c = __builtin_amdgcn_perm(v[3], v[2], 0x05010400u), e = __builtin_amdgcn_perm(v[3], v[2], 0x07030602u);
    v[0] = __builtin_amdgcn_perm(c, a, 0x05040100u);   // {v0.0 v1.0 v2.0 v3.0}
    v[1] = __builtin_amdgcn_perm(c, a, 0x07060302u);
    v[2] = __builtin_amdgcn_perm(e, b, 0x05040100u);
    v[3] = __builtin_amdgcn_perm(e, b, 0x07060302u);
}

// NG: chunk slots per thread (prow * cpr <= 256 NG): 4 covers rotations up to ~10 degrees and keeps 8 registers of prefetch state free
template <int NG>
__global__ __launch_bounds__(kBlock) void k_warp_gray_lds4(View s, View d, Affine A, int fpg, int pitch, int prow, int cpr, int gx, int gy, int ntiles,
                                                           int tiles_per_xcd, int strip)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t wg_lds[];
    int bx, by, bz;
    if (!wl_tile(tiles_per_xcd, strip, gx, gy, ntiles, bx, by, bz)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = bx * kWlTW + lane, ybase = by * kWlTH + wave * kWarpRows;
    const int f0 = bz * fpg, f1 = min(f0 + fpg, d.n);
    const float cx0 = (float)min(bx * kWlTW, d.cols - 1), cx1 = (float)min(bx * kWlTW + kWlTW - 1, d.cols - 1);
    const float cy0 = (float)min(by * kWlTH, d.rows - 1), cy1 = (float)min(by * kWlTH + kWlTH - 1, d.rows - 1);
    const float t0 = fmaf(A.m[1], cy0, A.m[2]), t1 = fmaf(A.m[1], cy1, A.m[2]), u0 = fmaf(A.m[4], cy0, A.m[5]), u1 = fmaf(A.m[4], cy1, A.m[5]);
    const float xa = fmaf(A.m[0], cx0, t0), xb = fmaf(A.m[0], cx1, t0), xc = fmaf(A.m[0], cx0, t1), xd = fmaf(A.m[0], cx1, t1);
    const float ya = fmaf(A.m[3], cx0, u0), yb = fmaf(A.m[3], cx1, u0), yc = fmaf(A.m[3], cx0, u1), yd = fmaf(A.m[3], cx1, u1);
    const float xmin = fminf(fminf(xa, xb), fminf(xc, xd)), xmax = fmaxf(fmaxf(xa, xb), fmaxf(xc, xd));
    const float ymin = fminf(fminf(ya, yb), fminf(yc, yd)), ymax = fmaxf(fmaxf(ya, yb), fmaxf(yc, yd));
    // (inside / border tiles: as in k_warp_affine_lds -- a patch that leaves a source whose width is a multiple of 4 is staged with
    //  zeros for the chunks outside)
    const bool inside = xmin >= 0.0f && xmax < (float)(s.cols - 1) && ymin >= 0.0f && ymax < (float)(s.rows - 2);   // NaN -> false
    const bool edge_ok = (s.cols & 3) == 0 && xmin > -1.0e6f && xmax < 1.0e6f && ymin > -1.0e6f && ymax < 1.0e6f;   // NaN -> false
    bool ok = (inside || edge_ok) &&
              s.step >= 16 && s.step < (1u << 24) && s.rows < (1 << 24) && (unsigned long long)s.rows * s.step < (1ull << 32) && d.step < (1u << 24) &&
              d.rows < (1 << 24) && (unsigned long long)d.rows * d.step < (1ull << 32);
    int ix0 = 0, iy0 = 0;
    if (ok) {
        ix0 = (int)floorf(xmin) & ~3;
        iy0 = (int)floorf(ymin);
        ok = (int)floorf(xmax) + 2 - ix0 <= 4 * cpr && (int)floorf(ymax) + 2 - iy0 <= prow;
    }
    const bool border = __builtin_amdgcn_readfirstlane((int)!inside) != 0;   // (uniform)
    if (!__builtin_amdgcn_readfirstlane((int)ok)) {
        for (int f = f0; f < f1; ++f) warp_gray_frame(s, d, A, f, x, ybase);
        return;
    }
    ix0 = __builtin_amdgcn_readfirstlane(ix0);
    iy0 = __builtin_amdgcn_readfirstlane(iy0);
    const float fxx = (float)min(x, d.cols - 1);
    typedef __attribute__((address_space(3))) uint8_t* lp;
    typedef const __attribute__((address_space(3))) uint32_t* lcu;
    const unsigned lds0 = (unsigned)(uintptr_t)(lp)wg_lds;   // (the LDS base folded into every precomputed offset)
    const unsigned bufbytes = (unsigned)(pitch * prow);
    f2 fxy[kWarpRows];
    unsigned la[2][kWarpRows];   // (per patch buffer: the pass loop is unrolled by two)
#pragma unroll
    for (int r = 0; r < kWarpRows; ++r) {
        const float fyy = (float)min(ybase + r, d.rows - 1);
        const f2 sxy = __builtin_elementwise_fma(f2{A.m[0], A.m[3]}, f2{fxx, fxx}, __builtin_elementwise_fma(f2{A.m[1], A.m[4]}, f2{fyy, fyy}, f2{A.m[2], A.m[5]}));
        const f2 fl = __builtin_elementwise_floor(sxy);
        fxy[r] = sxy - fl;   // (the specification's sx - floor(sx): exact; also left of / above zero)
        la[0][r] = lds0 + __umul24((unsigned)((int)fl.y - iy0), (unsigned)pitch) + 4u * (unsigned)((int)fl.x - ix0);
        la[1][r] = la[0][r] + bufbytes;
        asm volatile("" : "+v"(la[0][r]), "+v"(la[1][r]));
    }
    const int xq = x & ~3, yi = ybase + (lane & 3);
    unsigned so[kWarpRows / 4];
#pragma unroll
    for (int h = 0; h < kWarpRows / 4; ++h) so[h] = __umul24((unsigned)(yi + 4 * h), (unsigned)d.step) + (unsigned)xq;
    // ---- staging plan: chunk c = 4 pixels = one aligned source dword per frame -> 16 LDS bytes {4 pixels x 4 frames} ----
    const int nchunks = prow * cpr;
    const unsigned cpr_magic = (1u << 20) / (unsigned)cpr + 1u;   // c / cpr == (c * cpr_magic) >> 20 for every c < 1536 and cpr <= 755 (here cpr * prow <= 1536, prow >= 3): one division instead of one per chunk slot
    const unsigned frame_lim = ((unsigned)(s.rows - 1) * (unsigned)s.step + (unsigned)s.cols - 4u) & ~3u;
    unsigned goff[NG], loff[2][NG];
    unsigned gzero = 0;   // bit g: chunk slot g lies outside the source (border tiles: staged as zeros)
    bool gval[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int c = (int)threadIdx.x + kBlock * g;
        gval[g] = c < nchunks;
        const int row = gval[g] ? (int)(((unsigned)c * cpr_magic) >> 20) : 0, col = gval[g] ? c - row * cpr : 0;
        const int ry = iy0 + row, px = ix0 + 4 * col;   // (border tiles: either may lie outside the source)
        gzero |= (unsigned)(ry < 0 || ry >= s.rows || px < 0 || px + 4 > s.cols) << g;
        goff[g] = min(__umul24((unsigned)min(max(ry, 0), s.rows - 1), (unsigned)s.step) + (unsigned)max(px, 0), frame_lim);
        loff[0][g] = lds0 + (unsigned)(row * pitch + 16 * col);
        loff[1][g] = loff[0][g] + bufbytes;
        asm volatile("" : "+v"(loff[0][g]), "+v"(loff[1][g]));
    }
    const int ng = (nchunks + kBlock - 1) / kBlock;
    typedef const __attribute__((address_space(1))) uint8_t* cgp;
    typedef __attribute__((address_space(1))) uint8_t* gp;
    typedef __attribute__((address_space(1))) uint32_t gU1;
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u4v* lU4;
    uint32_t G[NG][4];
    auto gload = [&](int fb) {   // frames fb .. fb + 3 (past the group's last frame: that frame again, never stored)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cgp sf = (cgp)(s.p + (size_t)min(fb + k, f1 - 1) * s.fstride);
            asm("" : "+s"(sf));
#pragma unroll
            for (int g = 0; g < NG; ++g)
                if (g < ng) {   // uniform
                    asm volatile("" : "+v"(goff[g]));   // (scalar frame base + 32-bit thread offset in one instruction: see k_warp_affine_lds)
                    G[g][k] = *(const gU1*)(sf + goff[g]);
                }
        }
    };
    const bool ragd = (d.cols & 3) != 0;
    const QuadMasks qm = quad_masks();
    auto stage = [&](auto Bc) {
        constexpr int B = decltype(Bc)::value;
        if (border) {   // (uniform; kept a branch by the volatile asm)
            asm volatile("; tile whose patch leaves the source: chunks outside are zeros");
#pragma unroll
            for (int g = 0; g < NG; ++g)
                if ((gzero >> g) & 1u) G[g][0] = G[g][1] = G[g][2] = G[g][3] = 0u;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
            if (gval[g]) {
                uint32_t v[4] = {G[g][0], G[g][1], G[g][2], G[g][3]};   // frame k: pixels x .. x+3
                bytes4x4_transpose(v);                                  // pixel j: frames 0 .. 3
                *(lU4)(uintptr_t)(loff[B][g]) = u4v{v[0], v[1], v[2], v[3]};
            }
    };
    // INNER: a tile inside the destination, whose width is a multiple of 4, in a group of whole passes -- eight unconditional stores
    auto compute = [&](const int fb, auto Bc, auto Ic) {
        constexpr int B = decltype(Bc)::value;
        constexpr bool INNER = decltype(Ic)::value != 0;
#pragma unroll
        for (int h = 0; h < kWarpRows / 4; ++h) {
            uint32_t w[4];   // row 4h + i, this lane's pixel: {frame 0, 1, 2, 3}
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * h + i;
                const lcu pa = (lcu)(uintptr_t)(la[B][r]), pb = (lcu)(uintptr_t)(la[B][r] + (unsigned)pitch);
                const uint32_t a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
                const f2 half2 = {0.5f, 0.5f};
                // frames 0 and 1, frames 2 and 3: per frame top = fma(fx, p01 - p00, p00), bot = fma(fx, p11 - p10, p10),
                // v = fma(fy, bot - top, top) + 0.5 -- k_warp_affine_gray's operations, two frames per packed instruction
                const f2 p00 = {ub<0>(a0), ub<1>(a0)}, p01 = {ub<0>(a1), ub<1>(a1)}, p10 = {ub<0>(b0), ub<1>(b0)}, p11 = {ub<0>(b1), ub<1>(b1)};
                const f2 q00 = {ub<2>(a0), ub<3>(a0)}, q01 = {ub<2>(a1), ub<3>(a1)}, q10 = {ub<2>(b0), ub<3>(b0)}, q11 = {ub<2>(b1), ub<3>(b1)};
                const f2 top = pk_fma_bc<0>(fxy[r], p01 - p00, p00), bot = pk_fma_bc<0>(fxy[r], p11 - p10, p10);
                const f2 tpq = pk_fma_bc<0>(fxy[r], q01 - q00, q00), btq = pk_fma_bc<0>(fxy[r], q11 - q10, q10);
                const f2 v01 = pk_fma_bc<1>(fxy[r], bot - top, top) + half2, v23 = pk_fma_bc<1>(fxy[r], btq - tpq, tpq) + half2;
                uint32_t px = pack_floor3(v01.x, v01.y, v23.x);
                asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(px) : "v"(v23.y));
                w[i] = px;
            }
            quad_transpose4_dpp(w, qm);   // lane 4q + i: row 4h + i, pixels 4q .. 4q+3, each {frame 0 .. 3}
            bytes4x4_transpose(w);        // w[k]: frame k, pixels 4q .. 4q+3
            if constexpr (INNER) {
                asm volatile("" : "+v"(so[h]));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    gp dfr = (gp)(d.p + (size_t)(fb + k) * d.fstride);
                    asm("" : "+s"(dfr));
                    typedef uint32_t u1m __attribute__((aligned(1)));
                    typedef __attribute__((address_space(1))) u1m gU1m;
                    *(gU1m*)(dfr + so[h]) = w[k];
                }
            } else if (xq < d.cols && yi + 4 * h < d.rows) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (fb + k < f1) {   // uniform
                        typedef uint32_t u1m __attribute__((aligned(1)));
                        uint8_t* q = d.p + (size_t)(fb + k) * d.fstride + so[h];
                        if (!ragd || d.cols - xq >= 4) *(u1m*)q = w[k];
                        else
                            for (int j = 0; j < d.cols - xq; ++j) q[j] = (uint8_t)(w[k] >> (8 * j));
                    }
            }
        }
    };
    // the pass loop, rotated as in k_warp_affine_lds: compute(pass) | stage(pass + 1) | barrier | loads of pass + 2 -- the wait
    // in front of stage() then covers the chunk loads only, not the stores compute() has just issued
    auto run = [&](auto Ic) {
        gload(f0);
        stage(IntC<0>{});
        __syncthreads();
        if (f0 + 4 < f1) gload(f0 + 4);
        for (int fb = f0;; fb += 8) {
            compute(fb, IntC<0>{}, Ic);
            if (fb + 4 >= f1) break;
            stage(IntC<1>{});
            __syncthreads();
            if (fb + 8 < f1) gload(fb + 8);
            compute(fb + 4, IntC<1>{}, Ic);
            if (fb + 8 >= f1) break;
            stage(IntC<0>{});
            __syncthreads();
            if (fb + 12 < f1) gload(fb + 12);
        }
    };
    const bool inner = !ragd && ((f1 - f0) & 3) == 0 && bx * kWlTW + kWlTW <= d.cols && by * kWlTH + kWlTH <= d.rows;   // (uniform)
    if (inner) run(IntC<1>{});
    else run(IntC<0>{});
}

// One-channel images: the scheme of k_warp_affine_bgr with 2-byte tap pairs.  One thread per output column and kWarpRows
// rows; a wave whose taps all lie inside the source fetches each tap pair as the ALIGNED 8 bytes that contain it (one dwordx2
// per source row: 16 tap loads in flight per lane), shifts it into place with v_alignbyte, lerps top and bottom rows as one
// packed-f32 pair, and after a quad transpose every lane stores the 4 pixels of one row as a dword.  Waves that touch the
// border run warp_px<1> per pixel (same f32 operations, same order).
__device__ __forceinline__ void warp_gray_frame(const View& s, const View& d, const Affine& A, const int frame, const int x, const int ybase)
{
    const uint8_t* sf = s.p + (size_t)frame * s.fstride;
    uint8_t* dfr = d.p + (size_t)frame * d.fstride;
    const int lane = threadIdx.x & 63;
    const float fxx = (float)min(x, d.cols - 1);
    const float fy0 = (float)min(ybase, d.rows - 1), fy1 = (float)min(ybase + kWarpRows - 1, d.rows - 1);
    const float xa = fmaf(A.m[0], fxx, fmaf(A.m[1], fy0, A.m[2])), xb = fmaf(A.m[0], fxx, fmaf(A.m[1], fy1, A.m[2]));
    const float ya = fmaf(A.m[3], fxx, fmaf(A.m[4], fy0, A.m[5])), yb = fmaf(A.m[3], fxx, fmaf(A.m[4], fy1, A.m[5]));
    // (sx, sy are monotonic in the row index for a fixed lane: the first and the last row bound all eight; NaN -> false)
    const bool inter = fminf(xa, xb) >= 0.0f && fmaxf(xa, xb) < (float)(s.cols - 7) && fminf(ya, yb) >= 0.0f && fmaxf(ya, yb) < (float)(s.rows - 1);
    const bool small = s.step < (1u << 24) && s.rows < (1 << 24) && (unsigned long long)s.rows * s.step < (1ull << 32);
    // (source rows of any alignment: tap windows are the 8 ALIGNED bytes that contain each tap pair; the frame base is aligned down)
    const unsigned fmis = (unsigned)((uintptr_t)sf & 3);
    const uint8_t* sfa = sf - fmis;
    const int xq = x & ~3, yi = ybase + (lane & 3);
    uint32_t px[kWarpRows];
    if (small && __all(inter)) {
        const unsigned sstep = (unsigned)s.step;
        struct U2 { uint32_t a, b; };
        U2 ta[kWarpRows], tb[kWarpRows];
        f2 fxy[kWarpRows];
        unsigned sh[kWarpRows], shb[kWarpRows];
#pragma unroll
        for (int r = 0; r < kWarpRows; ++r) {
            const float fyy = (float)min(ybase + r, d.rows - 1);
            const f2 sxy = __builtin_elementwise_fma(f2{A.m[0], A.m[3]}, f2{fxx, fxx}, __builtin_elementwise_fma(f2{A.m[1], A.m[4]}, f2{fyy, fyy}, f2{A.m[2], A.m[5]}));
            fxy[r] = f2{__builtin_amdgcn_fractf(sxy.x), __builtin_amdgcn_fractf(sxy.y)};   // sx, sy >= 0: exact sx - floor(sx)
            const unsigned off = __umul24((unsigned)(int)sxy.y, sstep) + (unsigned)(int)sxy.x + fmis, offb = off + sstep;
            sh[r] = off & 3u;
            shb[r] = offb & 3u;
            ta[r] = *(const U2*)(sfa + (off & ~3u));
            tb[r] = *(const U2*)(sfa + (offb & ~3u));
        }
#pragma unroll
        for (int r = 0; r < kWarpRows; ++r) {
            const uint32_t a = __builtin_amdgcn_alignbyte(ta[r].b, ta[r].a, sh[r]), b = __builtin_amdgcn_alignbyte(tb[r].b, tb[r].a, shb[r]);
            // {top, bottom} as one packed pair: fma(fx, p01 - p00, p00), then v = fma(fy, bot - top, top), floor(v + 0.5)
            const f2 p0 = {ub<0>(a), ub<0>(b)}, p1 = {ub<1>(a), ub<1>(b)};
            const f2 tb2 = pk_fma_bc<0>(fxy[r], p1 - p0, p0);
            px[r] = trunc_u32(fmaf(fxy[r].y, tb2.y - tb2.x, tb2.x) + 0.5f);   // interior: an integer in [0, 255]
        }
    } else {
#pragma unroll 1
        for (int r = 0; r < kWarpRows; ++r) {
            uint8_t o1[1];
            warp_px<1>(sf, s, A, fxx, (float)min(ybase + r, d.rows - 1), o1);
            px[r] = o1[0];
        }
    }
#pragma unroll
    for (int h = 0; h < kWarpRows / 4; ++h) {
        uint32_t t[4] = {px[4 * h], px[4 * h + 1], px[4 * h + 2], px[4 * h + 3]};
        quad_transpose4(t, lane);
        if (xq < d.cols && yi + 4 * h < d.rows) {   // any destination alignment; the row's last quad of a ragged width byte by byte
            typedef uint32_t u1m __attribute__((aligned(1)));
            uint8_t* q = dfr + (size_t)(yi + 4 * h) * d.step + xq;
            const uint32_t v = t[0] | (t[1] << 8) | (t[2] << 16) | (t[3] << 24);
            if (d.cols - xq >= 4) *(u1m*)q = v;
            else
                for (int j = 0; j < d.cols - xq; ++j) q[j] = (uint8_t)(v >> (8 * j));
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_warp_affine_gray(View s, View d, Affine A)
{
    warp_gray_frame(s, d, A, (int)blockIdx.z, (int)(blockIdx.x * kBlock + threadIdx.x), (int)blockIdx.y * kWarpRows);
}

// ---- bilinear resize, BGR, any scale: the register scheme of k_warp_affine_bgr ------------------------------------------
// One thread per output column and kRszRows consecutive output rows: x0 and fx are computed once, every row costs two
// aligned 12-byte tap windows (all in flight together), packed-f32 lerps, and four lanes share a 12-byte store.  A wave
// whose lanes all have x0 <= cols-4 (the whole tap window inside the row) takes this path; the few waves at the right
// edge run resize_px<3> per pixel.  Same f32 operations in the same order as resize_px<3>.
constexpr int kRszRows = 4;

__global__ __launch_bounds__(kBlock) void k_resize_bgr(View s, View d, float scx, float scy)
{
    const uint8_t* sf = s.p + (size_t)blockIdx.z * s.fstride;
    uint8_t* dfr = d.p + (size_t)blockIdx.z * d.fstride;
    const int x = blockIdx.x * kBlock + threadIdx.x;   // d.cols % 4 == 0: quads never straddle the row end
    const int xq = min(x, d.cols - 1);
    const int ybase = blockIdx.y * kRszRows;
    float sx = ((float)xq + 0.5f) * scx - 0.5f;
    sx = sx < 0.0f ? 0.0f : sx;
    sx = sx > (float)(s.cols - 1) ? (float)(s.cols - 1) : sx;
    const float x0f = floorf(sx);
    const int x0 = (int)x0f;
    const float fx = sx - x0f;
    struct U3 { uint32_t a, b, c; };
    const bool small = s.step < (1u << 24) && s.rows < (1 << 24) && (unsigned long long)s.rows * s.step < (1ull << 32);
    // Source rows of any alignment (an odd width of a packed image): the tap window is the 12 ALIGNED bytes that contain the 6 tap
    // bytes; where they start depends on the row's own misalignment (a scalar per row) and the column's.  The frame base is
    // aligned down so that offsets stay non-negative.
    const unsigned fmis = (unsigned)((uintptr_t)sf & 3);
    const uint8_t* sfa = sf - fmis;
    if (small && __all(x0 <= s.cols - 4)) {
        U3 ta[kRszRows], tb[kRszRows];
        float fy[kRszRows];
        unsigned sha[kRszRows], shb[kRszRows];
        const unsigned xo = 3u * (unsigned)x0 + fmis;
#pragma unroll
        for (int r = 0; r < kRszRows; ++r) {
            int y0, y1;
            resize_row(s, scy, min(ybase + r, d.rows - 1), y0, y1, fy[r]);
            const unsigned oa = __umul24((unsigned)y0, (unsigned)s.step) + xo, ob = __umul24((unsigned)y1, (unsigned)s.step) + xo;
            sha[r] = oa & 3u;
            shb[r] = ob & 3u;
            ta[r] = *(const U3*)(sfa + (oa & ~3u));
            tb[r] = *(const U3*)(sfa + (ob & ~3u));
        }
        uint32_t px[kRszRows];
#pragma unroll
        for (int r = 0; r < kRszRows; ++r) {
            const uint32_t alo = __builtin_amdgcn_alignbyte(ta[r].b, ta[r].a, sha[r]), ahi = __builtin_amdgcn_alignbyte(ta[r].c, ta[r].b, sha[r]);
            const uint32_t blo = __builtin_amdgcn_alignbyte(tb[r].b, tb[r].a, shb[r]), bhi = __builtin_amdgcn_alignbyte(tb[r].c, tb[r].b, shb[r]);
            px[r] = bilerp_bgr<true>(alo, ahi, blo, bhi, f2{fx, fy[r]});
        }
        // quad transpose: lane 4q+i stores pixels 4q..4q+3 of row ybase+i -- one full-wave store for the four rows
        static_assert(kRszRows == 4, "one quad transpose per thread");
        const int lane = threadIdx.x & 63, xs = x & ~3, yi = ybase + (lane & 3);
        quad_transpose4(px, lane);
        if (xs < d.cols && yi < d.rows) {
            // (one store instruction for aligned and byte-aligned destination rows; the row's last quad of a width that is not a
            //  multiple of 4 goes out byte by byte)
            typedef uint32_t u3m __attribute__((ext_vector_type(3), aligned(1)));
            const uint32_t va = __builtin_amdgcn_perm(px[1], px[0], 0x04020100u), vb = __builtin_amdgcn_perm(px[2], px[1], 0x05040201u),
                           vc = __builtin_amdgcn_perm(px[3], px[2], 0x06050402u);
            uint8_t* q = dfr + (size_t)yi * d.step + (size_t)xs * 3;
            if (d.cols - xs < 4) store_quad_ragged(q, va, vb, vc, d.cols - xs);
            else *(u3m*)q = u3m{va, vb, vc};
        }
        return;
    }
#pragma unroll 1
    for (int r = 0; r < kRszRows; ++r) {
        if (x >= d.cols || ybase + r >= d.rows) continue;
        int y0, y1;
        float fy;
        resize_row(s, scy, ybase + r, y0, y1, fy);
        uint8_t o[3];
        resize_px<3>(sf + (size_t)y0 * s.step, sf + (size_t)y1 * s.step, s, scx, fy, x, o);
        uint8_t* q = dfr + (size_t)(ybase + r) * d.step + (size_t)x * 3;
        q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
    }
}

// One-channel images: k_resize_bgr's scheme with 2-byte tap pairs (one aligned dwordx2 per source row).
__global__ __launch_bounds__(kBlock) void k_resize_gray(View s, View d, float scx, float scy)
{
    const uint8_t* sf = s.p + (size_t)blockIdx.z * s.fstride;
    uint8_t* dfr = d.p + (size_t)blockIdx.z * d.fstride;
    const int x = blockIdx.x * kBlock + threadIdx.x;   // d.cols % 4 == 0: quads never straddle the row end
    const int xq = min(x, d.cols - 1);
    const int ybase = blockIdx.y * kRszRows;
    float sx = ((float)xq + 0.5f) * scx - 0.5f;
    sx = sx < 0.0f ? 0.0f : sx;
    sx = sx > (float)(s.cols - 1) ? (float)(s.cols - 1) : sx;
    const float x0f = floorf(sx);
    const int x0 = (int)x0f;
    const float fx = sx - x0f;
    const bool small = s.step < (1u << 24) && s.rows < (1 << 24) && (unsigned long long)s.rows * s.step < (1ull << 32);
    // (source rows of any alignment: the tap window is the 8 ALIGNED bytes that contain the tap pair, as in k_resize_bgr)
    const unsigned fmis = (unsigned)((uintptr_t)sf & 3);
    const uint8_t* sfa = sf - fmis;
    if (small && __all(x0 <= s.cols - 8)) {   // the aligned 8-byte window ends inside the row
        struct U2 { uint32_t a, b; };
        U2 ta[kRszRows], tb[kRszRows];
        float fy[kRszRows];
        unsigned sha[kRszRows], shb[kRszRows];
        const unsigned xo = (unsigned)x0 + fmis;
#pragma unroll
        for (int r = 0; r < kRszRows; ++r) {
            int y0, y1;
            resize_row(s, scy, min(ybase + r, d.rows - 1), y0, y1, fy[r]);
            const unsigned oa = __umul24((unsigned)y0, (unsigned)s.step) + xo, ob = __umul24((unsigned)y1, (unsigned)s.step) + xo;
            sha[r] = oa & 3u;
            shb[r] = ob & 3u;
            ta[r] = *(const U2*)(sfa + (oa & ~3u));
            tb[r] = *(const U2*)(sfa + (ob & ~3u));
        }
        uint32_t px[kRszRows];
#pragma unroll
        for (int r = 0; r < kRszRows; ++r) {
            const uint32_t a = __builtin_amdgcn_alignbyte(ta[r].b, ta[r].a, sha[r]), b = __builtin_amdgcn_alignbyte(tb[r].b, tb[r].a, shb[r]);
            const f2 p0 = {ub<0>(a), ub<0>(b)}, p1 = {ub<1>(a), ub<1>(b)};
            const f2 tb2 = pk_fma_bc<0>(f2{fx, fy[r]}, p1 - p0, p0);          // {top, bottom}: fma(fx, p01 - p00, p00)
            px[r] = trunc_u32(fmaf(fy[r], tb2.y - tb2.x, tb2.x) + 0.5f);   // an integer in [0, 255]
        }
        static_assert(kRszRows == 4, "one quad transpose per thread");
        const int lane = threadIdx.x & 63, xs = x & ~3, yi = ybase + (lane & 3);
        quad_transpose4(px, lane);
        if (xs < d.cols && yi < d.rows) {   // any destination alignment; the row's last quad of a ragged width byte by byte
            typedef uint32_t u1m __attribute__((aligned(1)));
            uint8_t* q = dfr + (size_t)yi * d.step + xs;
            const uint32_t v = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
            if (d.cols - xs >= 4) *(u1m*)q = v;
            else
                for (int j = 0; j < d.cols - xs; ++j) q[j] = (uint8_t)(v >> (8 * j));
        }
        return;
    }
#pragma unroll 1
    for (int r = 0; r < kRszRows; ++r) {
        if (x >= d.cols || ybase + r >= d.rows) continue;
        int y0, y1;
        float fy;
        resize_row(s, scy, ybase + r, y0, y1, fy);
        uint8_t o[1];
        resize_px<1>(sf + (size_t)y0 * s.step, sf + (size_t)y1 * s.step, s, scx, fy, x, o);
        dfr[(size_t)(ybase + r) * d.step + x] = o[0];
    }
}

// ---- RCV_32F images (SURVEY.md 8-A "warp_affine (u8/f32 ...)"; the cornerHarris response map) --------------------------------
// The same sampling rules and the same f32 operations in the same order as the u8 kernels (top = fma(fx, p01 - p00, p00), bot
// likewise, v = fma(fy, bot - top, top); oracle: orc_resize_f32 / orc_warp_affine_f32) with f32 taps and the unrounded v as the
// result.  One thread per output pixel; HBM-bound at 4 B read + 4 B written per sample, nothing to fuse.
template <int CH>
__global__ __launch_bounds__(kBlock) void k_resize_f32(View s, View d, float scx, float scy)
{
    const int y = blockIdx.y;
    const uint8_t* sf = s.p + (size_t)blockIdx.z * s.fstride;
    float* drow = (float*)(d.p + (size_t)blockIdx.z * d.fstride + (size_t)y * d.step);
    int y0, y1;
    float fy;
    resize_row(s, scy, y, y0, y1, fy);
    const float* ra = (const float*)(sf + (size_t)y0 * s.step);
    const float* rb = (const float*)(sf + (size_t)y1 * s.step);
    for (int x = blockIdx.x * kBlock + threadIdx.x; x < d.cols; x += gridDim.x * kBlock) {
        float sx = ((float)x + 0.5f) * scx - 0.5f;
        sx = sx < 0.0f ? 0.0f : sx;
        sx = sx > (float)(s.cols - 1) ? (float)(s.cols - 1) : sx;
        const int x0 = (int)floorf(sx);
        const float fx = sx - (float)x0;
        const int x1 = x0 + 1 < s.cols ? x0 + 1 : s.cols - 1;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float p00 = ra[(size_t)x0 * CH + c], p01 = ra[(size_t)x1 * CH + c];
            const float p10 = rb[(size_t)x0 * CH + c], p11 = rb[(size_t)x1 * CH + c];
            const float top = fmaf(fx, p01 - p00, p00);
            const float bot = fmaf(fx, p11 - p10, p10);
            drow[(size_t)x * CH + c] = fmaf(fy, bot - top, top);
        }
    }
}

// One-channel RCV_32F resize, four output rows per thread: the column arithmetic once, the eight tap PAIRS {p(x0), p(x0 + 1)} of the four
// rows as eight 8-byte loads issued before the first lerp (k_resize_f32<1> issues 4 scalar loads for one pixel).  The last column's
// pair {p(cols-1), p(cols-1)} is read as the pair one column to the left and its upper half used twice.  Same three fmaf per sample.
constexpr int kRszF32Rows = 4;
__global__ __launch_bounds__(kBlock) void k_resize_f32_rows(View s, View d, float scx, float scy)
{
    const int ybase = blockIdx.y * kRszF32Rows;
    const uint8_t* sf = s.p + (size_t)blockIdx.z * s.fstride;
    uint8_t* df = d.p + (size_t)blockIdx.z * d.fstride;
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    for (int x = blockIdx.x * kBlock + threadIdx.x; x < d.cols; x += gridDim.x * kBlock) {
        float sx = ((float)x + 0.5f) * scx - 0.5f;
        sx = sx < 0.0f ? 0.0f : sx;
        sx = sx > (float)(s.cols - 1) ? (float)(s.cols - 1) : sx;
        const int x0 = (int)floorf(sx);
        const float fx = sx - (float)x0;
        const int xp = min(x0, s.cols - 2);   // (host: s.cols >= 2)
        const bool last = x0 > xp;
        f2u ta[kRszF32Rows], tb[kRszF32Rows];
        float fy[kRszF32Rows];
#pragma unroll
        for (int r = 0; r < kRszF32Rows; ++r) {
            int y0, y1;
            resize_row(s, scy, min(ybase + r, d.rows - 1), y0, y1, fy[r]);
            ta[r] = *(const f2u*)(sf + (size_t)y0 * s.step + 4 * (size_t)xp);
            tb[r] = *(const f2u*)(sf + (size_t)y1 * s.step + 4 * (size_t)xp);
        }
#pragma unroll
        for (int r = 0; r < kRszF32Rows; ++r) {
            const float p00 = last ? ta[r].y : ta[r].x, p01 = ta[r].y, p10 = last ? tb[r].y : tb[r].x, p11 = tb[r].y;
            const float top = fmaf(fx, p01 - p00, p00);
            const float bot = fmaf(fx, p11 - p10, p10);
            if (ybase + r < d.rows) *(float*)(df + (size_t)(ybase + r) * d.step + 4 * (size_t)x) = fmaf(fy[r], bot - top, top);
        }
    }
}

template <int CH>
__global__ __launch_bounds__(kBlock) void k_warp_affine_f32(View s, View d, Affine A)
{
    const int y = blockIdx.y;
    const uint8_t* sf = s.p + (size_t)blockIdx.z * s.fstride;
    float* drow = (float*)(d.p + (size_t)blockIdx.z * d.fstride + (size_t)y * d.step);
    const float fyy = (float)y;
    for (int x = blockIdx.x * kBlock + threadIdx.x; x < d.cols; x += gridDim.x * kBlock) {
        const float fxx = (float)x;
        const float sx = fmaf(A.m[0], fxx, fmaf(A.m[1], fyy, A.m[2]));
        const float sy = fmaf(A.m[3], fxx, fmaf(A.m[4], fyy, A.m[5]));
        float* o = drow + (size_t)x * CH;
        if (!(sx > -1.0f && sx < (float)s.cols && sy > -1.0f && sy < (float)s.rows)) {
#pragma unroll
            for (int c = 0; c < CH; ++c) o[c] = 0.0f;
            continue;
        }
        const float x0f = floorf(sx), y0f = floorf(sy);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const float fx = sx - x0f, fy = sy - y0f;
        const bool vx0 = x0 >= 0, vx1 = x0 + 1 < s.cols, vy0 = y0 >= 0, vy1 = y0 + 1 < s.rows;
        const float* ra = (const float*)(sf + (size_t)(vy0 ? y0 : 0) * s.step);
        const float* rb = (const float*)(sf + (size_t)(vy1 ? y0 + 1 : 0) * s.step);
        const size_t xa = (size_t)(vx0 ? x0 : 0) * CH, xb = (size_t)(vx1 ? x0 + 1 : 0) * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float p00 = (vx0 && vy0) ? ra[xa + c] : 0.0f, p01 = (vx1 && vy0) ? ra[xb + c] : 0.0f;
            const float p10 = (vx0 && vy1) ? rb[xa + c] : 0.0f, p11 = (vx1 && vy1) ? rb[xb + c] : 0.0f;
            const float top = fmaf(fx, p01 - p00, p00);
            const float bot = fmaf(fx, p11 - p10, p10);
            o[c] = fmaf(fy, bot - top, top);
        }
    }
}

// One-channel RCV_32F warpAffine through an LDS-staged source patch (round 3): k_warp_affine_lds's tiles, patch geometry, pitch planner
// and double buffer with the patch held as the f32 samples themselves -- a chunk is 16 source bytes = one ds_write_b128, a tap
// pair one ds_read2_b32, no conversion anywhere; a thread owns one column of 8 rows, so a wave's stores are whole 256-byte row
// pieces.  Interior tiles only (every tap inside the source): border tiles and tiles outside run the per-pixel code of
// k_warp_affine_f32 -- the same three fmaf per sample in the same order either way.
__device__ __forceinline__ float warp_px_f32_1(const uint8_t* sf, const View& s, const Affine& A, float fxx, float fyy)
{
    const float sx = fmaf(A.m[0], fxx, fmaf(A.m[1], fyy, A.m[2]));
    const float sy = fmaf(A.m[3], fxx, fmaf(A.m[4], fyy, A.m[5]));
    if (!(sx > -1.0f && sx < (float)s.cols && sy > -1.0f && sy < (float)s.rows)) return 0.0f;
    const float x0f = floorf(sx), y0f = floorf(sy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float fx = sx - x0f, fy = sy - y0f;
    const bool vx0 = x0 >= 0, vx1 = x0 + 1 < s.cols, vy0 = y0 >= 0, vy1 = y0 + 1 < s.rows;
    const float* ra = (const float*)(sf + (size_t)(vy0 ? y0 : 0) * s.step);
    const float* rb = (const float*)(sf + (size_t)(vy1 ? y0 + 1 : 0) * s.step);
    const size_t xa = (size_t)(vx0 ? x0 : 0), xb = (size_t)(vx1 ? x0 + 1 : 0);
    const float p00 = (vx0 && vy0) ? ra[xa] : 0.0f, p01 = (vx1 && vy0) ? ra[xb] : 0.0f;
    const float p10 = (vx0 && vy1) ? rb[xa] : 0.0f, p11 = (vx1 && vy1) ? rb[xb] : 0.0f;
    const float top = fmaf(fx, p01 - p00, p00);
    const float bot = fmaf(fx, p11 - p10, p10);
    return fmaf(fy, bot - top, top);
}

__global__ __launch_bounds__(kBlock) void k_warp_f32_lds(View s, View d, Affine A, int fpg, int pitch, int prow, int cpr, int gx, int gy, int ntiles,
                                                         int tiles_per_xcd, int strip)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t wf_lds[];
    int bx, by, bz;
    if (!wl_tile(tiles_per_xcd, strip, gx, gy, ntiles, bx, by, bz)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = bx * kWlTW + lane, ybase = by * kWlTH + wave * kWarpRows;
    const int f0 = bz * fpg, f1 = min(f0 + fpg, d.n);
    const float cx0 = (float)min(bx * kWlTW, d.cols - 1), cx1 = (float)min(bx * kWlTW + kWlTW - 1, d.cols - 1);
    const float cy0 = (float)min(by * kWlTH, d.rows - 1), cy1 = (float)min(by * kWlTH + kWlTH - 1, d.rows - 1);
    const float t0 = fmaf(A.m[1], cy0, A.m[2]), t1 = fmaf(A.m[1], cy1, A.m[2]), u0 = fmaf(A.m[4], cy0, A.m[5]), u1 = fmaf(A.m[4], cy1, A.m[5]);
    const float xa = fmaf(A.m[0], cx0, t0), xb = fmaf(A.m[0], cx1, t0), xc = fmaf(A.m[0], cx0, t1), xd = fmaf(A.m[0], cx1, t1);
    const float ya = fmaf(A.m[3], cx0, u0), yb = fmaf(A.m[3], cx1, u0), yc = fmaf(A.m[3], cx0, u1), yd = fmaf(A.m[3], cx1, u1);
    const float xmin = fminf(fminf(xa, xb), fminf(xc, xd)), xmax = fmaxf(fmaxf(xa, xb), fmaxf(xc, xd));
    const float ymin = fminf(fminf(ya, yb), fminf(yc, yd)), ymax = fmaxf(fmaxf(ya, yb), fmaxf(yc, yd));
    // (inside / border tiles: as in k_warp_affine_lds -- a patch that leaves a source whose width is a multiple of 4 is staged with
    //  zeros for the chunks outside: a tap of 0.0f is the specification's border, orc_warp_affine_f32)
    const bool inside = xmin >= 0.0f && xmax < (float)(s.cols - 1) && ymin >= 0.0f && ymax < (float)(s.rows - 2);   // NaN -> false
    const bool edge_ok = (s.cols & 3) == 0 && xmin > -1.0e6f && xmax < 1.0e6f && ymin > -1.0e6f && ymax < 1.0e6f;   // NaN -> false
    bool ok = (inside || edge_ok) && s.step >= 64 && s.step < (1u << 24) && s.rows < (1 << 24) && (unsigned long long)s.rows * s.step < (1ull << 32);
    int ix0 = 0, iy0 = 0;
    if (ok) {
        ix0 = (int)floorf(xmin) & ~3;
        iy0 = (int)floorf(ymin);
        ok = (int)floorf(xmax) + 2 - ix0 <= 4 * cpr && (int)floorf(ymax) + 2 - iy0 <= prow;
    }
    const bool border = __builtin_amdgcn_readfirstlane((int)!inside) != 0;   // (uniform)
    if (!__builtin_amdgcn_readfirstlane((int)ok)) {
        if (x < d.cols)
            for (int f = f0; f < f1; ++f) {
                const uint8_t* sf = s.p + (size_t)f * s.fstride;
#pragma unroll 1
                for (int r = 0; r < kWarpRows; ++r)
                    if (ybase + r < d.rows) *(float*)(d.p + (size_t)f * d.fstride + (size_t)(ybase + r) * d.step + 4 * (size_t)x) = warp_px_f32_1(sf, s, A, (float)x, (float)(ybase + r));
            }
        return;
    }
    ix0 = __builtin_amdgcn_readfirstlane(ix0);
    iy0 = __builtin_amdgcn_readfirstlane(iy0);
    const float fxx = (float)min(x, d.cols - 1);
    typedef __attribute__((address_space(3))) uint8_t* lp;
    typedef const __attribute__((address_space(3))) float* lcf;
    const unsigned lds0 = (unsigned)(uintptr_t)(lp)wf_lds;   // (the LDS base folded into every precomputed offset)
    const unsigned bufbytes = (unsigned)(pitch * prow);
    f2 fxy[kWarpRows];
    unsigned la[2][kWarpRows];   // (per patch buffer: the frame loop is unrolled by two)
#pragma unroll
    for (int r = 0; r < kWarpRows; ++r) {
        const float fyy = (float)min(ybase + r, d.rows - 1);
        const f2 sxy = __builtin_elementwise_fma(f2{A.m[0], A.m[3]}, f2{fxx, fxx}, __builtin_elementwise_fma(f2{A.m[1], A.m[4]}, f2{fyy, fyy}, f2{A.m[2], A.m[5]}));
        const f2 fl = __builtin_elementwise_floor(sxy);
        fxy[r] = sxy - fl;   // (the specification's sx - floor(sx): exact; also left of / above zero)
        la[0][r] = lds0 + __umul24((unsigned)((int)fl.y - iy0), (unsigned)pitch) + 4u * (unsigned)((int)fl.x - ix0);
        la[1][r] = la[0][r] + bufbytes;
        // A sample the specification sets to 0 without reading a tap (sx <= -1, sx >= cols, likewise sy): zero taps give the same 0
        // -- except at sx == -1 (sy == -1) exactly, where the tap at column (row) 0 is data with weight 0, and 0 * inf is NaN.  Such
        // pixels read their four taps from a zeroed area behind the two patch buffers (pitch + 16 bytes: the host adds them).
        if (border && !(sxy.x > -1.0f && sxy.x < (float)s.cols && sxy.y > -1.0f && sxy.y < (float)s.rows)) la[0][r] = la[1][r] = lds0 + 2u * bufbytes;
        asm volatile("" : "+v"(la[0][r]), "+v"(la[1][r]));
    }
    // ---- staging plan: chunk c = 4 samples = 16 source bytes (rows are 4-byte aligned) -> 16 LDS bytes ----
    const int nchunks = prow * cpr;
    const unsigned cpr_magic = (1u << 20) / (unsigned)cpr + 1u;   // c / cpr == (c * cpr_magic) >> 20 for every c < 1536 and cpr <= 755 (here cpr * prow <= 1536, prow >= 3): one division instead of one per chunk slot
    const unsigned frame_lim = (unsigned)(s.rows - 1) * (unsigned)s.step + 4u * (unsigned)s.cols - 16u;
    unsigned goff[kWlMaxG], loff[2][kWlMaxG];
    unsigned gzero = 0;   // bit g: chunk slot g lies outside the source (border tiles: staged as zeros)
    bool gval[kWlMaxG];
#pragma unroll
    for (int g = 0; g < kWlMaxG; ++g) {
        const int c = (int)threadIdx.x + kBlock * g;
        gval[g] = c < nchunks;
        const int row = gval[g] ? (int)(((unsigned)c * cpr_magic) >> 20) : 0, col = gval[g] ? c - row * cpr : 0;
        const int ry = iy0 + row, px = ix0 + 4 * col;   // (border tiles: either may lie outside the source)
        gzero |= (unsigned)(ry < 0 || ry >= s.rows || px < 0 || px + 4 > s.cols) << g;
        goff[g] = min(__umul24((unsigned)min(max(ry, 0), s.rows - 1), (unsigned)s.step) + 4u * (unsigned)max(px, 0), frame_lim);
        loff[0][g] = lds0 + (unsigned)(row * pitch + 16 * col);
        loff[1][g] = loff[0][g] + bufbytes;
        asm volatile("" : "+v"(loff[0][g]), "+v"(loff[1][g]));
    }
    const int ng = (nchunks + kBlock - 1) / kBlock;
    typedef const __attribute__((address_space(1))) uint8_t* cgp;
    typedef __attribute__((address_space(1))) uint8_t* gp;
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    typedef uint32_t u4a __attribute__((ext_vector_type(4), aligned(4)));
    typedef __attribute__((address_space(1))) u4a gU4a;
    typedef __attribute__((address_space(3))) u4v* lU4;
    u4v G[kWlMaxG];
    auto gload = [&](int f) {
        cgp sf = (cgp)(s.p + (size_t)f * s.fstride);
        asm("" : "+s"(sf));
#pragma unroll
        for (int g = 0; g < kWlMaxG; ++g)
            if (g < ng) {   // uniform
                asm volatile("" : "+v"(goff[g]));   // (scalar frame base + 32-bit thread offset in one instruction: see k_warp_affine_lds)
                const u4a t = *(const gU4a*)(sf + goff[g]);
                G[g] = u4v{t.x, t.y, t.z, t.w};
            }
    };
    const bool inx = x < d.cols;
    unsigned so[kWarpRows];   // in-frame offset of the thread's pixel in each of its rows (INNER tiles: d.rows * d.step < 2^32 checked below)
#pragma unroll
    for (int r = 0; r < kWarpRows; ++r) so[r] = (unsigned)(ybase + r) * (unsigned)d.step + 4u * (unsigned)x;
    auto stage = [&](auto Bc) {
        constexpr int B = decltype(Bc)::value;
        if (border) {   // (uniform; kept a branch by the volatile asm)
            asm volatile("; tile whose patch leaves the source: chunks outside are zeros");
#pragma unroll
            for (int g = 0; g < kWlMaxG; ++g)
                if ((gzero >> g) & 1u) G[g] = u4v{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int g = 0; g < kWlMaxG; ++g)
            if (gval[g]) *(lU4)(uintptr_t)(loff[B][g]) = G[g];
    };
    auto compute = [&](const int f, auto Bc, auto Ic) {
        constexpr int B = decltype(Bc)::value;
        constexpr bool INNER = decltype(Ic)::value != 0;
        uint8_t* dcol = d.p + (size_t)f * d.fstride + 4 * (size_t)x;
        gp dfr = (gp)(d.p + (size_t)f * d.fstride);
        asm("" : "+s"(dfr));
#pragma unroll
        for (int r = 0; r < kWarpRows; ++r) {
            const lcf pa = (lcf)(uintptr_t)(la[B][r]), pb = (lcf)(uintptr_t)(la[B][r] + (unsigned)pitch);
            const f2 p0 = {pa[0], pb[0]}, p1 = {pa[1], pb[1]};
            const f2 tb = pk_fma_bc<0>(fxy[r], p1 - p0, p0);                     // {top, bottom}: fma(fx, p01 - p00, p00), fma(fx, p11 - p10, p10)
            const float v = fmaf(fxy[r].y, tb.y - tb.x, tb.x);
            if constexpr (INNER) {
                asm volatile("" : "+v"(so[r]));
                *(__attribute__((address_space(1))) float*)(dfr + so[r]) = v;
            } else if (inx && ybase + r < d.rows) *(float*)(dcol + (size_t)(ybase + r) * d.step) = v;
        }
    };
    // the frame loop, rotated as in k_warp_affine_lds: compute(f) | stage(f + 1) | barrier | loads of f + 2 -- the wait in
    // front of stage() covers the chunk loads only, not the eight stores compute() has just issued (INNER tiles: no branch)
    if (border)   // (the zeroed area: written once, before the first barrier)
        for (unsigned o = 4u * threadIdx.x; o < (unsigned)pitch + 16u; o += 4u * kBlock) *(__attribute__((address_space(3))) uint32_t*)(uintptr_t)(lds0 + 2u * bufbytes + o) = 0u;
    auto run = [&](auto Ic) {
        gload(f0);
        stage(IntC<0>{});
        __syncthreads();
        if (f0 + 1 < f1) gload(f0 + 1);
        for (int f = f0;; f += 2) {
            compute(f, IntC<0>{}, Ic);
            if (f + 1 >= f1) break;
            stage(IntC<1>{});
            __syncthreads();
            if (f + 2 < f1) gload(f + 2);
            compute(f + 1, IntC<1>{}, Ic);
            if (f + 2 >= f1) break;
            stage(IntC<0>{});
            __syncthreads();
            if (f + 3 < f1) gload(f + 3);
        }
    };
    const bool inner = bx * kWlTW + kWlTW <= d.cols && by * kWlTH + kWlTH <= d.rows && (unsigned long long)d.rows * d.step < (1ull << 32);   // (uniform)
    if (inner) run(IntC<1>{});
    else run(IntC<0>{});
}

// views of an RCV_32F source / destination pair: 1, 3 or 4 channels, 4-byte aligned rows and frames
int check_geom_f32(const rcv_batch* src, rcv_batch* dst, View* s, View* d)
{
    if (!src || !dst) return RCV_ERR_ARG;
    RCV_TRY(rcv_view_batch(src, RCV_32F, s));
    RCV_TRY(rcv_view_batch(dst, RCV_32F, d));
    if (s->ch != d->ch || s->n != d->n) return RCV_ERR_ARG;
    if (s->ch != 1 && s->ch != 3 && s->ch != 4) return RCV_ERR_UNSUPPORTED;
    if (d->rows > 65535 || d->n > 65535) return RCV_ERR_UNSUPPORTED;
    for (const View* v : {s, d})
        if ((uintptr_t)v->p % 4 || v->step % 4 || (v->n > 1 && v->fstride % 4)) return RCV_ERR_ARG;   // f32 samples are 4-byte aligned
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
    if (src && dst && src->frame0.depth == RCV_32F) {   // f32 images: the unrounded interpolated value
        RCV_TRY(check_geom_f32(src, dst, &s, &d));
        if (d.rows == 0 || d.cols == 0 || d.n == 0) return RCV_OK;
        if (s.rows == 0 || s.cols == 0) return RCV_ERR_ARG;
        const float fscx = (float)s.cols / (float)d.cols, fscy = (float)s.rows / (float)d.rows;
        if (s.ch == 1 && s.cols >= 2 && fscy < 2.5f) {   // (8 x 8K -> 5K: 0.41 -> 0.26 ms; a 4x down-scale reads rows 4 apart: 0.108 -> 0.115, stays on the one-row kernel)
            dim3 g = px_grid(d);
            g.y = (unsigned)((d.rows + kRszF32Rows - 1) / kRszF32Rows);
            RCV_LAUNCH(k_resize_f32_rows, g, dim3(kBlock), 0, ctx->stream, s, d, fscx, fscy);
        } else if (s.ch == 1) RCV_LAUNCH(k_resize_f32<1>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, fscx, fscy);
        else if (s.ch == 3) RCV_LAUNCH(k_resize_f32<3>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, fscx, fscy);
        else RCV_LAUNCH(k_resize_f32<4>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, fscx, fscy);
        return rcv_launch_check(ctx);
    }
    RCV_TRY(check_geom(src, dst, &s, &d));
    if (d.rows == 0 || d.cols == 0 || d.n == 0) return RCV_OK;
    if (s.rows == 0 || s.cols == 0) return RCV_ERR_ARG;
    for (int S = 2; S <= 4; S += 2) {
        if (s.ch == 3 && s.cols == S * d.cols && s.rows == S * d.rows && d.cols % 4 == 0 && (uintptr_t)s.p % 16 == 0 &&
            s.step % 16 == 0 && s.fstride % 16 == 0 && (uintptr_t)d.p % 4 == 0 && d.step % 4 == 0 && d.fstride % 4 == 0) {
            dim3 grid((unsigned)((d.cols / 4 + kBlock - 1) / kBlock), d.rows, d.n);
            if (S == 2) RCV_LAUNCH(k_resize_box<2>, grid, dim3(kBlock), 0, ctx->stream, s, d);
            else RCV_LAUNCH(k_resize_box<4>, grid, dim3(kBlock), 0, ctx->stream, s, d);
            return rcv_launch_check(ctx);
        }
    }
    float scx = (float)s.cols / (float)d.cols, scy = (float)s.rows / (float)d.rows;
    if (s.ch == 3 && s.cols >= 4 && d.rows <= 65535 * kRszRows) {   // (any width / alignment of source and destination)
        RCV_LAUNCH(k_resize_bgr, dim3((unsigned)((d.cols + kBlock - 1) / kBlock), (unsigned)((d.rows + kRszRows - 1) / kRszRows), d.n),
                           dim3(kBlock), 0, ctx->stream, s, d, scx, scy);
        return rcv_launch_check(ctx);
    }
    if (s.ch == 1 && s.cols >= 8 && d.rows <= 65535 * kRszRows) {   // (any width / alignment of source and destination)
        RCV_LAUNCH(k_resize_gray, dim3((unsigned)((d.cols + kBlock - 1) / kBlock), (unsigned)((d.rows + kRszRows - 1) / kRszRows), d.n),
                           dim3(kBlock), 0, ctx->stream, s, d, scx, scy);
        return rcv_launch_check(ctx);
    }
    if (s.ch == 1) RCV_LAUNCH(k_resize<1>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, scx, scy);
    else if (s.ch == 3) RCV_LAUNCH(k_resize<3>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, scx, scy);
    else RCV_LAUNCH(k_resize<4>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, scx, scy);
    return rcv_launch_check(ctx);
}

// Patch geometry of k_warp_affine_lds for the map M: staged rows, 4-pixel chunks per row and the LDS row pitch.  Extents
// from the matrix with slack for the floor, the right / lower tap and the 4-pixel alignment of the patch's first column.  The
// pitch is the one -- of the multiples of 16 bytes that keep both buffers within 64 KB -- for which the 32 lanes of a tap read
// (ds_read_b32 groups, bank = dword index mod 32) collide least: the lanes of a wave walk along a source row and step to the
// next row every 1 / |m3| pixels, so the best row pitch depends on the matrix.  A model of the first wave of two tiles decides.
static bool warp_lds_plan(const float* M, int* pitch_out, int* prow_out, int* cpr_out)
{
    for (int i = 0; i < 6; ++i)
        if (!(fabsf(M[i]) <= 3.0e38f)) return false;   // NaN, inf
    const int ew = kWlTW - 1, eh = kWlTH - 1;   // extent of the tile's sample grid
    const double wx = fabs((double)M[0]) * ew + fabs((double)M[1]) * eh;
    const double wy = fabs((double)M[3]) * ew + fabs((double)M[4]) * eh;
    if (!(wx < 2048.0 && wy < 2048.0)) return false;
    // Columns a tile needs: floor(xmax) - floor(xmin) + 2 (the right tap) <= floor(wx) + 3, + up to 3 for the patch's 4-pixel aligned
    // start; rows: floor(wy) + 3.  1/64 of slack for the kernel's f32 coordinates (a tile that needs more than the plan holds fails
    // the kernel's own test and takes the gather path: slower, never wrong).  Round 4: exact instead of generous bounds -- 7 degrees:
    // 19 x 42 -> 18 x 41 chunks, which is 3 instead of 4 chunk slots per thread (738 <= 768).
    const int cpr = ((int)floor(wx + 1.0 / 64) + 6 + 3) / 4;
    const int prow = (int)floor(wy + 1.0 / 64) + 3;
    if ((long long)cpr * prow > (long long)kWlMaxG * kBlock) return false;
    int best = 0;
    long long best_cost = -1;
    for (int pitch = 16 * cpr; pitch < 16 * cpr + 128; pitch += 16) {   // (the bank pattern repeats every 128 bytes of pitch)
        if (2LL * pitch * prow + pitch + 16 > 65536) break;   // (+ the f32 kernel's zeroed row)
        long long cost = 0;
        for (int tile = 0; tile < 2; ++tile) {
            const double ox = tile ? 7.0 * kWlTW : 0.0, oy = tile ? 3.0 * kWlTH : 0.0;
            for (int r = 0; r < kWarpRows; ++r)
                for (int half = 0; half < 2; ++half) {
                    int cnt[32] = {0}, seen[32][8];
                    for (int l = 32 * half; l < 32 * half + 32; ++l) {
                        const double px = l, py = r;
                        const double sx = M[0] * (ox + px) + M[1] * (oy + py) + (M[2] - floor(M[2])) + 4096.0;
                        const double sy = M[3] * (ox + px) + M[4] * (oy + py) + (M[5] - floor(M[5])) + 4096.0;
                        const long long dw = (long long)floor(sy) * (pitch / 4) + (long long)floor(sx);   // dword index up to a constant
                        const int bank = (int)(((dw % 32) + 32) % 32);
                        bool dup = false;
                        for (int k = 0; k < cnt[bank] && k < 8; ++k) dup = dup || seen[bank][k] == (int)(dw & 0x7fffffff);
                        if (!dup) {
                            if (cnt[bank] < 8) seen[bank][cnt[bank]] = (int)(dw & 0x7fffffff);
                            ++cnt[bank];
                        }
                    }
                    int worst = 1;
                    for (int b = 0; b < 32; ++b) worst = cnt[b] > worst ? cnt[b] : worst;
                    cost += worst;
                }
        }
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = pitch;
        }
    }
    if (best == 0) return false;
    *pitch_out = best;
    *prow_out = prow;
    *cpr_out = cpr;
    return true;
}

extern "C" int rcv_warp_affine_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst, const float* M)
{
    RCV_TRY(rcv_bind(ctx));
    if (!M) return RCV_ERR_ARG;
    View s, d;
    if (src && dst && src->frame0.depth == RCV_32F) {   // f32 images: the unrounded interpolated value, constant border 0.0f
        RCV_TRY(check_geom_f32(src, dst, &s, &d));
        if (d.rows == 0 || d.cols == 0 || d.n == 0) return RCV_OK;
        Affine Af;
        for (int i = 0; i < 6; ++i) Af.m[i] = M[i];
        if (s.ch == 1 && rcv_knobs().warp_lds != 0 && s.cols >= 16 && s.rows >= 4) {   // one channel: the LDS-staged kernel when the patch of a tile fits
            if (!ctx->wl_valid || memcmp(ctx->wl_M, M, sizeof(ctx->wl_M)) != 0) {
                ctx->wl_ok = warp_lds_plan(M, &ctx->wl_pitch, &ctx->wl_prow, &ctx->wl_cpr);
                memcpy(ctx->wl_M, M, sizeof(ctx->wl_M));
                ctx->wl_valid = true;
            }
            if (ctx->wl_ok) {
                const unsigned lgx = (unsigned)((d.cols + kWlTW - 1) / kWlTW), lgy = (unsigned)((d.rows + kWlTH - 1) / kWlTH);
                const unsigned long long t1 = (unsigned long long)lgx * lgy;
                int fpg = min(d.n, 4);   // (sharing the per-workgroup state beats the number of workgroups; groups of 1 / 2 / 4 / 8: 8 x 8K 0.486 / 0.416 / 0.399 / 0.419 ms, 8 x 1080p 0.0336 / 0.0237 / 0.0219 / 0.0235, 4 x 4K 0.0615 / 0.0419 / 0.0450)
                if ((rcv_knobs().warp_fpg & 255) > 0) fpg = min(rcv_knobs().warp_fpg & 255, d.n);
                const unsigned gz = (unsigned)((d.n + fpg - 1) / fpg);
                const unsigned long long tiles = t1 * gz;
                // (RCV_WARP_FPG >= 256: XCD-contiguous runs in strips of (value >> 8) - 1 tile columns, 0 = raster: tools/ablate_warp_order.py)
                // 8 x 8K rot 7: plain grid order 0.464 ms, XCD runs in raster order 0.473, strips of 2 / 4 / 8 / 16: 0.489 / 0.483 / 0.477 / 0.455
                // tile order (re-measured after the border tiles went onto the staged path, tools/ablate_warp_order.py --kind f32; plain grid /
                // XCD-contiguous raster / strips of 6): 8 x 8K 0.417 / 0.447 / 0.443 ms, 2 x 8K 0.113 / 0.118 / 0.114, 4 x 4K 0.058 / 0.042 / 0.043,
                // 1 x 1080p 0.0073 all: contiguous runs for launches of up to 16K tiles, the plain order above
                const bool xcd = rcv_knobs().warp_fpg >= 256 ? tiles < (1ull << 30) : tiles <= 16384;
                const int strip = rcv_knobs().warp_fpg >= 256 ? (rcv_knobs().warp_fpg >> 8) - 1 : 0;
                const int tpx = xcd ? (int)((tiles + 7) / 8) : 0;
                const dim3 grid = xcd ? dim3((unsigned)tpx * 8) : dim3(lgx, lgy, gz);
                RCV_LAUNCH(k_warp_f32_lds, grid, dim3(kBlock), 2u * (unsigned)ctx->wl_pitch * (unsigned)ctx->wl_prow + (unsigned)ctx->wl_pitch + 16u, ctx->stream, s, d, Af, fpg, ctx->wl_pitch,
                           ctx->wl_prow, ctx->wl_cpr, (int)lgx, (int)lgy, (int)tiles, tpx, strip);
                return rcv_launch_check(ctx);
            }
        }
        if (s.ch == 1) RCV_LAUNCH(k_warp_affine_f32<1>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, Af);
        else if (s.ch == 3) RCV_LAUNCH(k_warp_affine_f32<3>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, Af);
        else RCV_LAUNCH(k_warp_affine_f32<4>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, Af);
        return rcv_launch_check(ctx);
    }
    RCV_TRY(check_geom(src, dst, &s, &d));
    if (d.rows == 0 || d.cols == 0 || d.n == 0) return RCV_OK;
    Affine A;
    for (int i = 0; i < 6; ++i) A.m[i] = M[i];
    if ((s.ch == 3 && s.cols >= 3) || (s.ch == 1 && s.cols >= 8)) {   // (any destination width / alignment: ragged destinations store their quads piecewise)
        const unsigned gx = (unsigned)((d.cols + kWarpTW - 1) / kWarpTW), band = kWarpRows * (kBlock / kWarpTW);
        const unsigned gy = (unsigned)((d.rows + band - 1) / band);
        // frames per workgroup (the coordinate arithmetic is shared inside a group): as many of 8 / 4 / 2 as still leave >= 8192 workgroups
        const unsigned long long wgs = (unsigned long long)gx * gy * d.n;
        int fpg = wgs / 8 >= 8192 ? 8 : (wgs / 4 >= 8192 ? 4 : (wgs / 2 >= 8192 ? 2 : 1));
        if (s.ch == 3 && d.n % 16 == 0 && wgs / 16 >= 8192) fpg = 16;   // (32 x 8K: 1.479 -> 1.459 ms; 32 per group: 1.546)
        if ((rcv_knobs().warp_fpg & 255) > 0) fpg = min(rcv_knobs().warp_fpg & 255, d.n);
        const unsigned gz = (unsigned)((d.n + fpg - 1) / fpg);
        // the LDS-staged kernel when the source patch of a 64 x 32 tile is small enough (rotations, shears and scales near 1)
        int pitch = 0, prow = 0, cpr = 0;
        if (!ctx->wl_valid || memcmp(ctx->wl_M, M, sizeof(ctx->wl_M)) != 0) {   // (the plan depends on the matrix only: kept for the next call)
            ctx->wl_ok = warp_lds_plan(M, &ctx->wl_pitch, &ctx->wl_prow, &ctx->wl_cpr);
            memcpy(ctx->wl_M, M, sizeof(ctx->wl_M));
            ctx->wl_valid = true;
        }
        pitch = ctx->wl_pitch; prow = ctx->wl_prow; cpr = ctx->wl_cpr;
        if (rcv_knobs().warp_lds != 0 && s.cols >= 8 && s.rows >= 4 && ctx->wl_ok) {
            const unsigned lgx = (unsigned)((d.cols + kWlTW - 1) / kWlTW), lgy = (unsigned)((d.rows + kWlTH - 1) / kWlTH);
            // frames per workgroup of the staged kernels: sharing the coordinates, weights and the staging plan is worth more than the
            // number of workgroups -- groups of min(n, 8), 16 on frames of 4K and up (tools/ablate_warp_order.py --combos, BGR, after the
            // border tiles went onto the staged path; groups of 1 / 2 / 4 / 8 / 16): 4 x 4K 0.094 / 0.061 / 0.050 ms, 8 x 1080p 0.049 /
            // 0.033 / 0.028 / 0.029, 16 x 1080p 0.094 / 0.061 / 0.051 / 0.047 / 0.051, 16 x 4K 0.389 / 0.256 / 0.212 / 0.192 / 0.185,
            // 8 x 8K 0.764 / 0.503 / 0.415 / 0.374; the old rule (at least 8192 workgroups) ran 4 x 4K and 8 x 1080p in groups of 1
            fpg = min(d.n, 8);
            if (d.n % 16 == 0 && (unsigned long long)lgx * lgy >= 4000) fpg = 16;
            if ((rcv_knobs().warp_fpg & 255) > 0) fpg = min(rcv_knobs().warp_fpg & 255, d.n);
            const unsigned gzl = (unsigned)((d.n + fpg - 1) / fpg);
            const unsigned long long tiles = (unsigned long long)lgx * lgy * gzl;
            const unsigned lds = 2u * (unsigned)pitch * (unsigned)prow;
            const bool rags = (uintptr_t)s.p % 4 || s.step % 4 || (s.n > 1 && s.fstride % 4);   // byte-aligned source rows
            const bool xcd = tiles < (1ull << 30);
            const int tpx = xcd ? (int)((tiles + 7) / 8) : 0;
            // tile order inside an XCD's run: vertical strips of 6 tile columns walked row by row, so that the ~96 tiles an XCD has
            // in flight form a block whose patches overlap on all four sides inside ONE L2 (plain raster order: 0.8 of a tile row
            // in flight, the rows shared with the tiles above / below come from HBM again) -- when the map tilts the patch enough
            // for that overlap to matter: 8 x 8K, raster / strips of 6: 0 deg 0.381 / 0.432 ms, 1 deg 0.384 / 0.400, 3 deg 0.383 / 0.391,
            // 5 deg 0.389 / 0.383, 10 deg 0.394 / 0.372, 20 deg 0.431 / 0.390, 45 deg 0.443 / 0.416, 90 deg 0.378 / 0.409
            // (tools/ablate_warp_order.py --deg; strips of 6 .. 16 alike at 7 deg, 32 x 8K: 1.497 -> 1.465 ms)
            const int strip = rcv_knobs().warp_fpg >= 256 ? (rcv_knobs().warp_fpg >> 8) - 1
                              : (fabsf(M[3]) >= 0.07f && fabsf(M[4]) >= 0.5f * fabsf(M[3]) ? 6 : 0);   // (RCV_WARP_FPG = fpg + 256 * (strip + 1): the tool's override)
            const dim3 grid = xcd ? dim3((unsigned)tpx * 8) : dim3(lgx, lgy, gzl);
            if (s.ch == 1 && d.n >= 4 && rcv_knobs().warp_gray4 != 0 && (uintptr_t)s.p % 4 == 0 && s.step % 4 == 0 && s.fstride % 4 == 0) {
                // four frames per LDS pass, up to four passes per workgroup (the per-workgroup set-up is 0.12 of a 0.70-ms launch at two
                // passes).  Frames per workgroup x tile order, tools/ablate_warp_order.py --kind gray --combos, after the border tiles went
                // onto the staged path (before, more passes per workgroup only lengthened the tail of slow border tiles): 32 x 8K fq 8 / 16
                // raster 0.729 / 0.750, strips of 8 0.667 / 0.661 ms; 64 x 4K 0.376 / 0.368, 0.337 / 0.330; 16 x 4K 0.069 / 0.065, 0.070 /
                // 0.064 (fq 4: 0.087); 64 x 1080p 0.070 / 0.062, 0.072 / 0.064 (fq 4: 0.088)
                const unsigned long long t1 = (unsigned long long)lgx * lgy;
                int fq = d.n >= 16 ? 16 : (d.n >= 8 ? 8 : 4);
                if ((rcv_knobs().warp_fpg & 255) > 0) fq = max(4, min(rcv_knobs().warp_fpg & 255, d.n) & ~3);
                const unsigned gzq = (unsigned)((d.n + fq - 1) / fq);
                const unsigned long long tq = t1 * gzq;
                // tile order (re-measured after the border tiles went onto the staged path -- before, contiguous runs concentrated the slow
                // border tiles and lost 3 ... 17 % on small launches; tools/ablate_warp_order.py --kind gray; plain grid / XCD-contiguous raster /
                // strips of 8): 32 x 8K 0.671 / 0.675 / 0.640 ms, 8 x 8K 0.192 / 0.191 / 0.191, 16 x 4K 0.0985 / 0.0876 / 0.0893, 4 x 4K
                // 0.0296 / 0.0251 / 0.0257: contiguous runs always, in strips of 8 for a tilted map (the rule of the BGR kernel)
                const bool xq = tq < (1ull << 30);
                const int stripq = rcv_knobs().warp_fpg >= 256 ? (rcv_knobs().warp_fpg >> 8) - 1
                                   : (fabsf(M[3]) >= 0.07f && fabsf(M[4]) >= 0.5f * fabsf(M[3]) ? 8 : 0);
                const int tpq = xq ? (int)((tq + 7) / 8) : 0;
                const dim3 gridq = xq ? dim3((unsigned)tpq * 8) : dim3(lgx, lgy, gzq);
                if (prow * cpr <= 4 * kBlock) RCV_LAUNCH(k_warp_gray_lds4<4>, gridq, dim3(kBlock), lds, ctx->stream, s, d, A, fq, pitch, prow, cpr, (int)lgx, (int)lgy, (int)tq, tpq, stripq);
                else RCV_LAUNCH(k_warp_gray_lds4<kWlMaxG>, gridq, dim3(kBlock), lds, ctx->stream, s, d, A, fq, pitch, prow, cpr, (int)lgx, (int)lgy, (int)tq, tpq, stripq);
                return rcv_launch_check(ctx);
            }
            if (s.ch == 1) RCV_LAUNCH((k_warp_affine_lds<1, true>), grid, dim3(kBlock), lds, ctx->stream, s, d, A, fpg, pitch, prow, cpr, (int)lgx, (int)lgy, (int)tiles, tpx, strip);
            else if (rags) RCV_LAUNCH((k_warp_affine_lds<3, true>), grid, dim3(kBlock), lds, ctx->stream, s, d, A, fpg, pitch, prow, cpr, (int)lgx, (int)lgy, (int)tiles, tpx, strip);
            else RCV_LAUNCH((k_warp_affine_lds<3, false>), grid, dim3(kBlock), lds, ctx->stream, s, d, A, fpg, pitch, prow, cpr, (int)lgx, (int)lgy, (int)tiles, tpx, strip);
            return rcv_launch_check(ctx);
        }
        if (s.ch == 3) {
            RCV_LAUNCH(k_warp_affine_bgr, dim3(gx, gy, gz), dim3(kBlock), 0, ctx->stream, s, d, A, fpg);
            return rcv_launch_check(ctx);
        }
    }
    if (s.ch == 1 && s.cols >= 8) {   // (any width / alignment of source and destination)
        RCV_LAUNCH(k_warp_affine_gray, dim3((unsigned)((d.cols + kBlock - 1) / kBlock), (d.rows + kWarpRows - 1) / kWarpRows, d.n), dim3(kBlock), 0,
                           ctx->stream, s, d, A);
        return rcv_launch_check(ctx);
    }
    if (s.ch == 1) RCV_LAUNCH(k_warp_affine<1>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, A);
    else if (s.ch == 3) RCV_LAUNCH(k_warp_affine<3>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, A);
    else RCV_LAUNCH(k_warp_affine<4>, px_grid(d), dim3(kBlock), 0, ctx->stream, s, d, A);
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
