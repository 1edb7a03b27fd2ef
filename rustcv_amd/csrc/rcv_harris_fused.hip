// rcv_harris_fused.hip -- BGR u8 -> gray -> Sobel -> Harris response (blockSize 2, aperture 3) -> 3x3 NMS mask in ONE
// launch, as a register sliding window down the image.  Algorithmic traffic 3 B read + 1 B written per pixel
// (+4 B when the f32 response is requested); the unfused path moves ~30 B/px through HBM.
//
// One WAVE owns a strip of 496 px (62 lanes x 8 px; lanes 0 / 63 carry halo only) and walks down a row segment.
// Per source row a lane loads its 8 BGR pixels (3 x 8 B), converts to gray, and every stage pulls the one neighbour
// value it needs from the adjacent lane with a DPP wave shift:
//     gray      needs g[x-1], g[x+1]            (Sobel: packed f32 on aligned shapes, packed i16 as in rcv_sobel_rows.hip elsewhere)
//     box 2x2   needs P[x-1]  (P = Ix^2, IxIy, Iy^2 ; window offsets -1..0, anchor = blockSize/2 = 1)
//     NMS 3x3   needs r[x-1], r[x+1]
// Vertical neighbours are earlier rows kept in registers (2 rows of Sobel parts, 1 row of horizontal box sums,
// 2 rows of raw responses for the NMS).  The stream is fed VIRTUAL row indices v = ys-3 .. ye+1 resolved with BORDER_REFLECT_101,
// which reproduces the box filter's reflection of the product image at the top border (P(-1) := P(1)) provided Iy
// is negated on the mirrored row (a mirrored Sobel flips the sign of dy); the left border P(-1) := P(1) is a lane
// fix-up; response values outside the image are -inf for the NMS.  f32 ops follow the oracle's order exactly
// (six separate IEEE ops, -ffp-contract=off).
// Round 6, from the ISA: 192 -> ~165 vector instructions per row.  The packed-f32 stages work on pairs {pixel j, pixel j + 4};
// pairs 0 and 3 are kept with their halves swapped (op_sel where they are plain operands) so that the strip-end neighbour pairs are those
// registers after one in-place DPP move; aligned shapes address their rows through buffer resources (row offset in the instruction's scalar
// offset: no vector address arithmetic); the NMS takes column maxima first and forms the mask bytes from the sign of centre - maximum.
// RAG instantiation: any width >= 8 and any alignment (Mat::new gives step = cols * channels: odd widths mean byte-aligned
// rows): unaligned vector loads / stores, the lane with the row's last, partial run rebuilds its 8 logical gray pixels --
// the valid ones and their mirror images -- with one byte permute per dword, sets the responses right of the image to -inf
// for the NMS and stores only its valid samples (same scheme as rcv_sobel_rows.hip).
#include "rcv_internal.h"
#include "rcv_kernels.h"
#include "rcv_device_utils.h"
#include <math.h>
#include <type_traits>
#include <stdlib.h>

namespace {

typedef short s2v __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
#ifndef RCV_HF_AHEAD
#define RCV_HF_AHEAD 2
#endif
#ifndef RCV_HF_FSOB
#define RCV_HF_FSOB 1
#endif
#ifndef RCV_HF_FSOB_GRAY
#define RCV_HF_FSOB_GRAY 1
#endif
#ifndef RCV_HF_FSOB_RESP
#define RCV_HF_FSOB_RESP 1
#endif
#ifndef RCV_HF_RESP_LDS
#define RCV_HF_RESP_LDS 0
#endif
#ifndef RCV_HF_MASK_AUX
#define RCV_HF_MASK_AUX 0
#endif
#ifndef RCV_HF_RESP_T
#define RCV_HF_RESP_T 1
#endif
#ifndef RCV_HF_RESP_AUX
#define RCV_HF_RESP_AUX 2   // (non-temporal)
#endif
#ifndef RCV_HF_LIMROW
#define RCV_HF_LIMROW 1
#endif
constexpr int kAhead = RCV_HF_AHEAD;   // source rows in flight per lane (x 6 VGPRs)
constexpr int kStripPx = 62 * 8;
#ifndef RCV_HF_WPB
#define RCV_HF_WPB 4
#endif
constexpr int kWPB = RCV_HF_WPB;      // waves per workgroup (the waves of a workgroup share nothing)

struct HArgs {
    const uint8_t* src;
    uint8_t *mask, *resp;
    size_t sstep, mstep, rstep, sfs, mfs, rfs;
    int rows, cols, nstrips, seg_rows, nsegs, total_waves;
    int blocks_per_xcd;   // > 0: XCD-contiguous block order (as rcv_sobel_rows.hip)
    float s2, k, thr_up;   // thr_up: smallest float > thr (+inf for a NaN threshold: nothing is kept, as `rc > NaN` is never true)
#ifdef RCV_HF_BENCH
    unsigned long long* trace;   // (measurement library) per wave {start, end} of the chip-wide 100 MHz counter, {XCD, CU} in a third word
#endif
};

__device__ __forceinline__ uint32_t pk(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2v, a) - __builtin_bit_cast(s2v, b)); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2v, a) + __builtin_bit_cast(s2v, b)); }
// a + 2 * b on both halves in ONE instruction (the compiler turns the C expression into a packed shift and a packed add)
__device__ __forceinline__ uint32_t pk_add2x(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_pk_mad_i16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(b), "v"(a));
    return d;
}
// bound_ctrl: a lane without a source lane (lane 0 / lane 63) reads 0 -- the same value `old = 0` would leave, without the
// v_mov that initialises `old` before every DPP
__device__ __forceinline__ uint32_t shr1(uint32_t v) { return __builtin_amdgcn_update_dpp(0u, v, 0x138, 0xf, 0xf, true); }  // from lane-1
__device__ __forceinline__ uint32_t shl1(uint32_t v) { return __builtin_amdgcn_update_dpp(0u, v, 0x130, 0xf, 0xf, true); }  // from lane+1
__device__ __forceinline__ float shr1f(float v) { return __builtin_bit_cast(float, shr1(__builtin_bit_cast(uint32_t, v))); }
__device__ __forceinline__ float shl1f(float v) { return __builtin_bit_cast(float, shl1(__builtin_bit_cast(uint32_t, v))); }

// The NMS maxima as the instructions themselves: through fmaxf the compiler first canonicalises every operand it cannot prove quiet (a
// v_max_f32 x, x per pixel: values extracted from packed-f32 results, DPP moves); the hardware maxima give the same result for the values
// that occur here (finite responses and -inf) without it.  Round 4, from the ISA: 1 of 29 instructions per pixel.
__device__ __forceinline__ float vmax2(float a, float b)
{
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float vmax3(float a, float b, float c)
{
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// Packed-f32 operations that read one operand with its halves SWAPPED (op_sel), for the pairs kept as {pixel j + 4, pixel j}: round 6.  A lane's
// pairs are {pixel j, pixel j + 4}; the neighbour pair of j = 0 is {pixel -1, pixel 3} = {lane-1's pixel 7, own pixel 3}.  With pair 3 kept
// SWAPPED, {pixel 7, pixel 3}, that neighbour pair is pair 3 itself after ONE in-place DPP move of its low half -- no copy of pixel 3 into
// a second register pair (the compiler, given plain vector code, un-swaps the pair and copies).  The swap costs nothing where pair 3 is an
// ordinary operand: op_sel picks the halves.  Same for pair 0 = {pixel 4, pixel 0} and the neighbour pair {pixel 4, pixel 8} of j = 3.
__device__ __forceinline__ f2 pk_add_bs(f2 a, f2 b)   // {a.x + b.y, a.y + b.x}
{
    f2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f2 pk_sub_bs(f2 a, f2 b)   // {a.x - b.y, a.y - b.x}
{
    f2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f2 pk_sub_as(f2 a, f2 b)   // {a.y - b.x, a.x - b.y}
{
    f2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f2 pk_fma2_as(f2 a, f2 c)   // {2 a.y + c.x, 2 a.x + c.y}
{
    f2 d;
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(c));
    return d;
}
__device__ __forceinline__ f2 pk_mul_ss(f2 a, f2 b)   // {a.y * b.y, a.x * b.x}
{
    f2 d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// byte 2 of a dword as f32 in ONE instruction (as C the compiler pulls the conversion through the Sobel's additions and does those in integer)
__device__ __forceinline__ float cvb2(uint32_t w)
{
    float f;
    asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(w));
    return f;
}

struct Row6 { uint32_t d[6]; };
struct U2 { uint32_t a, b; };
// global-address-space views: a pointer laundered through an SGPR constraint would otherwise decay to a flat pointer
#define RCV_GLOBAL __attribute__((address_space(1)))
typedef RCV_GLOBAL uint8_t* gptr;
typedef const RCV_GLOBAL uint8_t* cgptr;
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
// unaligned views (RAG)
typedef uint32_t U2m __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t U1m __attribute__((aligned(1)));
typedef uint16_t H1m __attribute__((aligned(1)));
typedef float F4m __attribute__((ext_vector_type(4), aligned(4)));
typedef float F2m __attribute__((ext_vector_type(2), aligned(4)));

// YUYV = true: the source is packed YUYV (2 B/px, SURVEY.md 8(d) config 5 "[or YUYV]"); each macropixel goes through the
// reference's BT.601 conversion (rustcv/src/videoio/mod.rs:356-363, saturated to u8) and then the same gray formula, so the
// result equals harris_pipeline(yuyv_to_bgr(.)) bit for bit with 3 instead of 4 algorithmic bytes per pixel.
// SRCK: 0 = BGR, 1 = packed YUYV, 2 = one-channel gray (cornerHarris' own input: the window starts at the Sobel stage).
// WANT_MASK = false: response only (rcv_corner_harris): no NMS stage, no mask store.
#ifdef RCV_HF_OCC   // (measurement: cap the registers for RCV_HF_OCC waves per SIMD)
#define RCV_HF_OCC_ATTR __attribute__((amdgpu_waves_per_eu(RCV_HF_OCC, RCV_HF_OCC)))
#else
#define RCV_HF_OCC_ATTR
#endif
template <bool WANT_RESP, int SRCK, bool WANT_MASK = true, bool RAG = false>
__global__ __launch_bounds__(64 * kWPB) RCV_HF_OCC_ATTR void k_harris_fused(HArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float hf_lds[];   // (aligned launches with the response: 2 KB per wave, see the response store)
    const int lane = threadIdx.x & 63;
    // the wave index as a SCALAR: strip / segment / frame, the reflected row indices and every row base address below are
    // then SALU work (as VALU work the 64-bit row multiplies alone were ~100 quarter-rate slots per four rows)
    const int blk = a.blocks_per_xcd > 0 ? (int)(blockIdx.x & 7) * a.blocks_per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    int wid = __builtin_amdgcn_readfirstlane(blk * kWPB + (int)(threadIdx.x >> 6));
    if (wid >= a.total_waves) return;
#ifdef RCV_HF_BENCH
    const int wid0 = wid;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
    const int strip = wid % a.nstrips;
    wid /= a.nstrips;
    const int seg = wid % a.nsegs;
    const int frame = wid / a.nsegs;
    const int ys = seg * a.seg_rows, ye = min(a.rows, ys + a.seg_rows);
    const int x = strip * kStripPx + 8 * (lane - 1);
    const int xc = min(max(x, 0), a.cols - 8);
    const bool edgeL = x < 0, edgeR = x == a.cols;
    const bool live = lane >= 1 && lane <= 62 && x < a.cols;
    // (wave-uniform) the strip holds the lane left of x = 0 / the lane at x = cols.  Kept as a scalar INTEGER the compiler cannot see through: as a
    // bool it lived as a lane mask and one of its three uses per row rebuilt it with a v_cndmask + v_cmp pair (round 6, from the ISA)
    int has_edge_i = __builtin_amdgcn_readfirstlane((strip == 0 || (strip + 1) * kStripPx + 8 > a.cols) ? 1 : 0);
    asm volatile("" : "+s"(has_edge_i));
#define has_edge (has_edge_i != 0)
    // RAG: the lane's 8 logical pixels x .. x+7 (right of the image: their mirror images) as byte selectors into the gray run it
    // gets from the clamped position xc; identity wherever the run lies inside the image (and for the left halo lane, which
    // keeps its own fix-up below)
    uint32_t sel_lo = 0x03020100u, sel_hi = 0x07060504u;
    int nvalid = 8;
    if (RAG && x >= 0) {
        nvalid = min(max(a.cols - x, 0), 8);
        sel_lo = sel_hi = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = x + j, pr = p < a.cols ? p : 2 * a.cols - 2 - p;
            const uint32_t idx = (uint32_t)min(max(pr - xc, 0), 7);
            if (j < 4) sel_lo |= idx << (8 * j);
            else sel_hi |= idx << (8 * (j - 4));
        }
    }
    // uniform frame bases + 32-bit per-lane offsets: the loads and stores take the scalar-base + vector-offset form
    const uint8_t* const sf = a.src + (size_t)frame * a.sfs;
    uint8_t* const mf = a.mask + (size_t)frame * a.mfs;
    uint8_t* const rf = WANT_RESP ? a.resp + (size_t)frame * a.rfs : nullptr;
    constexpr bool YUYV = SRCK == 1, GRAY = SRCK == 2;
    // Round 6, aligned shapes: BUFFER loads / stores -- the frame base in the resource, the row's byte offset in the instruction's SCALAR offset,
    // the lane's column offset in its 32-bit vector offset: no vector address arithmetic at all (the global form added the 64-bit row base
    // to a 64-bit lane offset with one v_lshl_add_u64 per row and per store), and one s_mul_i32 per row where the 64-bit row base took six
    // scalar instructions.  The host sends frames of 4 GB and more (rows * step) to the RAG instantiation, which keeps 64-bit pointers.
    constexpr int kRsrc = 0x00020000;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)sf, 0, 0xffffffff, kRsrc);
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void*)mf, 0, 0xffffffff, kRsrc);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)rf, 0, 0xffffffff, kRsrc);
    const uint32_t sstep32 = (uint32_t)a.sstep, mstep32 = (uint32_t)a.mstep, rstep32 = (uint32_t)a.rstep;
    constexpr bool TQ = !GRAY && !RAG;   // gray values stay one per dword (byte 2), see feed
    // the Sobel stage in packed f32 (mask-only launches: with the response the 16 more registers cost the third wave per SIMD); same box,
    // same run, 64 x 4K: 0.4984-0.5012 ms against 0.5148-0.5189 with the packed-i16 Sobel
    // (round 6: a one-channel source too -- its eight pixels are the bytes of the two source dwords, v_cvt_f32_ubyte0..3 reads them in place)
    constexpr bool FSOB = (TQ || (GRAY && !RAG && RCV_HF_FSOB_GRAY)) && (!WANT_RESP || RCV_HF_FSOB_RESP) && RCV_HF_FSOB;
    const uint32_t sx = (uint32_t)((GRAY ? 1 : (YUYV ? 2 : 3)) * xc), mx = (uint32_t)max(x, 0);
    const float NEG_INF = -INFINITY;
    const float thr_v = a.thr_up;

    auto load_row = [&](int v) -> Row6 {  // virtual row -> reflected source row (clamped past what the segment needs)
        // (round 6, scalar instruction count: a wave issues one instruction of ANY kind every ~10 cycles, tools/ubench_dpp.hip)  BORDER_REFLECT_101 of
        // v <= rows + 1 in three instructions: |v|, then the smaller of it and its mirror image at the bottom edge (for 0 <= v < rows the image is >= v)
        v = min(v, ye + 1);
        const int av = abs(v);
        const int r = min(av, 2 * a.rows - 2 - av);
        if constexpr (RAG) {
            cgptr p = (cgptr)(sf + (size_t)r * a.sstep);
            asm("" : "+s"(p));   // the row base stays in SGPRs
            const U2m q0 = *(const RCV_GLOBAL U2m*)(p + sx);
            if constexpr (GRAY) return Row6{{q0.x, q0.y, 0u, 0u, 0u, 0u}};
            const U2m q1 = *(const RCV_GLOBAL U2m*)(p + sx + 8);
            if constexpr (YUYV) return Row6{{q0.x, q0.y, q1.x, q1.y, 0u, 0u}};
            const U2m q2 = *(const RCV_GLOBAL U2m*)(p + sx + 16);
            return Row6{{q0.x, q0.y, q1.x, q1.y, q2.x, q2.y}};
        }
        const uint32_t so = (uint32_t)r * sstep32;
        if constexpr (GRAY) {
            const u2v q0 = __builtin_amdgcn_raw_buffer_load_b64(srs, sx, so, 0);
            return Row6{{q0.x, q0.y, 0u, 0u, 0u, 0u}};
        }
        const u4v q0 = __builtin_amdgcn_raw_buffer_load_b128(srs, sx, so, 0);
        if constexpr (YUYV) return Row6{{q0.x, q0.y, q0.z, q0.w, 0u, 0u}};
        else {
            const u2v q2 = __builtin_amdgcn_raw_buffer_load_b64(srs, sx + 16, so, 0);
            return Row6{{q0.x, q0.y, q0.z, q0.w, q2.x, q2.y}};
        }
    };

    // ---- pipeline state --------------------------------------------------------------------------------------
    uint32_t h1a[4], h1b[4], h2a[4], h2b[4];      // Sobel horizontal parts of gray rows v-2, v-1 (packed i16 pairs)
    f2 f1a[4], f1b[4], f2a[4], f2b[4];            // ... the same as packed f32 pairs {pixel j, pixel j + 4} (FSOB)
    f2 hsxx[4], hsxy[4], hsyy[4];                 // horizontal box sums of product row u-1 -- kept in f32: every product
                                                  // (<= 1020^2) and every 2x2 sum (< 2^24) is an exactly representable
                                                  // integer, so f32 adds/muls are exact and can use the packed f32 ALU
    float ra[8], rb[8];                           // NMS: responses of rows u-2, u-1 (outside the image: -inf)
    int cand_b = 0;                               // ... and whether row u-1 holds a pixel >= thr_up (wave-uniform; an integer: as a bool it lived as a lane mask)
    // aligned shapes: the mask row formed at the end of one feed is stored after the Sobel stage of the NEXT one.  The wait for the next row
    // group's loads at the head of the loop counts every vector-memory operation issued before it, stores included (one in-order vmcnt): a
    // store issued a third of a row earlier has long left, the one the feed just issued had not
    uint32_t pm0 = 0, pm1 = 0;
    int pdirty = 0;                               // (wave-uniform) pm0 / pm1 hold a row's mask
    int pw = -1;
    auto flush_mask = [&]() {
        if ((unsigned)(pw - ys) < (unsigned)(ye - ys) && live) __builtin_amdgcn_raw_buffer_store_b64(u2v{pm0, pm1}, mrs, mx, (uint32_t)pw * mstep32, RCV_HF_MASK_AUX);
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h1a[j] = h1b[j] = h2a[j] = h2b[j] = 0;
        f1a[j] = f1b[j] = f2a[j] = f2b[j] = f2{0.0f, 0.0f};
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hsxx[j & 3] = hsxy[j & 3] = hsyy[j & 3] = f2{0.0f, 0.0f};
        ra[j] = rb[j] = NEG_INF;
    }

    auto feed = [&](const Row6& q, int v) {
        // ---- gray (8 px) -----------------------------------------------------------------------------------------
        // weights 1868, 9617, 4899 = 256*{7,37,19} + {76,145,35}: two v_dot4_u32_u8 per pixel on the pixel's (B,G,R,x) dword
        uint32_t g[8];
        // TQ (aligned BGR / YUYV sources): the weights times 4 -- 7472, 38468, 19596 = 256*{29,150,76} + {48,68,140}, rounding term
        // 4*8192 -- put the gray value (sum >> 14) into bits 16..23, a whole byte that the Sobel's byte permutes read in place:
        // no shift, and no packing of the eight values
        // (up = true: the pixel sits in bytes 1..3 of its dword -- pixels 3 and 7 of a 24-byte run -- and the weights move up instead
        //  of the data moving down: no v_alignbyte for those two)
        auto gray_of = [](uint32_t px, bool up = false) -> uint32_t {
            if constexpr (TQ) {
                const uint32_t hi8 = __builtin_amdgcn_udot4(px, up ? 0x4c961d00u : 0x004c961du, 0u, false);
                const uint32_t lo8 = __builtin_amdgcn_udot4(px, up ? 0x8c443000u : 0x008c4430u, 32768u, false);
                return (hi8 << 8) + lo8;
            } else {
                const uint32_t hi8 = __builtin_amdgcn_udot4(px, 0x00132507u, 0u, false);   // 7*B + 37*G + 19*R
                const uint32_t lo8 = __builtin_amdgcn_udot4(px, 0x0023914cu, 8192u, false);  // 76*B + 145*G + 35*R + 8192
                return ((hi8 << 8) + lo8) >> 14;
            }
        };
        if constexpr (GRAY) {
            // (the 8 gray pixels are the two source dwords themselves)
        } else if constexpr (YUYV) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {   // macropixel [Y0 U Y1 V] -> two (B,G,R,0) dwords, as the reference converts them
                const uint32_t w = q.d[m];
                const int y0 = (int)(w & 0xff), u = (int)((w >> 8) & 0xff) - 128, y1 = (int)((w >> 16) & 0xff), vv = (int)(w >> 24) - 128;
                const int c0 = 298 * (y0 - 16) + 128, c1 = 298 * (y1 - 16) + 128;
                const int db = 516 * u, dg = -100 * u - 208 * vv, dr = 409 * vv;
                g[2 * m] = gray_of(rcv_ashr_sat_pk4(c0 + db, c0 + dg, c0 + dr, 0, 8));
                g[2 * m + 1] = gray_of(rcv_ashr_sat_pk4(c1 + db, c1 + dg, c1 + dr, 0, 8));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k0 = 3 * j, w0 = k0 >> 2, sh = k0 & 3;   // pixel j = bytes k0..k0+2 of the 24-byte run
                if (TQ && sh == 1) g[j] = gray_of(q.d[w0], true);
                else g[j] = gray_of(sh == 0 ? q.d[w0] : __builtin_amdgcn_alignbyte(q.d[w0 + 1 < 6 ? w0 + 1 : 5], q.d[w0], sh));
            }
        }
        // ---- Sobel: I(u) for u = v-1 -------------------------------------------------------------------------------
        uint32_t L[5], Cc[4];   // zero-extended gray pairs (g[2j-1], g[2j]) and (g[2j], g[2j+1])
        f2 fix[4], fiy[4];      // (FSOB) Ix, Iy of row u as f32 pairs
        if constexpr (TQ || FSOB) {
            // the gray values sit in byte 2 of their dwords: the pairs are picked straight from there, the eight bytes are
            // never packed into two dwords
            // (the two mirror lanes exist in the first / last strip only: a wave-uniform branch around the selects -- 13 of 15 strips
            //  of a 4K row skip them; the volatile asm keeps the compiler from turning it back into selects)
            if constexpr (TQ) {
                if (has_edge) {
                    asm volatile("; strip with a mirror lane");
                    if (edgeL) g[7] = g[1];   // x = -1 mirrors x = 1
                    if (edgeR) g[0] = g[6];   // x = cols mirrors cols-2
                }
            }
            if constexpr (FSOB) {
                // Round 4: the Sobel in PACKED F32 on the pairs {pixel j, pixel j + 4} the later stages use anyway.  v_cvt_f32_ubyte2 takes
                // the gray value straight out of byte 2 of its dword (one instruction per value), every intermediate is a small integer, so
                // the f32 arithmetic is exact and Ix, Iy come out as the SAME f32 values the integer path converts to -- without its nine
                // byte permutes and sixteen i16 -> f32 conversions.
                // Round 6: pairs 0 and 3 are kept SWAPPED ({4, 0} and {7, 3}); the neighbour pairs {-1, 3} and {4, 8} are then those two
                // registers after one in-place DPP move each (of the f32 values: the neighbours' conversions are not repeated here) -- eight
                // conversions and two DPP moves per row where there were ten, two DPP moves and two copies.  The sums of j = 0 / j = 3 are
                // split so that everything that reads the lane's own pixel 7 / pixel 0 comes before the move that replaces it (exact integers:
                // the order of the additions does not matter).
                auto gf = [&](int k) -> float {   // pixel k of the lane as f32
                    if constexpr (GRAY) return (float)((q.d[k >> 2] >> (8 * (k & 3))) & 0xffu);   // (v_cvt_f32_ubyteN)
                    else return cvb2(g[k]);
                };
                const f2 P1 = f2{gf(1), gf(5)}, P2 = f2{gf(2), gf(6)};
                f2 S0 = f2{gf(4), gf(0)};   // pair 0 swapped
                f2 S3 = f2{gf(7), gf(3)};   // pair 3 swapped
                if constexpr (GRAY) {
                    if (has_edge) {
                        asm volatile("; strip with a mirror lane");
                        if (edgeL) S3.x = P1.x;   // x = -1 mirrors x = 1
                        if (edgeR) S0.y = P2.y;   // x = cols mirrors cols-2
                    }
                }
                f2 h1[4], h2[4];
                h1[1] = pk_sub_bs(P2, S0);                                                        // P2 - P0
                h2[1] = __builtin_elementwise_fma(P1, f2{2.0f, 2.0f}, pk_add_bs(P2, S0));         // 2 P1 + P0 + P2
                h1[2] = pk_sub_as(S3, P1);                                                        // P3 - P1
                h2[2] = __builtin_elementwise_fma(P2, f2{2.0f, 2.0f}, pk_add_bs(P1, S3));         // 2 P2 + P1 + P3
                const f2 a0 = pk_fma2_as(S0, P1), b3 = pk_fma2_as(S3, P2);                        // 2 P0 + P1, 2 P3 + P2
                const f2 Pm = f2{shr1f(S3.x), S3.y};   // pixels {-1, 3}
                const f2 Pp = f2{S0.x, shl1f(S0.y)};   // pixels {4, 8}
                h1[0] = P1 - Pm;
                h2[0] = a0 + Pm;
                h1[3] = Pp - P2;
                h2[3] = b3 + Pp;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    fix[j] = __builtin_elementwise_fma(f1b[j], f2{2.0f, 2.0f}, f1a[j] + h1[j]);
                    fiy[j] = h2[j] - f2a[j];
                    f1a[j] = f1b[j];
                    f1b[j] = h1[j];
                    f2a[j] = f2b[j];
                    f2b[j] = h2[j];
                }
            } else {
            const uint32_t lf = shr1(g[7]), rt = shl1(g[0]);
            constexpr uint32_t kPair = 0x0c060c02u;   // (byte 2 of the low source, byte 2 of the high source) as two u16
            L[0] = pk(g[0], lf, kPair);
            L[1] = pk(g[2], g[1], kPair);
            L[2] = pk(g[4], g[3], kPair);
            L[3] = pk(g[6], g[5], kPair);
            L[4] = pk(rt, g[7], kPair);
#pragma unroll
            for (int j = 0; j < 4; ++j) Cc[j] = pk(g[2 * j + 1], g[2 * j], kPair);
            }
        } else {
            uint32_t lo = GRAY ? q.d[0] : g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24), hi = GRAY ? q.d[1] : g[4] | (g[5] << 8) | (g[6] << 16) | (g[7] << 24);
            if (edgeL) hi = pk(lo, hi, 0x05020100u);   // x = -1 mirrors x = 1
            if (RAG) {
                const uint32_t l2 = pk(hi, lo, sel_lo), h2 = pk(hi, lo, sel_hi);
                lo = l2;
                hi = h2;
            } else if (edgeR) {
                lo = pk(hi, lo, 0x03020106u);   // x = cols mirrors cols-2
            }
            const uint32_t lf = shr1(hi), rt = shl1(lo);
            L[0] = pk(lf, lo, 0x0c000c07u);
            L[1] = pk(lo, lo, 0x0c020c01u);
            L[2] = pk(hi, lo, 0x0c040c03u);
            L[3] = pk(hi, hi, 0x0c020c01u);
            L[4] = pk(rt, hi, 0x0c040c03u);
            Cc[0] = pk(lo, lo, 0x0c010c00u);
            Cc[1] = pk(lo, lo, 0x0c030c02u);
            Cc[2] = pk(hi, hi, 0x0c010c00u);
            Cc[3] = pk(hi, hi, 0x0c030c02u);
        }
        if constexpr (WANT_MASK && !RAG) flush_mask();
        const int u = v - 1;
        const bool mirrored = (unsigned)u >= (unsigned)a.rows;  // I(u) was formed from a vertically mirrored window: dy changes sign
        constexpr bool kLimRow = !WANT_RESP && !RAG && RCV_HF_LIMROW;
        // f32 stages on PACKED pairs {pixel j, pixel j+4} (v_pk_mul/add_f32): with this pairing the horizontal neighbour
        // P(x-1) of a pair is simply the previous pair register (j >= 1) -- pairing adjacent pixels {2j, 2j+1} instead leaves
        // every neighbour pair {2j-1, 2j} straddling two registers, and the compiler rebuilds it with ~3 v_mov per pixel
        f2 ix2[4], iy2[4];
        float ixs[8], iys[8];
        if constexpr (FSOB) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ix2[j] = fix[j];
                iy2[j] = fiy[j];
            }
        } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t h1 = pk_sub(L[j + 1], L[j]);
            const uint32_t h2 = pk_add2x(pk_add(L[j], L[j + 1]), Cc[j]);
            const uint32_t ox = pk_add2x(pk_add(h1a[j], h1), h1b[j]);
            const uint32_t oy = pk_sub(h2, h2a[j]);
            h1a[j] = h1b[j];
            h1b[j] = h1;
            h2a[j] = h2b[j];
            h2b[j] = h2;
            ixs[2 * j] = (float)(int)(short)(ox & 0xffff);
            ixs[2 * j + 1] = (float)((int)ox >> 16);
            iys[2 * j] = (float)(int)(short)(oy & 0xffff);
            iys[2 * j + 1] = (float)((int)oy >> 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ix2[j] = f2{ixs[j], ixs[j + 4]};
            iy2[j] = f2{iys[j], iys[j + 4]};
        }
        }
        // ---- products and 2x2 box sums: S(u) = Hs(u-1) + Hs(u), Hs(x) = P(x-1) + P(x) -------------------------------
        // Round 6: the products of pair 3 are formed SWAPPED, {pixel 7, pixel 3}: the neighbour pair of j = 0, {P(-1), P(3)}, is then that register
        // after one in-place DPP move of its low half (lane-1's pixel 7) -- the copy of P(3) into a second pair is gone (three per row), and the
        // select that makes P(-1) := P(1) at the image's left edge runs on the strips that hold that edge only (three more on the others).
        f2 pxx[4], pxy[4], pyy[4];   // ([3]: swapped)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            pxx[j] = ix2[j] * ix2[j];
            pyy[j] = iy2[j] * iy2[j];          // (-iy)^2 == iy^2 exactly
        }
        pxx[3] = pk_mul_ss(ix2[3], ix2[3]);
        pyy[3] = pk_mul_ss(iy2[3], iy2[3]);
        // `mirrored` is a scalar condition (rows outside the image only): a uniform branch between the product and its exact
        // negation (a free source modifier) instead of one v_cndmask per pixel
#pragma unroll
        for (int j = 0; j < 3; ++j) pxy[j] = ix2[j] * iy2[j];
        pxy[3] = pk_mul_ss(ix2[3], iy2[3]);
        if (mirrored) {
            // (the volatile asm keeps this a BRANCH: if-converted, the two arms met in register copies -- four v_mov_b64 on every row of
            //  the image for the sake of the two rows outside it; round 4, from the ISA)
            asm volatile("; row outside the image: dy changes sign");
#pragma unroll
            for (int j = 0; j < 4; ++j) pxy[j] = -pxy[j];   // exact
        }
        // Hs of pixels {3, 7} first: the last readers of the lane's own P(7)
        const f2 n3xx = pk_add_bs(pxx[2], pxx[3]), n3xy = pk_add_bs(pxy[2], pxy[3]), n3yy = pk_add_bs(pyy[2], pyy[3]);
        // P(x-1) of the lane's first pixel comes from lane-1's last pixel; at the image's left edge P(-1) := P(1): the lane left of x = 0
        // (it holds the pixels 0 .. 7 of the row) hands P(1) on
        if (has_edge) {
            asm volatile("; strip with the image's left edge: P(-1) := P(1)");
            if (edgeL) {
                pxx[3].x = pxx[1].x;
                pxy[3].x = pxy[1].x;
                pyy[3].x = pyy[1].x;
            }
        }
        const f2 q0xx = f2{shr1f(pxx[3].x), pxx[3].y}, q0xy = f2{shr1f(pxy[3].x), pxy[3].y}, q0yy = f2{shr1f(pyy[3].x), pyy[3].y};   // pixels {-1, 3}
        float r[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // neighbours of pixels {j, j+4}: pixels {j-1, j+3}
            const f2 nxx = j == 3 ? n3xx : (j ? pxx[j - 1] : q0xx) + pxx[j];
            const f2 nxy = j == 3 ? n3xy : (j ? pxy[j - 1] : q0xy) + pxy[j];
            const f2 nyy = j == 3 ? n3yy : (j ? pyy[j - 1] : q0yy) + pyy[j];
            const f2 sxx = hsxx[j] + nxx, sxy = hsxy[j] + nxy, syy = hsyy[j] + nyy;   // == (float)(exact integer sum)
            hsxx[j] = nxx;
            hsxy[j] = nxy;
            hsyy[j] = nyy;
            const f2 fa = sxx * a.s2, fb = sxy * a.s2, fc = syy * a.s2;
            const f2 t1 = fa * fc, t2 = fb * fb, t3 = fa + fc;
            const f2 t4 = a.k * t3;
            const f2 t5 = t4 * t3;
            f2 rr2 = (t1 - t2) - t5;
            // (kLimRow, rounds 5 / 6) a row outside the image: its responses stay what the arithmetic gives and the NMS below leaves that row out.
            // Overwriting r[] with -inf in a branch made every r[j] a merge of two definitions: six to seven register copies on EVERY row;
            // round 5 added a row-uniform 0 / -inf to the packed results instead: four packed adds per row.
            r[j] = rr2.x;
            r[j + 4] = rr2.y;
        }
        if (WANT_RESP) {
            // (a branch around a store makes the compiler wait for every outstanding load first; harmless here -- the next
            //  rows' loads were issued a whole row of arithmetic earlier)
            if constexpr (!RAG && RCV_HF_RESP_T) {
                // Round 6: the response row leaves the wave as WHOLE LINES.  A lane owns 8 pixels = 32 bytes of the f32 row; stored from there, each
                // of its two 16-byte stores writes every other 16 bytes of the wave's 2 KB -- two half-written visits to every 128-byte line, and a
                // launch time that depended on where the buffers lay (0.61-0.71 ms for one library on one box, `profiles/r06_harris_resp_stores.txt`;
                // the lesson of the one-launch config 3).  The row goes through 2 KB of wave-private LDS instead (no barrier: a wave's
                // LDS instructions execute in order) and comes back as float4 number `lane` and number 64 + `lane` of the strip's row.
                if ((unsigned)(u - ys) < (unsigned)(ye - ys)) {
                    float* const wl = hf_lds + 512 * (threadIdx.x >> 6);
                    if (lane >= 1 && lane <= 62) {
                        *(f4v*)(wl + 8 * (lane - 1)) = f4v{r[0], r[1], r[2], r[3]};
                        *(f4v*)(wl + 8 * (lane - 1) + 4) = f4v{r[4], r[5], r[6], r[7]};
                    }
                    const f4v o0 = *(const f4v*)(wl + 4 * lane), o1 = *(const f4v*)(wl + 256 + 4 * lane);
                    const int n4 = (min(a.cols, (strip + 1) * kStripPx) - strip * kStripPx) >> 2;   // float4s of the row inside this strip (cols % 8 == 0)
                    const uint32_t ro = (uint32_t)u * rstep32, vo = (uint32_t)(4 * strip * kStripPx + 16 * lane);
                    if (lane < n4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, o0), rrs, vo, ro, RCV_HF_RESP_AUX);
                    if (64 + lane < n4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, o1), rrs, vo + 1024, ro, RCV_HF_RESP_AUX);
                }
            } else if constexpr (!RAG) {
                if (live && u >= ys && u < ye) {
                    const uint32_t ro = (uint32_t)u * rstep32;
                    __builtin_amdgcn_raw_buffer_store_b128(u4v{__builtin_bit_cast(uint32_t, r[0]), __builtin_bit_cast(uint32_t, r[1]), __builtin_bit_cast(uint32_t, r[2]), __builtin_bit_cast(uint32_t, r[3])}, rrs, 4 * mx, ro, RCV_HF_RESP_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(u4v{__builtin_bit_cast(uint32_t, r[4]), __builtin_bit_cast(uint32_t, r[5]), __builtin_bit_cast(uint32_t, r[6]), __builtin_bit_cast(uint32_t, r[7])}, rrs, 4 * mx + 16, ro, RCV_HF_RESP_AUX);
                }
            } else if (live && u >= ys && u < ye) {
                gptr orow = (gptr)(rf + (size_t)u * a.rstep);
                asm("" : "+s"(orow));
                gptr o = orow + 4 * mx;
                if constexpr (RAG) {
                    if (nvalid == 8) {
                        *(RCV_GLOBAL F4m*)o = F4m{r[0], r[1], r[2], r[3]};
                        *(RCV_GLOBAL F4m*)(o + 16) = F4m{r[4], r[5], r[6], r[7]};
                    } else {   // the row's last, partial run: 4 + 2 + 1 samples as its length says
                        int j = 0;
                        if (nvalid & 4) {
                            *(RCV_GLOBAL F4m*)o = F4m{r[0], r[1], r[2], r[3]};
                            j = 4;
                        }
                        if (nvalid & 2) {
                            *(RCV_GLOBAL F2m*)(o + 4 * j) = j ? F2m{r[4], r[5]} : F2m{r[0], r[1]};
                            j += 2;
                        }
                        if (nvalid & 1) *(RCV_GLOBAL float*)(o + 4 * j) = j == 0 ? r[0] : (j == 2 ? r[2] : (j == 4 ? r[4] : r[6]));
                    }
                }
            }
        }
        if constexpr (!WANT_MASK) return;
        // ---- NMS: response outside the image is -inf ------------------------------------------------------------------
        // A row outside the image (scalar condition, two rows per frame edge): every response is -inf.  Columns outside the
        // image belong to whole lanes (cols % 8 == 0: the lane left of x = 0, lanes right of the last column) that never
        // store; only the values they hand to their neighbours matter, and those are replaced right here.
        if (!kLimRow && (unsigned)u >= (unsigned)a.rows) {
            // (a branch, not eight selects on every row: the compiler had if-converted this into a v_cndmask per pixel AND, no longer
            //  knowing the selected value canonical, a v_max_f32 x, x per pixel in front of the maxima below -- 2 of 29 instructions per
            //  pixel; round 4, from the ISA)
            asm volatile("; row outside the image: every response is -inf");
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = NEG_INF;
        }
        if (RAG) {   // columns right of the image inside the row's last, partial run
#pragma unroll
            for (int j = 1; j < 8; ++j)
                if (j >= nvalid) r[j] = NEG_INF;
        }
        // Round 5: THRESHOLD FIRST.  The mask of output row w = u - 1 is all zeros unless some pixel of row w reaches the threshold, and
        // on real scenes most rows of a 496-pixel strip hold no such pixel (scene family of bench.py, thr 1e-4: 0.34 % of the pixels,
        // 28 % of the (strip, row) pairs).  The window therefore keeps the RAW responses of rows u - 2 and u - 1 (16 registers; it kept
        // 32 of running maxima) and a wave-uniform flag per row -- 4 maxima, a compare and a scalar branch -- and forms the eight
        // neighbour maxima of row w only when its flag is set: 57 instructions on those rows (round 6: 47), 5 on the others, against 43 on every row.
        // keep = rc >= max(8 neighbours, thr_up) as before: the same maxima of the same values (round 6: grouped by column first, see `maxima`).
        if (has_edge) {   // (uniform) what the lane left of x = 0 and the lanes right of the image hand to their neighbours
            asm volatile("; strip with an edge lane: responses outside the image are -inf");
            if (edgeL) r[7] = NEG_INF;
            if (x >= a.cols) r[0] = NEG_INF;
        }
        const int cand_u = __builtin_amdgcn_ballot_w64(vmax3(vmax3(r[0], r[1], r[2]), vmax3(r[3], r[4], r[5]), vmax2(r[6], r[7])) >= thr_v) != 0ull ? 1 : 0;
        uint32_t mbits[2] = {0, 0};
        if (cand_b != 0) {   // (uniform) row w holds a candidate
            asm volatile("; row with a candidate: 3x3 maxima");
            const float bl = shr1f(rb[7]), br = shl1f(rb[0]);
            // USE_A / USE_C: the rows above / below row w = u - 1 take part.  kLimRow launches leave the responses of the two rows outside the image
            // as they come out of the arithmetic (finite values of mirrored pixels) and skip those rows here, where they would be read: the first /
            // last row of the image runs a variant of this block without them -- the same maxima as with -inf in their place.
            // Round 6: COLUMNS first -- v = max(above, below, thr) per pixel, then max(v[x-1], v[x], v[x+1], b[x-1], b[x+1]): 24 maxima and four
            // DPP moves per row where the row-wise grouping took 32 and six.  The mask bytes come from the SIGN of b - m (set iff b < m; b == m
            // gives +0; no NaN: b is finite or -inf, m >= thr_up > -inf): v_perm_b32's selectors 9 / 11 replicate bit 31 of either source
            // through a byte, so two differences become two bytes in one instruction -- 8 subtractions, 6 permutes and 2 complements per row
            // where compare + select + or took 20, and the four byte-mask registers of the selects are gone.
            auto maxima = [&](auto use_a, auto use_c) {
                constexpr bool USE_A = decltype(use_a)::value, USE_C = decltype(use_c)::value;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (USE_A && USE_C) ? vmax3(ra[j], r[j], thr_v) : vmax2(USE_A ? ra[j] : r[j], thr_v);
                const float vl = shr1f(v[7]), vr = shl1f(v[0]);
                uint32_t t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float hv = vmax3(j ? v[j - 1] : vl, v[j], j < 7 ? v[j + 1] : vr);
                    const float m = vmax3(hv, j ? rb[j - 1] : bl, j < 7 ? rb[j + 1] : br);
                    t[j] = __builtin_bit_cast(uint32_t, rb[j] - m);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t lo = pk(t[4 * h + 1], t[4 * h], 0x0c0c0b09u), hi = pk(t[4 * h + 3], t[4 * h + 2], 0x0b090c0cu);   // 0xff where b < m
                    mbits[h] = ~(lo | hi);
                }
            };
            if (kLimRow && u - 2 < 0) {
                asm volatile("; first row of the image: no row above");
                maxima(std::false_type{}, std::true_type{});
            } else if (kLimRow && u >= a.rows) {
                asm volatile("; last row of the image: no row below");
                maxima(std::true_type{}, std::false_type{});
            } else {
                maxima(std::true_type{}, std::true_type{});
            }
            if constexpr (!RAG) {
                pm0 = mbits[0];
                pm1 = mbits[1];
            }
        } else if (!RAG && pdirty != 0) {
            // (round 6) the pending mask registers are zero unless a row with candidates wrote them: they are cleared on the first row without
            // candidates after such a row, not on every row (two v_mov per row)
            asm volatile("; first row without candidates after one with: pending mask back to zeros");
            pm0 = 0;
            pm1 = 0;
        }
        if constexpr (!RAG) pdirty = cand_b;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ra[j] = rb[j];
            rb[j] = r[j];
        }
        cand_b = cand_u;
        const int w = u - 1;
        if constexpr (!RAG) {
            pw = w;
            return;
        }
        if (live && w >= ys && w < ye) {
            gptr mrow = (gptr)(mf + (size_t)w * a.mstep);
            asm("" : "+s"(mrow));
            if constexpr (RAG) {
                if (nvalid == 8) {
                    *(RCV_GLOBAL U2m*)(mrow + mx) = U2m{mbits[0], mbits[1]};
                } else {
                    int j = 0;
                    if (nvalid & 4) {
                        *(RCV_GLOBAL U1m*)(mrow + mx) = mbits[0];
                        j = 4;
                    }
                    const uint32_t rest = j ? mbits[1] : mbits[0];
                    if (nvalid & 2) *(RCV_GLOBAL H1m*)(mrow + mx + j) = (uint16_t)rest;
                    if (nvalid & 1) *(mrow + mx + j + (nvalid & 2)) = (uint8_t)((nvalid & 2) ? rest >> 16 : rest);
                }
            } else {
                *(RCV_GLOBAL u2v*)(mrow + mx) = u2v{mbits[0], mbits[1]};
            }
        }
    };

    // virtual gray rows v = ys-3 .. ye+1  (mask row w is emitted when v = w + 2 arrives)
    const int v0 = ys - 3, nrows = ye - ys + 5;
    // two row groups per trip, the two buffers swapping roles, so that no group is copied from `nxt` to `cur`; rows fed
    // past v = ye+1 (odd group count) re-read row ye+1 and write nothing (w >= ye)
    Row6 cur[kAhead], nxt[kAhead];
#pragma unroll
    for (int i = 0; i < kAhead; ++i) cur[i] = load_row(v0 + i);
    for (int g0 = 0; g0 < nrows; g0 += 2 * kAhead) {
#pragma unroll
        for (int i = 0; i < kAhead; ++i) nxt[i] = load_row(v0 + g0 + kAhead + i);
#pragma unroll
        for (int i = 0; i < kAhead; ++i) feed(cur[i], v0 + g0 + i);
#pragma unroll
        for (int i = 0; i < kAhead; ++i) cur[i] = load_row(v0 + g0 + 2 * kAhead + i);
#pragma unroll
        for (int i = 0; i < kAhead; ++i) feed(nxt[i], v0 + g0 + kAhead + i);
    }
    if constexpr (WANT_MASK && !RAG) flush_mask();
#ifdef RCV_HF_BENCH
    if (a.trace) {
        __builtin_amdgcn_s_waitcnt(0);   // (the wave's last store has left)
        if (lane == 0) {
            unsigned xcc, hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            a.trace[3 * (size_t)wid0] = t_start;
            a.trace[3 * (size_t)wid0 + 1] = __builtin_amdgcn_s_memrealtime();
            a.trace[3 * (size_t)wid0 + 2] = ((unsigned long long)(xcc & 0xf) << 32) | hwid;
        }
    }
#endif
}

#ifdef RCV_HF_BENCH
int g_hf_seg = 0;             // rows per segment (0: the product's plan)
void* g_hf_trace = nullptr;
#endif

} // namespace

#ifdef RCV_HF_BENCH   // (the measurement library carries its own copy under another name: the product's symbol stays the product's)
static int hf_launch(rcv_ctx* ctx, const View& s, const View* mask, const View* resp, int block, float k, float thr)
#else
int rcv_harris_fused(rcv_ctx* ctx, const View& s, const View* mask, const View* resp, int block, float k, float thr)
#endif
{
    if (block != 2 || (s.ch != 3 && s.ch != 2 && s.ch != 1)) return RCV_ERR_UNSUPPORTED;
    if (!mask && (!resp || s.ch != 1)) return RCV_ERR_UNSUPPORTED;   // response only: the gray-source cornerHarris
    View m0 = s;          // (placeholder when there is no mask: never dereferenced by the kernel)
    m0.p = nullptr;
    const View& m = mask ? *mask : m0;
    if (s.cols < 8 || s.rows < 4 || (s.ch == 2 && (s.cols & 1))) return RCV_ERR_UNSUPPORTED;
    // widths that are not a multiple of 8 and rows that are not 8 / 16-byte aligned: the RAG instantiations
    // (... and frames of 4 GB and more: the aligned instantiations address a frame's rows with 32-bit scalar offsets)
    const bool huge = (unsigned long long)s.rows * s.step > 0xffffffffull || (mask && (unsigned long long)m.rows * m.step > 0xffffffffull) ||
                      (resp && (unsigned long long)resp->rows * resp->step > 0xffffffffull);
    const bool rag = huge || s.cols % 8 != 0 || (uintptr_t)s.p % 8 || s.step % 8 || (s.n > 1 && s.fstride % 8) ||
                     (mask && ((uintptr_t)m.p % 8 || m.step % 8 || (m.n > 1 && m.fstride % 8))) ||
                     (resp && ((uintptr_t)resp->p % 16 || resp->step % 16 || (resp->n > 1 && resp->fstride % 16)));
    if (rag && resp && ((uintptr_t)resp->p % 4 || resp->step % 4 || resp->fstride % 4)) return RCV_ERR_UNSUPPORTED;
    HArgs a;
    a.src = s.p;
    a.mask = m.p;
    a.resp = resp ? resp->p : nullptr;
    a.sstep = s.step;
    a.mstep = m.step;
    a.rstep = resp ? resp->step : 0;
    a.sfs = s.fstride;
    a.mfs = m.fstride;
    a.rfs = resp ? resp->fstride : 0;
    a.rows = s.rows;
    a.cols = s.cols;
    a.nstrips = (s.cols + kStripPx - 1) / kStripPx;
    // row segments: whole rounds of the kernel's residency (register-limited; asked from the runtime once), >= 2 rounds,
    // segments >= 64 rows
    int seg = s.rows;
    {
        int& wpc = ctx->harris_wpc[resp ? 1 : 0];   // per context: contexts may be driven from different threads
        if (wpc == 0) {
            int nb = 0;
            hipError_t e = resp ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_harris_fused<true, 0>, 64 * kWPB, 0)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_harris_fused<false, 0>, 64 * kWPB, 0);
            wpc = (e == hipSuccess && nb > 0) ? kWPB * nb : 8;
            (void)hipGetLastError();
        }
        // Round 6 (tools/harris_timeline.py, profiles/r06_harris_timeline.txt): identical waves take 0.65 ... 1.25 of the median time (rows with
        // corners run the 3x3 maxima, the others do not), so whole "rounds" of the resident waves mean little; what a segment costs is its five
        // halo / fill rows + ~8 rows of set-up, what long segments cost is the ragged end of the launch: cost = (rows + 13) x (waves / slots +
        // 0.26).  64 x 4K: 135 rows (was 180: +1 %); a half of a split call plans for the pair of launches that share the chip (32 frames:
        // 135 rows, was 90: +1.5 %).
        const long long slots = (long long)wpc * ctx->cu_count, per_seg = (long long)a.nstrips * max(s.n, ctx->split_n);
        double best = 1e30;
        for (int ns = 1; ns <= 64; ++ns) {
            const int sr = (s.rows + ns - 1) / ns;
            if (ns > 1 && sr < 64) break;
            const long long tot = per_seg * ((s.rows + sr - 1) / sr);
            const double cost = (double)(sr + 13) * ((double)tot / (double)slots + 0.26);
            if (cost < best) { best = cost; seg = sr; }
        }
    }
    {   // small launches: rcv_plan_seg_rows (a segment streams 5 rows of halo / pipeline fill, ~8 more in set-up time)
        const int small = rcv_plan_seg_rows(s.rows, (long long)a.nstrips * s.n, ctx->cu_count, 13, 16);
        if (small > 0) seg = small;
    }
#ifdef RCV_HF_SEG   // (measurement builds: rows per segment fixed)
    seg = RCV_HF_SEG;
#endif
#ifdef RCV_HF_BENCH
    if (g_hf_seg > 0) seg = g_hf_seg;
    a.trace = (unsigned long long*)g_hf_trace;
#endif
    a.seg_rows = seg;
    a.nsegs = (s.rows + seg - 1) / seg;
    long long waves = (long long)a.nstrips * a.nsegs * s.n;
    if (waves > 0x3fffffff) return RCV_ERR_UNSUPPORTED;
    a.total_waves = (int)waves;
    double sc = 1.0 / (4.0 * 2.0 * 255.0);
    a.s2 = (float)(sc * sc);
    a.k = k;
    a.thr_up = thr != thr ? INFINITY : nextafterf(thr, INFINITY);
    // response-only launches (cornerHarris: 4 of 5 bytes per pixel are stores).  Rounds 2-5: untouched dynamic LDS held them at 3 workgroups per CU (0.698
    // against 0.803 ms).  Round 6: with whole-line non-temporal response stores the full occupancy is the faster one (0.51-0.56 against 0.55-0.59 ms):
    // RCV_HF_RESP_LDS = 0, the launch takes the 2 KB per wave of the row buffer only.
    constexpr unsigned kRespOnlyLds = RCV_HF_RESP_LDS * kWPB / 4;
    constexpr unsigned kRespLds = 2048 * kWPB;   // (aligned launches with the response: the row's way through LDS)
    static_assert(kRespOnlyLds == 0 || kRespOnlyLds >= kRespLds, "the response-only launch's LDS is also its row buffer");
    const long long nblocks = (waves + kWPB - 1) / kWPB;
    a.blocks_per_xcd = (int)((nblocks + 7) / 8);
    dim3 grid((unsigned)(a.blocks_per_xcd > 0 ? a.blocks_per_xcd * 8 : nblocks));
    if (rag) {
        if (s.ch == 1) {
            if (!mask) RCV_LAUNCH((k_harris_fused<true, 2, false, true>), grid, dim3(64 * kWPB), kRespOnlyLds, ctx->stream, a);
            else if (resp) RCV_LAUNCH((k_harris_fused<true, 2, true, true>), grid, dim3(64 * kWPB), 0, ctx->stream, a);
            else RCV_LAUNCH((k_harris_fused<false, 2, true, true>), grid, dim3(64 * kWPB), 0, ctx->stream, a);
        } else if (s.ch == 2) {
            if (resp) RCV_LAUNCH((k_harris_fused<true, 1, true, true>), grid, dim3(64 * kWPB), 0, ctx->stream, a);
            else RCV_LAUNCH((k_harris_fused<false, 1, true, true>), grid, dim3(64 * kWPB), 0, ctx->stream, a);
        } else {
            if (resp) RCV_LAUNCH((k_harris_fused<true, 0, true, true>), grid, dim3(64 * kWPB), 0, ctx->stream, a);
            else RCV_LAUNCH((k_harris_fused<false, 0, true, true>), grid, dim3(64 * kWPB), 0, ctx->stream, a);
        }
        return rcv_launch_check(ctx);
    }
    if (s.ch == 1) {
        if (!mask) RCV_LAUNCH((k_harris_fused<true, 2, false>), grid, dim3(64 * kWPB), kRespOnlyLds ? kRespOnlyLds : kRespLds, ctx->stream, a);
        else if (resp) RCV_LAUNCH((k_harris_fused<true, 2>), grid, dim3(64 * kWPB), kRespLds, ctx->stream, a);
        else RCV_LAUNCH((k_harris_fused<false, 2>), grid, dim3(64 * kWPB), 0, ctx->stream, a);
    } else if (s.ch == 2) {
        if (resp) RCV_LAUNCH((k_harris_fused<true, 1>), grid, dim3(64 * kWPB), kRespLds, ctx->stream, a);
        else RCV_LAUNCH((k_harris_fused<false, 1>), grid, dim3(64 * kWPB), 0, ctx->stream, a);
    } else {
        if (resp) RCV_LAUNCH((k_harris_fused<true, 0>), grid, dim3(64 * kWPB), kRespLds, ctx->stream, a);
        else RCV_LAUNCH((k_harris_fused<false, 0>), grid, dim3(64 * kWPB), 0, ctx->stream, a);
    }
    return rcv_launch_check(ctx);
}

#ifdef RCV_HF_BENCH
// Measurement entry (librustcv_hip_bench.so): the Harris pipeline BGR -> mask of a device-resident batch with the rows per segment given (0: the
// product's plan) and, if trace != nullptr, three 64-bit words per wave {start, end (100 MHz counter), XCD << 32 | HW_ID}; *waves = the launch's waves
extern "C" int rcv__harris_fused_bench(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* mask, float k, float thr, int seg_rows, void* trace, int* waves)
{
    RCV_TRY(rcv_bind(ctx));
    if (!src || !mask) return RCV_ERR_ARG;
    View s, m;
    RCV_TRY(rcv_view_batch(src, RCV_8U, &s));
    RCV_TRY(rcv_view_batch(mask, RCV_8U, &m));
    if (s.rows != m.rows || s.cols != m.cols || s.n != m.n || m.ch != 1) return RCV_ERR_ARG;
    g_hf_seg = seg_rows;
    g_hf_trace = trace;
    const int rc = hf_launch(ctx, s, &m, nullptr, 2, k, thr);
    g_hf_seg = 0;
    g_hf_trace = nullptr;
    if (waves) {
        const int nstrips = (s.cols + kStripPx - 1) / kStripPx;
        *waves = nstrips * s.n;   // (x segments: the caller knows seg_rows)
    }
    return rc;
}
#endif
