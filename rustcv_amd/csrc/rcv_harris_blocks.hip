// rcv_harris_blocks.hip -- cornerHarris response for ANY blockSize 1..7 from the Sobel planes Ix, Iy (i16), as a register
// sliding window down the image.  (blockSize 2 on aligned shapes runs the fully fused kernel of rcv_harris_fused.hip; this
// is what every other block size runs instead of the per-sample kernel k_harris_resp -- 6.4 ms against 0.66 ms on 64 4K frames
// before this kernel existed.)  Not in the reference; semantics SURVEY.md 8-A == oracle/rcv_oracle.c:
// Sxx, Sxy, Syy = exact i32 sums of the products over the block (anchor block/2, BORDER_REFLECT_101 of the product image),
// one (float) cast each, then the six separate IEEE f32 operations of the response.
//
// One WAVE owns a strip of 496 px (62 lanes x 8 px; lanes 0 / 63 carry halo only) and walks down a row segment, as the fused
// kernel does.  State per lane: the VERTICAL window sums V = sum over the block's rows of Ix^2, IxIy, Iy^2 for its 8 pixels
// (24 registers, i32).  Moving down one row adds the products of the row that enters the window and subtracts those of the
// row that leaves it, so the arithmetic per row does not depend on the block size (24-bit multiplies on the sign-extended i16
// values, exact).  The block's rows of Ix, Iy stay in a register ring (8 registers per row; the row loop is unrolled by the
// block size, so the ring slots are static): the first version re-loaded the leaving row instead, and since `block` rows of
// every wave's strip do not survive in L2 next to the response stream that doubled the plane reads (rocprofv3: 4.2 GB for
// 2.1 GB of planes).  The HORIZONTAL sums are sliding sums over V with the up to 3 pixels either side taken
// from the neighbouring lanes by DPP wave shifts.  Reflection: rows by reflected row index (scalar); columns through MARGINS of
// the planes -- the planes are this library's own workspace images, laid out with a margin either side of every row, and a tiny
// launch (k_mirror_margins) copies Ix, Iy of columns 1..3 to -1..-3 and of cols-2..cols-4 to cols..cols+2 (P(-j) = P(j) is a
// function of Ix(j), Iy(j), so mirroring the planes mirrors the products).  The window kernel then needs no edge cases at all:
// lanes left and right of the image read the margins, and a width that is not a multiple of 8 only shortens the last lane's
// store.
#include "rcv_internal.h"
#include "rcv_kernels.h"
#include "rcv_device_utils.h"
#include <math.h>

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int kStripPx = 62 * 8;

struct HBArgs {
    const uint8_t *ix, *iy;   // i16 planes
    uint8_t *resp, *mask;     // either may be null (MODE)
    size_t pstep, pfs, rstep, rfs, mstep, mfs;   // plane row step / frame stride (bytes), response and mask likewise
    int rows, cols, nstrips, seg_rows, nsegs, total_waves, blocks_per_xcd;
    float s2, k, thr_up;      // thr_up: smallest float > thr (+inf for a NaN threshold), as in rcv_harris_fused.hip
};

__device__ __forceinline__ int shr1i(int v) { return (int)__builtin_amdgcn_update_dpp(0u, (uint32_t)v, 0x138, 0xf, 0xf, true); }   // from lane-1
__device__ __forceinline__ int shl1i(int v) { return (int)__builtin_amdgcn_update_dpp(0u, (uint32_t)v, 0x130, 0xf, 0xf, true); }   // from lane+1

__device__ __forceinline__ float shr1f(float v) { return __builtin_bit_cast(float, shr1i(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ float shl1f(float v) { return __builtin_bit_cast(float, shl1i(__builtin_bit_cast(int, v))); }

__device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }   // v_mad_i32_i24: |Ix|, |Iy| <= 1020

// Blocks up to 4: block^2 * 1020^2 < 2^24, so the window sums (and every partial sum on the way) are integers f32 holds exactly.
// The gradients of a row are converted on the way in and on the way out and the sums run on packed f32 -- two pixels per
// instruction, pixels j and j+4 of the lane in one pair -- already in the form the response arithmetic wants (block 3, 4K batch
// of 64: 0.75 -> 0.66 ms from BGR, 0.64 -> 0.54 ms from gray; the i32 form stays for blocks 5..7).
template <int B>
struct WindowF32 {
    static constexpr int AN = B / 2, RT = B - 1 - AN;
    f2 xx[4], xy[4], yy[4];   // vertical sums of Ix*Ix, Ix*Iy, Iy*Iy of the lane's pixel pairs
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) xx[j] = xy[j] = yy[j] = f2{0.0f, 0.0f};
    }
    // gx, gy: the lane's 8 gradients as packed i16 pairs; leave = false: the row enters the window, true: it leaves
    __device__ __forceinline__ void account(const u4v& gx, const u4v& gy, bool leave)
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = j >> 1;
            f2 x, y;
            if (j & 1) {
                x = f2{(float)((int)gx[d] >> 16), (float)((int)gx[d + 2] >> 16)};
                y = f2{(float)((int)gy[d] >> 16), (float)((int)gy[d + 2] >> 16)};
            } else {
                x = f2{(float)(short)(gx[d] & 0xffff), (float)(short)(gx[d + 2] & 0xffff)};
                y = f2{(float)(short)(gy[d] & 0xffff), (float)(short)(gy[d + 2] & 0xffff)};
            }
            const f2 nx = leave ? -x : x, ny = leave ? -y : y;
            xx[j] = __builtin_elementwise_fma(nx, x, xx[j]);
            xy[j] = __builtin_elementwise_fma(nx, y, xy[j]);
            yy[j] = __builtin_elementwise_fma(ny, y, yy[j]);
        }
    }
    // horizontal sums of one plane: E[i] = columns (i - AN, i - AN + 4) of the lane, the outer ones from the neighbours by DPP
    __device__ __forceinline__ static void hsum(const f2 (&v)[4], f2 (&h)[4])
    {
        f2 E[B + 3];
#pragma unroll
        for (int j = 0; j < 4; ++j) E[AN + j] = v[j];
#pragma unroll
        for (int i = 0; i < AN; ++i) E[i] = f2{shr1f(v[4 - AN + i].y), v[4 - AN + i].x};
#pragma unroll
        for (int i = AN + 4; i < B + 3; ++i) E[i] = f2{v[i - AN - 4].y, shl1f(v[i - AN - 4].x)};
        if constexpr (B == 3) {
            const f2 t = E[1] + E[2], u = E[3] + E[4];
            h[0] = t + E[0];
            h[1] = t + E[3];
            h[2] = u + E[2];
            h[3] = u + E[5];
        } else {
            f2 s = E[0];
#pragma unroll
            for (int i = 1; i < B; ++i) s += E[i];
            h[0] = s;
#pragma unroll
            for (int xx = 1; xx < 4; ++xx) {
                s += E[xx + B - 1] - E[xx - 1];
                h[xx] = s;
            }
        }
    }
};

// columns -1..-3 := 1..3 and cols..cols+2 := cols-2..cols-4 of both planes (BORDER_REFLECT_101 of the product image, see above);
// one thread per (row, frame)
__global__ __launch_bounds__(256) void k_mirror_margins(uint8_t* ix, uint8_t* iy, size_t pstep, size_t pfs, int rows, int cols, int nrows_total)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nrows_total) return;
    const int frame = t / rows, y = t - frame * rows;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        int16_t* row = (int16_t*)((pl ? iy : ix) + (size_t)frame * pfs + (size_t)y * pstep);
#pragma unroll
        for (int j = 1; j <= 3; ++j) {
            row[-j] = row[min(j, cols - 1)];
            row[cols - 1 + j] = row[max(cols - 1 - j, 0)];
        }
    }
}

// MODE: 0 response only (cornerHarris), 1 mask only, 2 both (the Harris pipeline: the 3x3 NMS of rcv_harris_fused.hip on the
// response rows as they are formed -- the response of row y completes the mask of row y-1, so a segment computes the responses of
// rows ys-1 .. ye; responses outside the image are -inf)
template <int B, bool RAG, int MODE>
__global__ __launch_bounds__(256) void k_harris_resp_rows(HBArgs a)
{
    constexpr int AN = B / 2, RT = B - 1 - AN;   // window offsets -AN .. +RT
    constexpr bool WANT_RESP = MODE != 1, WANT_MASK = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) float hr_lds[];   // (aligned launches with the response: 2 KB per wave)
    const int lane = threadIdx.x & 63;
    const int blk = a.blocks_per_xcd > 0 ? (int)(blockIdx.x & 7) * a.blocks_per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    int wid = __builtin_amdgcn_readfirstlane(blk * 4 + (int)(threadIdx.x >> 6));
    if (wid >= a.total_waves) return;
    const int strip = wid % a.nstrips;
    wid /= a.nstrips;
    const int seg = wid % a.nsegs;
    const int frame = wid / a.nsegs;
    const int ys = seg * a.seg_rows, ye = min(a.rows, ys + a.seg_rows);
    const int x = strip * kStripPx + 8 * (lane - 1);
    const bool live = lane >= 1 && lane <= 62 && x < a.cols;
    const int nvalid = min(max(a.cols - x, 0), 8);   // < 8 only in the lane that holds the row's last, partial run
    // byte offset of the lane's 8 pixels in a plane row (the row pointer is column 0; the margins -- 8 pixels left, 16 right --
    // make -8 and the first multiple of 8 at or beyond cols valid places; lanes further right re-read that place: their values
    // feed no stored pixel)
    const int o1 = 2 * min(x, (a.cols + 7) & ~7);
    const uint8_t* const fx = a.ix + (size_t)frame * a.pfs;
    const uint8_t* const fy = a.iy + (size_t)frame * a.pfs;
    uint8_t* const rf = WANT_RESP ? a.resp + (size_t)frame * a.rfs : nullptr;
    uint8_t* const mf = WANT_MASK ? a.mask + (size_t)frame * a.mfs : nullptr;
    const float NEG_INF = -INFINITY;

    auto refl = [&](int v) { return v < 0 ? -v : (v >= a.rows ? 2 * a.rows - 2 - v : v); };   // (rows >= block: one reflection is enough)
    struct RowPix { u4v x, y; };   // the lane's 8 pixels of Ix and of Iy
    auto load_row = [&](int v) -> RowPix {   // virtual row -> reflected plane row
        const ptrdiff_t ro = (ptrdiff_t)refl(min(max(v, -AN - 2), a.rows + RT + 1)) * (ptrdiff_t)a.pstep + o1;   // (rows >= block >= ... see host: one reflection)
        return RowPix{*(const u4v*)(fx + ro), *(const u4v*)(fy + ro)};
    };

    constexpr bool F32SUM = B <= 4;
    WindowF32<B> wf;
    wf.clear();
    int vxx[8], vxy[8], vyy[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) vxx[j] = vxy[j] = vyy[j] = 0;
    // leave = false: the row enters the window (+products), true: it leaves (-products)
    auto accumulate_i = [&](const RowPix& w, bool leave) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int gx0 = (int)(short)(w.x[d] & 0xffff), gx1 = (int)w.x[d] >> 16, gy0 = (int)(short)(w.y[d] & 0xffff), gy1 = (int)w.y[d] >> 16;
            const int nx0 = leave ? -gx0 : gx0, nx1 = leave ? -gx1 : gx1, ny0 = leave ? -gy0 : gy0, ny1 = leave ? -gy1 : gy1;
            vxx[2 * d] = mad24(nx0, gx0, vxx[2 * d]);
            vxy[2 * d] = mad24(nx0, gy0, vxy[2 * d]);
            vyy[2 * d] = mad24(ny0, gy0, vyy[2 * d]);
            vxx[2 * d + 1] = mad24(nx1, gx1, vxx[2 * d + 1]);
            vxy[2 * d + 1] = mad24(nx1, gy1, vxy[2 * d + 1]);
            vyy[2 * d + 1] = mad24(ny1, gy1, vyy[2 * d + 1]);
        }
    };
    // horizontal sums of one plane: h[x] = sum of e[x - AN .. x + RT], e = the lane's 8 values with the neighbours' either side
    auto hsum = [&](const int (&v)[8], int (&h)[8]) {
        int e[8 + AN + RT];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[AN + j] = v[j];
#pragma unroll
        for (int j = 1; j <= AN; ++j) e[AN - j] = shr1i(v[8 - j]);
#pragma unroll
        for (int j = 1; j <= RT; ++j) e[AN + 7 + j] = shl1i(v[j - 1]);
        int s = e[0];
#pragma unroll
        for (int i = 1; i < B; ++i) s += e[i];
        h[0] = s;
#pragma unroll
        for (int xx = 1; xx < 8; ++xx) {
            s += e[xx + B - 1] - e[xx - 1];
            h[xx] = s;
        }
    };

    auto accumulate = [&](const RowPix& w, bool leave) {
        if constexpr (F32SUM) wf.account(w.x, w.y, leave);
        else accumulate_i(w, leave);
    };

    // NMS state: rowmax3 of response rows u-2, u-1; response and left/right max of row u-1 (as rcv_harris_fused.hip)
    float m3a[8], m3b[8], rc[8], mlr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m3a[j] = m3b[j] = rc[j] = mlr[j] = NEG_INF;

    // first / last response row of the segment (the mask needs one more either side)
    const int yf = WANT_MASK ? ys - 1 : ys, yl = WANT_MASK ? ye : ye - 1;
    // the window of response row yf: virtual rows yf-AN .. yf+RT in ring slots 0 .. B-1 (virtual row v lives in slot (v - yf + AN) % B)
    RowPix ring[B];
#pragma unroll
    for (int i = 0; i < B; ++i) ring[i] = load_row(yf - AN + i);
#pragma unroll
    for (int i = 0; i < B; ++i) accumulate(ring[i], false);
    RowPix ent = load_row(yf + RT + 1);
    for (int y0 = yf; y0 <= yl; y0 += B) {
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const int y = y0 + i;
            if (y > yl) break;
            // the row that moves the window to y+2 is in flight while row y is computed and the window moves to y+1
            const RowPix ent2 = load_row(y + RT + 2);
            f2 pxx[4], pxy[4], pyy[4];
            if constexpr (F32SUM) {
                wf.hsum(wf.xx, pxx);
                wf.hsum(wf.xy, pxy);
                wf.hsum(wf.yy, pyy);
            } else {
                int hxx[8], hxy[8], hyy[8];
                hsum(vxx, hxx);
                hsum(vxy, hxy);
                hsum(vyy, hyy);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pxx[j] = f2{(float)hxx[j], (float)hxx[j + 4]};
                    pxy[j] = f2{(float)hxy[j], (float)hxy[j + 4]};
                    pyy[j] = f2{(float)hyy[j], (float)hyy[j + 4]};
                }
            }
            float r[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {   // packed pairs {pixel j, pixel j+4}: two IEEE operations per instruction, same bits as the scalar ops
                const f2 fa = pxx[j] * a.s2, fb = pxy[j] * a.s2, fc = pyy[j] * a.s2;
                const f2 t1 = fa * fc, t2 = fb * fb, t3 = fa + fc;
                const f2 t4 = a.k * t3;
                const f2 t5 = t4 * t3;
                const f2 rr = (t1 - t2) - t5;
                r[j] = rr.x;
                r[j + 4] = rr.y;
            }
            if (WANT_RESP && y >= ys && y < ye) {
                uint8_t* o = rf + (size_t)y * a.rstep + 4 * (size_t)x;
                if constexpr (RAG) {
                  if (live) {   // response rows that are only 4-byte aligned; the row's last, partial run
                    typedef float f4m __attribute__((ext_vector_type(4), aligned(4)));
                    if (nvalid == 8) {
                        *(f4m*)o = f4m{r[0], r[1], r[2], r[3]};
                        *(f4m*)(o + 16) = f4m{r[4], r[5], r[6], r[7]};
                    } else {
                        for (int j = 0; j < nvalid; ++j) ((float*)o)[j] = r[j];
                    }
                  }
                } else {   // (whole lines through wave-private LDS: rcv_store_strip_row_f32)
                    const int n4 = (min(a.cols, (strip + 1) * kStripPx) - strip * kStripPx) >> 2;
                    rcv_store_strip_row_f32(hr_lds + 512 * (threadIdx.x >> 6), r, lane, rf + (size_t)y * a.rstep + 4 * (size_t)(strip * kStripPx), n4);
                }
            }
            if constexpr (WANT_MASK) {
                // responses outside the image are -inf: rows (scalar condition), columns left of 0 / right of cols-1 (per lane)
                const bool rowout = y < 0 || y >= a.rows;
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = (rowout || x + j < 0 || x + j >= a.cols) ? NEG_INF : r[j];
                const float rl = shr1f(r[7]), rr = shl1f(r[0]);   // r[x-1] of the lane's first pixel, r[x+8]
                uint32_t mbits[2] = {0, 0};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float left = j ? r[j - 1] : rl, right = j < 7 ? r[j + 1] : rr;
                    // the threshold rides in the neighbour maxima: rc > thr <=> rc >= thr_up, keep = rc >= max(8 neighbours, thr_up)
                    const float lrmax = fmaxf(fmaxf(left, right), a.thr_up);
                    const float m3 = fmaxf(lrmax, r[j]);
                    const float m8 = fmaxf(fmaxf(m3a[j], mlr[j]), m3);   // mask row y-1: rowmax3(y-2), left/right of y-1, rowmax3(y)
                    const bool keep = rc[j] >= m8;
                    mbits[j >> 2] |= keep ? (0xffu << ((j & 3) * 8)) : 0u;
                    m3a[j] = m3b[j];
                    m3b[j] = m3;
                    rc[j] = r[j];
                    mlr[j] = lrmax;
                }
                const int w = y - 1;
                if (live && w >= ys && w < ye) {
                    uint8_t* o = mf + (size_t)w * a.mstep + (size_t)x;
                    if constexpr (RAG) {
                        typedef uint32_t u2m __attribute__((ext_vector_type(2), aligned(1)));
                        if (nvalid == 8) *(u2m*)o = u2m{mbits[0], mbits[1]};
                        else
                            for (int j = 0; j < nvalid; ++j) o[j] = (uint8_t)(mbits[j >> 2] >> (8 * (j & 3)));
                    } else {
                        typedef uint32_t u2v __attribute__((ext_vector_type(2)));
                        *(u2v*)o = u2v{mbits[0], mbits[1]};
                    }
                }
            }
            // row y-AN (slot i) leaves, row y+RT+1 takes its slot
            accumulate(ring[i], true);
            ring[i] = ent;
            accumulate(ring[i], false);
            ent = ent2;
        }
    }
}

// ---- the same window, fed straight from the image: gray conversion and Sobel in front of it ---------------------------------
// k_harris_blocks_fused: aligned shapes (width a multiple of 8, 8-byte aligned source rows) of a BGR or a gray source run ONE
// launch for any block size: the front end of rcv_harris_fused.hip (8 pixels per lane, gray by two v_dot4 per pixel, Sobel in
// packed i16 with the two previous rows' horizontal parts in registers, neighbours by DPP) produces the row of Ix, Iy that
// enters the window, so the i16 planes (8 of the 13 / 12 bytes the two-launch path moves per pixel) never exist.  Rows outside
// the image: the gray rows are fed by REFLECTED row index, which makes the Sobel of a virtual row u the mirror image of row -u
// (Ix equal, Iy negated: negated back here), i.e. P(u) = P(-u) -- the box filter's reflection of the product image.  Columns:
// the lane left of the image takes Ix, Iy of columns 1..3 from its right neighbour, the lane right of it those of
// cols-2..cols-4 from its left neighbour (first / last strip only).
typedef short s2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t hpk(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ uint32_t hpk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2v, a) - __builtin_bit_cast(s2v, b)); }
__device__ __forceinline__ uint32_t hpk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2v, a) + __builtin_bit_cast(s2v, b)); }
__device__ __forceinline__ uint32_t hpk_add2x(uint32_t a, uint32_t b)   // a + 2 * b on both halves, one instruction (as rcv_harris_fused.hip)
{
    uint32_t d;
    asm("v_pk_mad_i16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(b), "v"(a));
    return d;
}
__device__ __forceinline__ uint32_t shr1u(uint32_t v) { return __builtin_amdgcn_update_dpp(0u, v, 0x138, 0xf, 0xf, true); }   // from lane-1
__device__ __forceinline__ uint32_t shl1u(uint32_t v) { return __builtin_amdgcn_update_dpp(0u, v, 0x130, 0xf, 0xf, true); }   // from lane+1

struct HFBArgs {
    const uint8_t* src;
    uint8_t *resp, *mask;
    size_t sstep, sfs, rstep, rfs, mstep, mfs;
    int rows, cols, nstrips, seg_rows, nsegs, total_waves, blocks_per_xcd;
    float s2, k, thr_up;
};

// SRCK: 0 BGR, 2 gray.  MODE: 0 response only, 1 mask only, 2 both.
#ifndef RCV_HB_OCC
#define RCV_HB_OCC 0   // (3: forced to 168 registers the block-3 launches spill and take 0.81 instead of 0.66 ms)
#endif
#if RCV_HB_OCC > 0
#define RCV_HB_OCC_ATTR __attribute__((amdgpu_waves_per_eu(RCV_HB_OCC)))
#else
#define RCV_HB_OCC_ATTR
#endif
template <int B, int SRCK, int MODE>
__global__ __launch_bounds__(256) RCV_HB_OCC_ATTR void k_harris_blocks_fused(HFBArgs a)
{
    constexpr int AN = B / 2, RT = B - 1 - AN;
    constexpr bool WANT_RESP = MODE != 1, WANT_MASK = MODE != 0, GRAY = SRCK == 2;
    extern __shared__ __attribute__((aligned(16))) float hb_lds[];   // (launches with the response: 2 KB per wave, see the response store)
    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const int blk = a.blocks_per_xcd > 0 ? (int)(blockIdx.x & 7) * a.blocks_per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    int wid = __builtin_amdgcn_readfirstlane(blk * 4 + (int)(threadIdx.x >> 6));
    if (wid >= a.total_waves) return;
    const int strip = wid % a.nstrips;
    wid /= a.nstrips;
    const int seg = wid % a.nsegs;
    const int frame = wid / a.nsegs;
    const int ys = seg * a.seg_rows, ye = min(a.rows, ys + a.seg_rows);
    const int x = strip * kStripPx + 8 * (lane - 1);
    const int xc = min(max(x, 0), a.cols - 8);
    const bool edgeL = x < 0, edgeR = x == a.cols;
    const bool edge_wave = strip == 0 || strip == a.nstrips - 1;   // wave-uniform
    const bool live = lane >= 1 && lane <= 62 && x < a.cols;
    // (round 6, as rcv_harris_fused.hip) BUFFER loads / stores: the frame base in the resource, the row's byte offset in the instruction's scalar offset, the
    // lane's column offset in a 32-bit vector offset -- no 64-bit per-lane pointers and row sums in vector registers.  Frames of 4 GB and more: the host takes
    // the two-launch path.
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src + (size_t)frame * a.sfs), 0, 0xffffffff, 0x00020000);
    const uint32_t sx = (uint32_t)((GRAY ? 1 : 3) * xc), sstep32 = (uint32_t)a.sstep;
    uint8_t* const rf = WANT_RESP ? a.resp + (size_t)frame * a.rfs : nullptr;
    uint8_t* const mf = WANT_MASK ? a.mask + (size_t)frame * a.mfs : nullptr;
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void*)mf, 0, 0xffffffff, 0x00020000);
    const float NEG_INF = -INFINITY;

    struct Raw { uint32_t d[GRAY ? 2 : 6]; };
    auto load_row = [&](int v) -> Raw {   // virtual gray row -> reflected source row (host: rows >= 8, one reflection is enough)
        v = min(max(v, -6), a.rows + 5);
        const int r = v < 0 ? -v : (v >= a.rows ? 2 * a.rows - 2 - v : v);
        const uint32_t so = (uint32_t)r * sstep32;
        Raw w;
        if constexpr (GRAY) {
            const u2v q0 = __builtin_amdgcn_raw_buffer_load_b64(srs, sx, so, 0);
            w.d[0] = q0.x; w.d[1] = q0.y;
        } else {
            const u4v q0 = __builtin_amdgcn_raw_buffer_load_b128(srs, sx, so, 0);
            const u2v q2 = __builtin_amdgcn_raw_buffer_load_b64(srs, sx + 16, so, 0);
            w.d[0] = q0.x; w.d[1] = q0.y; w.d[2] = q0.z; w.d[3] = q0.w; w.d[4] = q2.x; w.d[5] = q2.y;
        }
        return w;
    };
    struct RowPix { u4v x, y; };   // Ix and Iy of the lane's 8 pixels, packed i16 pairs
    uint32_t h1a[4], h1b[4], h2a[4], h2b[4];   // Sobel horizontal parts of gray rows v-2, v-1
#pragma unroll
    for (int j = 0; j < 4; ++j) h1a[j] = h1b[j] = h2a[j] = h2b[j] = 0;
    // gray row v in, gradient row u = v-1 out (valid from the third row fed)
    auto feed = [&](const Raw& q, int v) -> RowPix {
        uint32_t L[5], Cc[4];   // zero-extended gray pairs (g[2j-1], g[2j]) and (g[2j], g[2j+1])
        if constexpr (GRAY) {
            uint32_t lo = q.d[0], hi = q.d[1];
            if (edgeL) hi = hpk(lo, hi, 0x05020100u);   // x = -1 mirrors x = 1
            if (edgeR) lo = hpk(hi, lo, 0x03020106u);   // x = cols mirrors cols-2
            const uint32_t lf = shr1u(hi), rt = shl1u(lo);
            L[0] = hpk(lf, lo, 0x0c000c07u);
            L[1] = hpk(lo, lo, 0x0c020c01u);
            L[2] = hpk(hi, lo, 0x0c040c03u);
            L[3] = hpk(hi, hi, 0x0c020c01u);
            L[4] = hpk(rt, hi, 0x0c040c03u);
            Cc[0] = hpk(lo, lo, 0x0c010c00u);
            Cc[1] = hpk(lo, lo, 0x0c030c02u);
            Cc[2] = hpk(hi, hi, 0x0c010c00u);
            Cc[3] = hpk(hi, hi, 0x0c030c02u);
        } else {
            // gray with the weights times 4 (7472, 38468, 19596 = 256*{29,150,76} + {48,68,140}, rounding term 4*8192): the value
            // lands in bits 16..23, a whole byte the pair permutes read in place -- no shift, the eight values are never packed
            // (rcv_harris_fused.hip, same front end)
            auto gray_at_byte2 = [](uint32_t px) -> uint32_t {   // (B,G,R,x) dword -> RCV_BGR2GRAY << 16 (+ low bits)
                const uint32_t hi8 = __builtin_amdgcn_udot4(px, 0x004c961du, 0u, false);
                const uint32_t lo8 = __builtin_amdgcn_udot4(px, 0x008c4430u, 32768u, false);
                return (hi8 << 8) + lo8;
            };
            uint32_t g[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k0 = 3 * j, w0 = k0 >> 2, sh = k0 & 3;
                g[j] = gray_at_byte2(sh == 0 ? q.d[w0] : __builtin_amdgcn_alignbyte(q.d[w0 + 1 < 6 ? w0 + 1 : 5], q.d[w0], sh));
            }
            if (edgeL) g[7] = g[1];
            if (edgeR) g[0] = g[6];
            const uint32_t lf = shr1u(g[7]), rt = shl1u(g[0]);
            constexpr uint32_t kPair = 0x0c060c02u;
            L[0] = hpk(g[0], lf, kPair);
            L[1] = hpk(g[2], g[1], kPair);
            L[2] = hpk(g[4], g[3], kPair);
            L[3] = hpk(g[6], g[5], kPair);
            L[4] = hpk(rt, g[7], kPair);
#pragma unroll
            for (int j = 0; j < 4; ++j) Cc[j] = hpk(g[2 * j + 1], g[2 * j], kPair);
        }
        const int u = v - 1;
        const bool mirrored = u < 0 || u >= a.rows;   // scalar: the window of a virtual row is the mirror image of row -u: dy changes sign
        uint32_t ox[4], oy[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t h1 = hpk_sub(L[j + 1], L[j]);
            const uint32_t h2 = hpk_add2x(hpk_add(L[j], L[j + 1]), Cc[j]);
            ox[j] = hpk_add2x(hpk_add(h1a[j], h1), h1b[j]);
            oy[j] = hpk_sub(h2, h2a[j]);
            h1a[j] = h1b[j];
            h1b[j] = h1;
            h2a[j] = h2b[j];
            h2b[j] = h2;
        }
        if (mirrored) {
#pragma unroll
            for (int j = 0; j < 4; ++j) oy[j] = hpk_sub(0u, oy[j]);
        }
        if (edge_wave) {
            // P(-j) = P(j), P(cols-1+j) = P(cols-1-j): the gradients of the mirrored columns from the neighbouring lane
            const uint32_t tx0 = shl1u(ox[0]), tx1 = shl1u(ox[1]), ty0 = shl1u(oy[0]), ty1 = shl1u(oy[1]);   // lane+1: columns 0..3
            const uint32_t sx2 = shr1u(ox[2]), sx3 = shr1u(ox[3]), sy2 = shr1u(oy[2]), sy3 = shr1u(oy[3]);   // lane-1: its columns 4..7
            if (edgeL) {   // logical columns -3, -2, -1 (pixels 5, 6, 7 of this lane) := 3, 2, 1
                ox[2] = tx1;
                ox[3] = hpk(tx0, tx1, 0x07060100u);
                oy[2] = ty1;
                oy[3] = hpk(ty0, ty1, 0x07060100u);
            }
            if (edgeR) {   // logical columns cols, cols+1, cols+2 (pixels 0, 1, 2) := cols-2, cols-3, cols-4
                ox[0] = hpk(sx2, sx3, 0x07060100u);
                ox[1] = sx2;
                oy[0] = hpk(sy2, sy3, 0x07060100u);
                oy[1] = sy2;
            }
        }
        return RowPix{u4v{ox[0], ox[1], ox[2], ox[3]}, u4v{oy[0], oy[1], oy[2], oy[3]}};
    };

    constexpr bool F32SUM = B <= 4;
    WindowF32<B> wf;
    wf.clear();
    int vxx[8], vxy[8], vyy[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) vxx[j] = vxy[j] = vyy[j] = 0;
    auto accumulate = [&](const RowPix& w, bool leave) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int gx0 = (int)(short)(w.x[d] & 0xffff), gx1 = (int)w.x[d] >> 16, gy0 = (int)(short)(w.y[d] & 0xffff), gy1 = (int)w.y[d] >> 16;
            const int nx0 = leave ? -gx0 : gx0, nx1 = leave ? -gx1 : gx1, ny0 = leave ? -gy0 : gy0, ny1 = leave ? -gy1 : gy1;
            vxx[2 * d] = mad24(nx0, gx0, vxx[2 * d]);
            vxy[2 * d] = mad24(nx0, gy0, vxy[2 * d]);
            vyy[2 * d] = mad24(ny0, gy0, vyy[2 * d]);
            vxx[2 * d + 1] = mad24(nx1, gx1, vxx[2 * d + 1]);
            vxy[2 * d + 1] = mad24(nx1, gy1, vxy[2 * d + 1]);
            vyy[2 * d + 1] = mad24(ny1, gy1, vyy[2 * d + 1]);
        }
    };
    auto hsum = [&](const int (&v)[8], int (&h)[8]) {
        int e[8 + AN + RT];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[AN + j] = v[j];
#pragma unroll
        for (int j = 1; j <= AN; ++j) e[AN - j] = shr1i(v[8 - j]);
#pragma unroll
        for (int j = 1; j <= RT; ++j) e[AN + 7 + j] = shl1i(v[j - 1]);
        int s = e[0];
#pragma unroll
        for (int i = 1; i < B; ++i) s += e[i];
        h[0] = s;
#pragma unroll
        for (int xx = 1; xx < 8; ++xx) {
            s += e[xx + B - 1] - e[xx - 1];
            h[xx] = s;
        }
    };
    float m3a[8], m3b[8], rc[8], mlr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m3a[j] = m3b[j] = rc[j] = mlr[j] = NEG_INF;

    const int yf = WANT_MASK ? ys - 1 : ys, yl = WANT_MASK ? ye : ye - 1;
    // gradient rows come out of `advance` in order, starting two rows early (the Sobel state has to fill): the third one is row
    // yf-AN, the first row of the window; two source rows are kept in flight
    int vn = yf - AN - 1;
    Raw r0 = load_row(vn), r1 = load_row(vn + 1);
    auto advance = [&]() -> RowPix {
        const Raw cur = r0;
        r0 = r1;
        r1 = load_row(vn + 2);
        const RowPix g = feed(cur, vn);
        ++vn;
        return g;
    };
    (void)advance();
    (void)advance();
    // the ring keeps the rows as packed i16 (8 registers a row) and converts on the way in and on the way out: holding them as
    // f32 costs 16 registers a row and one wave per SIMD (measured: B=4 response 0.65 -> 0.76 ms, nothing gained elsewhere)
    typedef RowPix Row;
    auto next_row = [&]() -> Row { return advance(); };
    auto account = [&](const Row& w, bool leave) {
        if constexpr (F32SUM) wf.account(w.x, w.y, leave);
        else accumulate(w, leave);
    };
    Row ring[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
        ring[i] = next_row();
        account(ring[i], false);
    }
    Row ent = next_row();   // row yf+RT+1
    // the Sobel rows, the source rows in flight and the NMS rows rotate with period 2, the ring with period B: B=1 is unrolled
    // twice so that no state is copied at the back edge (-8..-14 %); B=3 unrolled six times is slower than three (+5..+10 %)
    constexpr int U = B == 1 ? 2 : B;
    for (int y0 = yf; y0 <= yl; y0 += U) {
#pragma unroll
        for (int iu = 0; iu < U; ++iu) {
            const int y = y0 + iu, i = iu % B;
            if (y > yl) break;
            f2 pxx[4], pxy[4], pyy[4];
            if constexpr (F32SUM) {
                wf.hsum(wf.xx, pxx);
                wf.hsum(wf.xy, pxy);
                wf.hsum(wf.yy, pyy);
            } else {
                int hxx[8], hxy[8], hyy[8];
                hsum(vxx, hxx);
                hsum(vxy, hxy);
                hsum(vyy, hyy);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pxx[j] = f2{(float)hxx[j], (float)hxx[j + 4]};
                    pxy[j] = f2{(float)hxy[j], (float)hxy[j + 4]};
                    pyy[j] = f2{(float)hyy[j], (float)hyy[j + 4]};
                }
            }
            float r[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f2 fa = pxx[j] * a.s2, fb = pxy[j] * a.s2, fc = pyy[j] * a.s2;
                const f2 t1 = fa * fc, t2 = fb * fb, t3 = fa + fc;
                const f2 t4 = a.k * t3;
                const f2 t5 = t4 * t3;
                const f2 rr = (t1 - t2) - t5;
                r[j] = rr.x;
                r[j + 4] = rr.y;
            }
            if (WANT_RESP && y >= ys && y < ye) {   // (whole lines through wave-private LDS: rcv_store_strip_row_f32)
                const int n4 = (min(a.cols, (strip + 1) * kStripPx) - strip * kStripPx) >> 2;
                rcv_store_strip_row_f32(hb_lds + 512 * (threadIdx.x >> 6), r, lane, rf + (size_t)y * a.rstep + 4 * (size_t)(strip * kStripPx), n4);
            }
            if constexpr (WANT_MASK) {
                const bool rowout = y < 0 || y >= a.rows;
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = (rowout || x < 0 || x >= a.cols) ? NEG_INF : r[j];   // (cols % 8 == 0: whole lanes)
                const float rl = shr1f(r[7]), rr = shl1f(r[0]);
                // (round 6, as rcv_harris_fused.hip) the mask bytes from the SIGN of centre - maximum (set iff the centre is smaller; equal gives +0; no NaN:
                // the centre is finite or -inf, the maximum >= thr_up > -inf): v_perm_b32's selectors 9 / 11 replicate bit 31 of either source through a
                // byte -- and the four byte-mask registers of compare + select + or are gone (the mask-only block-3 launch: 174 -> 166 registers = three
                // waves per SIMD instead of two)
                uint32_t t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float left = j ? r[j - 1] : rl, right = j < 7 ? r[j + 1] : rr;
                    const float lrmax = fmaxf(fmaxf(left, right), a.thr_up);
                    const float m3 = fmaxf(lrmax, r[j]);
                    const float m8 = fmaxf(fmaxf(m3a[j], mlr[j]), m3);
                    t[j] = __builtin_bit_cast(uint32_t, rc[j] - m8);
                    m3a[j] = m3b[j];
                    m3b[j] = m3;
                    rc[j] = r[j];
                    mlr[j] = lrmax;
                }
                uint32_t mbits[2];
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    mbits[h] = ~(hpk(t[4 * h + 1], t[4 * h], 0x0c0c0b09u) | hpk(t[4 * h + 3], t[4 * h + 2], 0x0b090c0cu));
                const int w = y - 1;
                if (live && w >= ys && w < ye) __builtin_amdgcn_raw_buffer_store_b64(u2v{mbits[0], mbits[1]}, mrs, (uint32_t)x, (uint32_t)w * (uint32_t)a.mstep, 0);
            }
            account(ring[i], true);
            ring[i] = ent;
            account(ring[i], false);
            ent = next_row();
        }
    }
}

template <int B, int SRCK>
void launch_fused(const HFBArgs& a, dim3 grid, hipStream_t st)
{
    const int mode = a.mask ? (a.resp ? 2 : 1) : 0;
    if (mode == 0) RCV_LAUNCH((k_harris_blocks_fused<B, SRCK, 0>), grid, dim3(256), 8192, st, a);   // (8 KB: the response rows' way through LDS)
    else if (mode == 1) RCV_LAUNCH((k_harris_blocks_fused<B, SRCK, 1>), grid, dim3(256), 0, st, a);
    else RCV_LAUNCH((k_harris_blocks_fused<B, SRCK, 2>), grid, dim3(256), 8192, st, a);
}

template <int B>
void launch_resp(const HBArgs& a, dim3 grid, bool rag, hipStream_t st)
{
    const int mode = a.mask ? (a.resp ? 2 : 1) : 0;
    if (rag) {
        if (mode == 0) RCV_LAUNCH((k_harris_resp_rows<B, true, 0>), grid, dim3(256), 0, st, a);
        else if (mode == 1) RCV_LAUNCH((k_harris_resp_rows<B, true, 1>), grid, dim3(256), 0, st, a);
        else RCV_LAUNCH((k_harris_resp_rows<B, true, 2>), grid, dim3(256), 0, st, a);
    } else {
        if (mode == 0) RCV_LAUNCH((k_harris_resp_rows<B, false, 0>), grid, dim3(256), 8192, st, a);
        else if (mode == 1) RCV_LAUNCH((k_harris_resp_rows<B, false, 1>), grid, dim3(256), 0, st, a);
        else RCV_LAUNCH((k_harris_resp_rows<B, false, 2>), grid, dim3(256), 8192, st, a);
    }
}

} // namespace

// Layout of the Sobel planes this kernel reads: 8 pixels of margin left and 16 right of every row, rows 16-byte aligned.
size_t rcv_harris_plane_step(int cols) { return ((size_t)(cols + 24) * 2 + 15) & ~(size_t)15; }
size_t rcv_harris_plane_margin() { return 16; }   // bytes in front of column 0

// Does k_harris_resp_rows take an output image of this shape?  Any width >= 8, at least block + 2 rows (one reflection per row
// index is enough then), 4-byte aligned response rows (f32); mask rows may have any alignment.
bool rcv_harris_resp_rows_ok(const View& o, int block, bool is_resp)
{
    if (block < 1 || block > 7) return false;
    if (o.cols < 8 || o.rows < block + 2) return false;
    return !is_resp || !((uintptr_t)o.p % 4 || o.step % 4 || (o.n > 1 && o.fstride % 4));
}

// Response (r, may be null) and / or NMS mask (m, may be null; thr: its threshold) from the Sobel planes for any block 1..7.
// ix / iy: views of column 0 of planes laid out as above (the margins are filled here); RCV_ERR_UNSUPPORTED for shapes it does
// not take (per-sample kernels).
int rcv_harris_resp_rows(rcv_ctx* ctx, const View& ix, const View& iy, const View* r, const View* m, int block, float k, float thr)
{
    if (!r && !m) return RCV_ERR_ARG;
    const View& o = r ? *r : *m;
    if (r && !rcv_harris_resp_rows_ok(*r, block, true)) return RCV_ERR_UNSUPPORTED;
    if (m && !rcv_harris_resp_rows_ok(*m, block, false)) return RCV_ERR_UNSUPPORTED;
    if (r && m && (r->rows != m->rows || r->cols != m->cols || r->n != m->n)) return RCV_ERR_ARG;
    if (ix.step != iy.step || ix.fstride != iy.fstride || ix.step % 16 || (uintptr_t)ix.p % 16 || (uintptr_t)iy.p % 16 || (ix.n > 1 && ix.fstride % 16)) return RCV_ERR_UNSUPPORTED;
    if (ix.step < rcv_harris_plane_step(o.cols) || (long long)o.rows * o.n > 0x7fffffff) return RCV_ERR_UNSUPPORTED;
    {
        const int total = o.rows * o.n;
        RCV_LAUNCH(k_mirror_margins, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, ix.p, iy.p, ix.step, ix.fstride, o.rows, o.cols, total);
    }
    const bool rag = o.cols % 8 != 0 || (r && ((uintptr_t)r->p % 16 || r->step % 16 || (r->n > 1 && r->fstride % 16))) ||
                     (m && ((uintptr_t)m->p % 8 || m->step % 8 || (m->n > 1 && m->fstride % 8)));
    HBArgs a;
    a.ix = ix.p;
    a.iy = iy.p;
    a.resp = r ? r->p : nullptr;
    a.mask = m ? m->p : nullptr;
    a.pstep = ix.step;
    a.pfs = ix.fstride;
    a.rstep = r ? r->step : 0;
    a.rfs = r ? r->fstride : 0;
    a.mstep = m ? m->step : 0;
    a.mfs = m ? m->fstride : 0;
    a.rows = o.rows;
    a.cols = o.cols;
    a.thr_up = thr != thr ? INFINITY : nextafterf(thr, INFINITY);
    a.nstrips = (o.cols + kStripPx - 1) / kStripPx;
    int seg = o.rows;
    while ((long long)a.nstrips * ((o.rows + seg - 1) / seg) * o.n < 8192 && seg > 48) seg = (seg + 1) / 2;
    a.seg_rows = seg;
    a.nsegs = (o.rows + seg - 1) / seg;
    const long long waves = (long long)a.nstrips * a.nsegs * o.n;
    if (waves > 0x3fffffff) return RCV_ERR_UNSUPPORTED;
    a.total_waves = (int)waves;
    const long long nblocks = (waves + 3) / 4;
    a.blocks_per_xcd = (int)((nblocks + 7) / 8);
    const dim3 grid((unsigned)(a.blocks_per_xcd > 0 ? a.blocks_per_xcd * 8 : nblocks));
    const double s = 1.0 / (4.0 * (double)block * 255.0);   // 2^(aperture-1) * blockSize * 255, aperture = 3
    a.s2 = (float)(s * s);
    a.k = k;
    switch (block) {
    case 1: launch_resp<1>(a, grid, rag, ctx->stream); break;
    case 2: launch_resp<2>(a, grid, rag, ctx->stream); break;
    case 3: launch_resp<3>(a, grid, rag, ctx->stream); break;
    case 4: launch_resp<4>(a, grid, rag, ctx->stream); break;
    case 5: launch_resp<5>(a, grid, rag, ctx->stream); break;
    case 6: launch_resp<6>(a, grid, rag, ctx->stream); break;
    default: launch_resp<7>(a, grid, rag, ctx->stream); break;
    }
    return rcv_launch_check(ctx);
}

// One launch for any block size on aligned shapes: BGR or gray source with 8-byte aligned rows, width a multiple of 8, >= 16
// columns, at least max(block + 4, 8) rows; 16-byte aligned response rows, 8-byte aligned mask rows.  r and / or m (either may
// be null).  RCV_ERR_UNSUPPORTED otherwise (the two-launch path through i16 planes takes ragged shapes).
int rcv_harris_blocks_fused(rcv_ctx* ctx, const View& s, const View* r, const View* m, int block, float k, float thr)
{
    if (block < 1 || block > 7 || (!r && !m)) return RCV_ERR_UNSUPPORTED;
    if (s.ch != 1 && s.ch != 3) return RCV_ERR_UNSUPPORTED;
    if (s.cols % 8 != 0 || s.cols < 16 || s.rows < block + 4 || s.rows < 8) return RCV_ERR_UNSUPPORTED;
    if ((uintptr_t)s.p % 8 || s.step % 8 || (s.n > 1 && s.fstride % 8)) return RCV_ERR_UNSUPPORTED;
    if (r && ((uintptr_t)r->p % 16 || r->step % 16 || (r->n > 1 && r->fstride % 16))) return RCV_ERR_UNSUPPORTED;
    if (m && ((uintptr_t)m->p % 8 || m->step % 8 || (m->n > 1 && m->fstride % 8))) return RCV_ERR_UNSUPPORTED;
    HFBArgs a;
    a.src = s.p;
    a.resp = r ? r->p : nullptr;
    a.mask = m ? m->p : nullptr;
    a.sstep = s.step;
    a.sfs = s.fstride;
    a.rstep = r ? r->step : 0;
    a.rfs = r ? r->fstride : 0;
    a.mstep = m ? m->step : 0;
    a.mfs = m ? m->fstride : 0;
    a.rows = s.rows;
    a.cols = s.cols;
    a.nstrips = (s.cols + kStripPx - 1) / kStripPx;
    int seg = s.rows;
    while ((long long)a.nstrips * ((s.rows + seg - 1) / seg) * s.n < 8192 && seg > 96) seg = (seg + 1) / 2;
    {   // small launches: rcv_plan_seg_rows (block + 3 rows of pipeline fill, + set-up)
        const int small = rcv_plan_seg_rows(s.rows, (long long)a.nstrips * s.n, ctx->cu_count, block + 10, 16);
        if (small > 0) seg = small;
    }
    a.seg_rows = seg;
    a.nsegs = (s.rows + seg - 1) / seg;
    const long long waves = (long long)a.nstrips * a.nsegs * s.n;
    if (waves > 0x3fffffff) return RCV_ERR_UNSUPPORTED;
    a.total_waves = (int)waves;
    const long long nblocks = (waves + 3) / 4;
    a.blocks_per_xcd = (int)((nblocks + 7) / 8);
    const dim3 grid((unsigned)(a.blocks_per_xcd > 0 ? a.blocks_per_xcd * 8 : nblocks));
    const double sc = 1.0 / (4.0 * (double)block * 255.0);
    a.s2 = (float)(sc * sc);
    a.k = k;
    a.thr_up = thr != thr ? INFINITY : nextafterf(thr, INFINITY);
#define RCV_HB_CASE(BB)                                                         \
    case BB:                                                                    \
        if (s.ch == 1) launch_fused<BB, 2>(a, grid, ctx->stream);               \
        else launch_fused<BB, 0>(a, grid, ctx->stream);                         \
        break;
    switch (block) {
        RCV_HB_CASE(1) RCV_HB_CASE(2) RCV_HB_CASE(3) RCV_HB_CASE(4) RCV_HB_CASE(5) RCV_HB_CASE(6)
    default:
        if (s.ch == 1) launch_fused<7, 2>(a, grid, ctx->stream);
        else launch_fused<7, 0>(a, grid, ctx->stream);
        break;
    }
#undef RCV_HB_CASE
    return rcv_launch_check(ctx);
}
