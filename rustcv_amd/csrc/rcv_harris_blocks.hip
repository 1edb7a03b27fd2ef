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

    int vxx[8], vxy[8], vyy[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) vxx[j] = vxy[j] = vyy[j] = 0;
    // leave = false: the row enters the window (+products), true: it leaves (-products)
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
            int hxx[8], hxy[8], hyy[8];
            hsum(vxx, hxx);
            hsum(vxy, hxy);
            hsum(vyy, hyy);
            float r[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {   // packed pairs {pixel j, pixel j+4}: two IEEE operations per instruction, same bits as the scalar ops
                const f2 fa = f2{(float)hxx[j], (float)hxx[j + 4]} * a.s2, fb = f2{(float)hxy[j], (float)hxy[j + 4]} * a.s2, fc = f2{(float)hyy[j], (float)hyy[j + 4]} * a.s2;
                const f2 t1 = fa * fc, t2 = fb * fb, t3 = fa + fc;
                const f2 t4 = a.k * t3;
                const f2 t5 = t4 * t3;
                const f2 rr = (t1 - t2) - t5;
                r[j] = rr.x;
                r[j + 4] = rr.y;
            }
            if (WANT_RESP && live && y >= ys && y < ye) {
                uint8_t* o = rf + (size_t)y * a.rstep + 4 * (size_t)x;
                if constexpr (RAG) {   // response rows that are only 4-byte aligned; the row's last, partial run
                    typedef float f4m __attribute__((ext_vector_type(4), aligned(4)));
                    if (nvalid == 8) {
                        *(f4m*)o = f4m{r[0], r[1], r[2], r[3]};
                        *(f4m*)(o + 16) = f4m{r[4], r[5], r[6], r[7]};
                    } else {
                        for (int j = 0; j < nvalid; ++j) ((float*)o)[j] = r[j];
                    }
                } else {
                    __builtin_nontemporal_store(f4v{r[0], r[1], r[2], r[3]}, (f4v*)o);
                    __builtin_nontemporal_store(f4v{r[4], r[5], r[6], r[7]}, (f4v*)(o + 16));
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

template <int B>
void launch_resp(const HBArgs& a, dim3 grid, bool rag, hipStream_t st)
{
    const int mode = a.mask ? (a.resp ? 2 : 1) : 0;
    if (rag) {
        if (mode == 0) RCV_LAUNCH((k_harris_resp_rows<B, true, 0>), grid, dim3(256), 0, st, a);
        else if (mode == 1) RCV_LAUNCH((k_harris_resp_rows<B, true, 1>), grid, dim3(256), 0, st, a);
        else RCV_LAUNCH((k_harris_resp_rows<B, true, 2>), grid, dim3(256), 0, st, a);
    } else {
        if (mode == 0) RCV_LAUNCH((k_harris_resp_rows<B, false, 0>), grid, dim3(256), 0, st, a);
        else if (mode == 1) RCV_LAUNCH((k_harris_resp_rows<B, false, 1>), grid, dim3(256), 0, st, a);
        else RCV_LAUNCH((k_harris_resp_rows<B, false, 2>), grid, dim3(256), 0, st, a);
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
    a.blocks_per_xcd = rcv_knobs().xcd_order == 0 ? 0 : (int)((nblocks + 7) / 8);
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
