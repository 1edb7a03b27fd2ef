// rcv_sobel_rows.hip -- Sobel 3x3 (u8 gray -> i16 dx, dy) as a register sliding window.  HBM-bound: 1 B read and
// 4 B written per pixel (5 algorithmic B/px).
//
// One WAVE owns a strip of 512 px (64 lanes x 8 px: 512 B of gray, 1024 B of each i16 output per row -- whole 128-B
// lines, so no two waves ever write parts of the same line; strips with seams inside a line measured ~35 % lower
// store rates, see DESIGN_HISTORY.md 4.1) and walks down a row segment.  Per source row a lane loads its 8 pixels as one
// 8-byte vector, gets the pixel left/right of its run from the neighbouring lanes with one DPP wave shift each (the
// wave's outermost two pixels come from one extra byte load per lane 0 / lane 63), and forms the horizontal parts
// with packed 16-bit math:
//     h1(x) = p[x+1] - p[x-1]             h2(x) = p[x-1] + 2 p[x] + p[x+1]
//     dx(y) = h1(y-1) + 2 h1(y) + h1(y+1)   dy(y) = h2(y+1) - h2(y-1)
// The two previous rows' h1/h2 stay in registers, so every source row is read exactly once per strip and no LDS
// is used.  Rows are prefetched a group of 8 ahead into registers; all loads and stores are unconditional
// (clamped addresses / dump line) so the compiler can keep counted vmcnt waits.  BORDER_REFLECT_101: rows by
// index reflection, the single reflected column at each image edge by a byte move inside the edge lane.
// RAG instantiation: ANY width >= 8 and ANY alignment (the reference's Mat::new gives step = cols * channels,
// rustcv/src/core/mat.rs:18-29, so odd widths mean byte-aligned rows): the same window on unaligned 8-byte loads and
// 16-byte stores (one wave instruction then costs ~45 instead of ~16 cycles of the address path -- still far from the
// generic per-sample kernels these shapes used to fall to), the lane that holds the last, partial run of a row rebuilds
// its 8 logical pixels -- the valid ones and their mirror images -- from the clamped load with one byte permute per
// dword, and stores only its valid samples.
#include "rcv_internal.h"
#include "rcv_kernels.h"
#include "rcv_device_utils.h"
#include <type_traits>


namespace {

typedef short s2v __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

constexpr int kRowsAhead = 8;
constexpr int kStripPx = 64 * 8;

struct SobelArgs {
    const uint8_t* src;
    uint8_t *dx, *dy, *dump;
    size_t sstep, xstep, ystep, sfs, xfs, yfs;
    int rows, cols, nstrips, seg_rows, nsegs, total_waves;
    int blocks_per_xcd;   // > 0: XCD-contiguous block order (see the kernel)
};

__device__ __forceinline__ uint32_t pk(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)
{
    s2v r = __builtin_bit_cast(s2v, a) - __builtin_bit_cast(s2v, b);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)
{
    s2v r = __builtin_bit_cast(s2v, a) + __builtin_bit_cast(s2v, b);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_add2x(uint32_t a, uint32_t b)  // a + 2*b
{
    s2v r = __builtin_bit_cast(s2v, a) + __builtin_bit_cast(s2v, b) * (short)2;
    return __builtin_bit_cast(uint32_t, r);
}

struct U2 { uint32_t lo, hi; };
struct __attribute__((packed, aligned(1))) U2u { uint32_t lo, hi; };          // unaligned views (RAG)
struct __attribute__((packed, aligned(1))) U1u { uint32_t v; };
struct __attribute__((packed, aligned(2))) U4u { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(2))) U2h { uint32_t a, b; };
struct __attribute__((packed, aligned(2))) U1h { uint32_t a; };

// DBG (instantiated by hand when profiling; the product launches DBG = 0 only): 1 skip global stores, 2 skip global loads
// BGR = true: the source is a BGR image and the gradient is taken of its gray conversion (the fixed-point BT.601 of
// RCV_BGR2GRAY, two v_dot4 per pixel) -- cvtColor + Sobel in one launch, 7 instead of 9 bytes per pixel and no gray image.
template <int DBG, bool BGR, bool RAG = false>
__global__ __launch_bounds__(256) void k_sobel_rows(SobelArgs a)
{
    const int lane = threadIdx.x & 63;
    // Block order (speed only): hardware places block b on XCD b % 8.  Each XCD works through its own contiguous eighth of the
    // (frame, segment, strip) list, so that what ONE XCD has in flight is a compact address range (measured on the filter
    // kernel: -9 % against dealing neighbouring work round-robin to the XCDs, DESIGN_HISTORY.md 6)
    const int blk = a.blocks_per_xcd > 0 ? (int)(blockIdx.x & 7) * a.blocks_per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    int wid = __builtin_amdgcn_readfirstlane(blk * 4 + (int)(threadIdx.x >> 6));   // scalar: row indices and row bases on the SALU
    if (wid >= a.total_waves) return;
    const int strip = wid % a.nstrips;
    wid /= a.nstrips;
    const int seg = wid % a.nsegs;
    const int frame = wid / a.nsegs;
    // segments of equal height up to one row (no short last segment: the launch ends when its slowest wave does)
    const int ys = (int)((long long)a.rows * seg / a.nsegs), ye = (int)((long long)a.rows * (seg + 1) / a.nsegs);
    const int x = strip * kStripPx + 8 * lane;                 // first pixel of this lane's run (may be >= cols in the last strip)
    const int xc = min(x, a.cols - 8);                         // clamped load position
    const bool edgeR = x == a.cols;                            // lane right of the image: supplies the mirrored column cols-2
    const bool live = x < a.cols;
    // RAG: the lane's 8 logical pixels x .. x+7 (beyond the image: their mirror images) as byte selectors into the 8 bytes it
    // loads from the clamped position xc; identity for every lane whose run lies inside the image
    uint32_t sel_lo = 0x03020100u, sel_hi = 0x07060504u;
    int nvalid = 8;
    if (RAG) {
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
    // the pixel outside the wave: lane 0 needs x-1 (mirror: 1), lane 63 needs x+8 (mirror: cols-2); other lanes load
    // a harmless in-row byte so that the load stays unconditional
    const int xe = lane == 0 ? (x == 0 ? 1 : x - 1) : min(x + 8 >= a.cols ? a.cols - 2 : x + 8, a.cols - 1);
    constexpr int PX = BGR ? 3 : 1;   // source bytes per pixel
    // the one pixel outside the wave's 512: only lanes 0 and 63 use it, so every other lane reads one fixed cached address
    const bool needs_edge = lane == 0 || lane == 63;
    const unsigned eoff = needs_edge ? (unsigned)(PX * max(xe, 0)) : 0u;
    const uint8_t* se = a.src + (size_t)frame * a.sfs;
    const uint8_t* sf = a.src + (size_t)frame * a.sfs + PX * xc;
    uint8_t* dxp = a.dx + (size_t)frame * a.xfs + 2 * (size_t)max(x, 0);
    uint8_t* dyp = a.dy + (size_t)frame * a.yfs + 2 * (size_t)max(x, 0);
    uint8_t* const dump = a.dump + lane * 16;

    auto gray_of = [](uint32_t px) -> uint32_t {   // (B,G,R,x) dword -> gray, bit-identical to k_bgr2gray
        const uint32_t hi8 = __builtin_amdgcn_udot4(px, 0x00132507u, 0u, false);       // 7*B + 37*G + 19*R
        const uint32_t lo8 = __builtin_amdgcn_udot4(px, 0x0023914cu, 8192u, false);    // 76*B + 145*G + 35*R + 8192
        return ((hi8 << 8) + lo8) >> 14;
    };
    struct Row { U2 v; uint32_t e; };
    struct Raw { uint32_t d[BGR ? 6 : 2]; uint32_t e0, e1; };
    auto load_raw = [&](int ry) -> Raw {   // ry in [ys-1, ...]: reflect, and clamp past the segment to a valid row
        ry = min(ry, ye);
        const int r = ry < 0 ? -ry : (ry >= a.rows ? 2 * a.rows - 2 - ry : ry);
        Raw w;
        if (DBG & 2) {
#pragma unroll
            for (int i = 0; i < (BGR ? 6 : 2); ++i) w.d[i] = (uint32_t)(r + i + lane);
            w.e0 = w.e1 = 0;
            return w;
        }
        const uint8_t* row = sf + (size_t)r * a.sstep;
        if constexpr (RAG) {
            const U2u q0 = *(const U2u*)row;
            w.d[0] = q0.lo; w.d[1] = q0.hi;
            if constexpr (BGR) {
                const U2u q1 = *(const U2u*)(row + 8), q2 = *(const U2u*)(row + 16);
                w.d[2] = q1.lo; w.d[3] = q1.hi; w.d[4] = q2.lo; w.d[5] = q2.hi;
                // edge pixel: the (unaligned) dword at its first byte, clamped so that it ends inside the row
                const unsigned ec = min(eoff, (unsigned)(3 * a.cols - 4));
                w.e0 = (*(const U1u*)(se + (size_t)r * a.sstep + ec)).v >> (8 * (eoff - ec));
                w.e1 = 0;
            } else {
                w.e0 = se[(size_t)r * a.sstep + eoff];
                w.e1 = 0;
            }
            return w;
        }
        const U2 q0 = *(const U2*)row;
        w.d[0] = q0.lo; w.d[1] = q0.hi;
        if constexpr (BGR) {
            const U2 q1 = *(const U2*)(row + 8), q2 = *(const U2*)(row + 16);
            w.d[2] = q1.lo; w.d[3] = q1.hi; w.d[4] = q2.lo; w.d[5] = q2.hi;
            // edge pixel = 3 bytes at byte eoff: the two aligned dwords that contain them (rows are 8-byte aligned)
            const uint8_t* er = se + (size_t)r * a.sstep;
            const unsigned a4 = eoff & ~3u, lim = (unsigned)(3 * a.cols - 4) & ~3u;
            w.e0 = *(const uint32_t*)(er + a4);
            w.e1 = *(const uint32_t*)(er + min(a4 + 4, lim));
        } else {
            w.e0 = se[(size_t)r * a.sstep + eoff];
            w.e1 = 0;
        }
        return w;
    };
    auto to_row = [&](const Raw& w) -> Row {   // BGR: 8 pixels + the edge pixel to gray
        if constexpr (!BGR) return Row{U2{w.d[0], w.d[1]}, w.e0};
        else {
            uint32_t g[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k0 = 3 * j, w0 = k0 >> 2, sh = k0 & 3;
                g[j] = gray_of(sh == 0 ? w.d[w0] : __builtin_amdgcn_alignbyte(w.d[w0 + 1 < 6 ? w0 + 1 : 5], w.d[w0], sh));
            }
            const uint32_t ge = gray_of(RAG ? w.e0 : __builtin_amdgcn_alignbyte(w.e1, w.e0, eoff & 3u));
            return Row{U2{g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24), g[4] | (g[5] << 8) | (g[6] << 16) | (g[7] << 24)}, ge};
        }
    };

    uint32_t h1a[4], h1b[4], h2a[4], h2b[4];  // rows r-2 (a) and r-1 (b)
#pragma unroll
    for (int j = 0; j < 4; ++j) h1a[j] = h1b[j] = h2a[j] = h2b[j] = 0;

    // r = index of the row just loaded; emits output row r-1.  STORE = false: the segment's first two rows (no output row is complete yet)
    auto feed = [&](const Row& rw, int r, auto store_tag) {
        constexpr bool STORE = decltype(store_tag)::value;
        U2 v = rw.v;
        // x = cols mirrors cols-2: the lane just right of the image holds cols-8..cols-1 after clamping -> its byte 0 := byte 6
        if (RAG) v = U2{pk(v.hi, v.lo, sel_lo), pk(v.hi, v.lo, sel_hi)};
        else if (edgeR) v.lo = pk(v.hi, v.lo, 0x03020106u);
        uint32_t lf = __builtin_amdgcn_update_dpp(0u, v.hi, 0x138, 0xf, 0xf, true);  // wave_shr:1 -> lane-1's hi dword
        uint32_t rt = __builtin_amdgcn_update_dpp(0u, v.lo, 0x130, 0xf, 0xf, true);  // wave_shl:1 -> lane+1's lo dword
        if (lane == 0) lf = rw.e << 24;   // byte 3 of "lane -1's hi dword"
        if (lane == 63) rt = rw.e;        // byte 0 of "lane 64's lo dword"
        // 16-bit pairs: L_j = (p[2j-1], p[2j]), C_j = (p[2j], p[2j+1]), R_j = (p[2j+1], p[2j+2]) = L_{j+1}
        uint32_t L[5], Cc[4];
        L[0] = pk(lf, v.lo, 0x0c000c07u);
        L[1] = pk(v.lo, v.lo, 0x0c020c01u);
        L[2] = pk(v.hi, v.lo, 0x0c040c03u);
        L[3] = pk(v.hi, v.hi, 0x0c020c01u);
        L[4] = pk(rt, v.hi, 0x0c040c03u);
        Cc[0] = pk(v.lo, v.lo, 0x0c010c00u);
        Cc[1] = pk(v.lo, v.lo, 0x0c030c02u);
        Cc[2] = pk(v.hi, v.hi, 0x0c010c00u);
        Cc[3] = pk(v.hi, v.hi, 0x0c030c02u);
        uint32_t h1[4], h2[4], ox[4], oy[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h1[j] = pk_sub(L[j + 1], L[j]);
            h2[j] = pk_add2x(pk_add(L[j], L[j + 1]), Cc[j]);
            if constexpr (STORE) {
                ox[j] = pk_add2x(pk_add(h1a[j], h1[j]), h1b[j]);
                oy[j] = pk_sub(h2[j], h2a[j]);
            }
            h1a[j] = h1b[j];
            h1b[j] = h1[j];
            h2a[j] = h2b[j];
            h2b[j] = h2[j];
        }
        if constexpr (!STORE) return;
        const int y = r - 1;
        // (the caller only feeds rows whose output row lies in [ys, ye): no store of a whole wave ever goes to the dump line --
        //  non-temporal stores of many waves to one shared line serialise, which used to cost the rows a segment's last prefetch
        //  group held beyond its end: 64 x 4K 0.53 -> 0.45 ms with segments of 8k - 2 rows, round 3)
        const bool st = live;
        if (DBG & 1) {
            if (ox[0] == 0x12345678u && oy[1] == 0x9abcdef0u) *(uint4*)dump = make_uint4(ox[0], ox[1], oy[2], oy[3]);
            return;
        }
        if constexpr (RAG) {
            uint8_t* px = dxp + (size_t)y * a.xstep;
            uint8_t* py = dyp + (size_t)y * a.ystep;
            if (st && nvalid == 8) {
                *(U4u*)px = U4u{ox[0], ox[1], ox[2], ox[3]};
                *(U4u*)py = U4u{oy[0], oy[1], oy[2], oy[3]};
            } else if (st) {   // the row's last, partial run: 4 + 2 + 1 samples as its length says
                int j = 0;
                if (nvalid & 4) {
                    *(U2h*)px = U2h{ox[0], ox[1]};
                    *(U2h*)py = U2h{oy[0], oy[1]};
                    j = 2;
                }
                const uint32_t bx = j ? ox[2] : ox[0], by = j ? oy[2] : oy[0], cx = j ? ox[3] : ox[1], cy = j ? oy[3] : oy[1];
                if (nvalid & 2) {
                    *(U1h*)(px + 4 * j) = U1h{bx};
                    *(U1h*)(py + 4 * j) = U1h{by};
                }
                if (nvalid & 1) {
                    const uint32_t lx = (nvalid & 2) ? cx : bx, ly = (nvalid & 2) ? cy : by;
                    const int o = 4 * j + ((nvalid & 2) ? 4 : 0);
                    *(uint16_t*)(px + o) = (uint16_t)lx;
                    *(uint16_t*)(py + o) = (uint16_t)ly;
                }
            }
            return;
        }
        if (DBG & 4) {   // ablation: plain stores
            *(uint4*)(st ? dxp + (size_t)y * a.xstep : dump) = make_uint4(ox[0], ox[1], ox[2], ox[3]);
            *(uint4*)(st ? dyp + (size_t)y * a.ystep : dump) = make_uint4(oy[0], oy[1], oy[2], oy[3]);
            return;
        }
        // non-temporal: the gradients are never read back by this launch
        __builtin_nontemporal_store(v4u{ox[0], ox[1], ox[2], ox[3]}, (v4u*)(st ? dxp + (size_t)y * a.xstep : dump));
        __builtin_nontemporal_store(v4u{oy[0], oy[1], oy[2], oy[3]}, (v4u*)(st ? dyp + (size_t)y * a.ystep : dump));
    };

    // rows ys-1 .. ye are consumed (ye - ys + 2 rows); groups of kAheadRows, the next group in flight while this one computes.  The
    // first two rows complete no output row (static no-store variant); full groups run unconditionally (counted vmcnt waits);
    // the last, partial group checks row by row -- nothing is in flight behind it that a conservative wait could hurt.
    const int nrows = ye - ys + 2;
    constexpr int kAheadRows = BGR ? 4 : kRowsAhead;   // (a BGR row is 8 registers per lane)
    const std::true_type yes{};
    const std::false_type no{};
    Raw cur[kAheadRows], nxt[kAheadRows];
#pragma unroll
    for (int i = 0; i < kAheadRows; ++i) cur[i] = load_raw(ys - 1 + i);
    int g = 0;
    if (nrows >= kAheadRows) {   // (a segment has at least kAheadRows - 2 rows: host plan; tiny images take the tail below)
#pragma unroll
        for (int i = 0; i < kAheadRows; ++i) nxt[i] = load_raw(ys - 1 + kAheadRows + i);
        feed(to_row(cur[0]), ys - 1, no);
        feed(to_row(cur[1]), ys, no);
#pragma unroll
        for (int i = 2; i < kAheadRows; ++i) feed(to_row(cur[i]), ys - 1 + i, yes);
#pragma unroll
        for (int i = 0; i < kAheadRows; ++i) cur[i] = nxt[i];
        for (g = kAheadRows; g + kAheadRows <= nrows; g += kAheadRows) {
#pragma unroll
            for (int i = 0; i < kAheadRows; ++i) nxt[i] = load_raw(ys - 1 + g + kAheadRows + i);
#pragma unroll
            for (int i = 0; i < kAheadRows; ++i) feed(to_row(cur[i]), ys - 1 + g + i, yes);
#pragma unroll
            for (int i = 0; i < kAheadRows; ++i) cur[i] = nxt[i];
        }
    }
#pragma unroll
    for (int i = 0; i < kAheadRows; ++i) {
        const int r = ys - 1 + g + i;
        if (g + i >= nrows) break;
        if (r - 1 >= ys) feed(to_row(cur[i]), r, yes);
        else feed(to_row(cur[i]), r, no);
    }
}

} // namespace

int rcv_sobel_tiled(rcv_ctx* ctx, const View& s, const View& dx, const View& dy)
{
    if (s.ch != 1 && s.ch != 3) return RCV_ERR_UNSUPPORTED;
    if (s.cols < 8 || s.rows < 2) return RCV_ERR_UNSUPPORTED;
    // widths that are not a multiple of 8 and rows that are not 8 / 16-byte aligned: the RAG instantiation (i16 outputs are
    // at least 2-byte aligned by construction of rcv_mat)
    const bool rag = s.cols % 8 != 0 || (uintptr_t)s.p % 8 || s.step % 8 || (s.n > 1 && s.fstride % 8) || (uintptr_t)dx.p % 16 || dx.step % 16 ||
                     (dx.n > 1 && dx.fstride % 16) || (uintptr_t)dy.p % 16 || dy.step % 16 || (dy.n > 1 && dy.fstride % 16);
    if (rag && ((uintptr_t)dx.p % 2 || dx.step % 2 || dx.fstride % 2 || (uintptr_t)dy.p % 2 || dy.step % 2 || dy.fstride % 2)) return RCV_ERR_UNSUPPORTED;
    SobelArgs a;
    a.src = s.p;
    a.dx = dx.p;
    a.dy = dy.p;
    a.dump = ctx->kconst + RCV_KC_SOBEL_DUMP;
    a.sstep = s.step;
    a.xstep = dx.step;
    a.ystep = dy.step;
    a.sfs = s.fstride;
    a.xfs = dx.fstride;
    a.yfs = dy.fstride;
    a.rows = s.rows;
    a.cols = s.cols;
    a.nstrips = (s.cols + kStripPx - 1) / kStripPx;
    // Segments of ~24 rows, equal up to one row (round 3, tools/ablate_sobel.py on 64 4K frames: 12 ... 40 rows 0.417-0.420 ms, 68
    // rows 0.439, 90 rows 0.445 -- with the XCD-contiguous block order shorter segments keep what one XCD has in flight more
    // compact; a BGR source likes them shorter still: 12-20 rows 0.614-0.617 against 0.648 at 68).
    int seg = s.ch == 3 ? 16 : 24;
    a.seg_rows = seg;
    a.nsegs = s.rows >= 2 * seg ? (s.rows + seg / 2) / seg : 1;
    long long waves = (long long)a.nstrips * a.nsegs * s.n;
    if (waves > 0x3fffffff) return RCV_ERR_UNSUPPORTED;
    a.total_waves = (int)waves;
    const long long nblocks = (waves + 3) / 4;
    a.blocks_per_xcd = (int)((nblocks + 7) / 8);
    const dim3 grid((unsigned)(a.blocks_per_xcd > 0 ? a.blocks_per_xcd * 8 : nblocks));
    // occupancy cap (workgroups per CU) through an untouched dynamic-LDS request; 0 = what the registers allow
    // measured on 64 4K frames: 5 (registers) / 4 workgroups per CU 0.519 ms, 3 workgroups 0.474 ms (gray); 0.638 / 0.625 (BGR source)
    const int wgs = 3;
    const unsigned lds = wgs < 5 ? (unsigned)((163840 / wgs) & ~511) : 0u;
    if (rag) {
        if (s.ch == 3) RCV_LAUNCH((k_sobel_rows<0, true, true>), grid, dim3(256), lds, ctx->stream, a);
        else RCV_LAUNCH((k_sobel_rows<0, false, true>), grid, dim3(256), lds, ctx->stream, a);
        return rcv_launch_check(ctx);
    }
    if (s.ch == 3) {
        RCV_LAUNCH((k_sobel_rows<0, true>), grid, dim3(256), lds, ctx->stream, a);
        return rcv_launch_check(ctx);
    }
    RCV_LAUNCH((k_sobel_rows<0, false>), grid, dim3(256), lds, ctx->stream, a);
    return rcv_launch_check(ctx);
}
