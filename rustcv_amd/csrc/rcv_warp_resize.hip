// rcv_warp_resize.hip -- fused warpAffine -> exact SxS down-scale (S in {2, 4}), BGR: SURVEY.md 8(f) row f1, BASELINE config 4
// (8K warpAffine + resize -> 1080p).  Not in the reference; semantics = resize(warp_affine(.)) of SURVEY.md 8-A ==
// oracle/rcv_oracle.c orc_resize(orc_warp_affine(.)), bit for bit.
#include "rcv_geom_dev.h"
#include <limits.h>

namespace {

constexpr int kBlock = 256;

// ---- fused warpAffine -> exact SxS down-scale (S in {2, 4}), BGR ("next" row f1) ---------------------
// resize(warp_affine(src -> mid), dst) with mid = S * dst.  For an exact integer factor the bilinear resize reads only
// the centre 2x2 of every SxS block of `mid` (k_resize_box above), so the fused kernel evaluates just those four warped
// pixels per output pixel -- bit for bit what the warp kernels produce (same f32 ops, same order, same rounding) -- and
// averages them with (a+b+c+d+2)>>2.  `mid` never exists: 1/4 (S=2: all) of the warp arithmetic of the unfused pair and
// none of its 2 x 3 B/px intermediate traffic.  One thread per output pixel, four lanes share a 12-byte store.
#ifndef RCV_BOX_TW
#define RCV_BOX_TW 64
#endif
constexpr int kBoxTileW = RCV_BOX_TW, kBoxTileH = 256 / RCV_BOX_TW;   // output tile of one workgroup (4 waves of 16 x 4)

// Tile order of the fused launches.  Hardware places block b on XCD b % 8.
//  per_xcd == 0: the 3-D grid as launched (x fastest: raster order, neighbouring tiles on different XCDs);
//  per_xcd > 0:  wl_tile's order -- the whole tile list (frame group, strip, tile row, tile column) cut into eight contiguous runs, one
//                per XCD: neighbouring tiles share an L2, but the eight XCDs stream eight distant regions (other frame groups) at once;
//  per_xcd < 0:  SYNCHRONOUS STRIPES (round 5): every frame group's tile list (strip, tile row, tile column) is cut into eight runs of
//                -per_xcd tiles, XCD k takes run k of group 0, then run k of group 1, ...: with vertical strips each XCD walks its own
//                stripe of the image from top to bottom while all eight work on the same rows of the same frames -- a tile's lines
//                shared with the tile below (a rotated tile's row pieces end in lines that continue in its vertical neighbours:
//                x1.3 of the source at 7 degrees) are L2 hits a few workgroups later, and DRAM sees one band of one frame group.
__device__ __forceinline__ bool wr_tile(int per_xcd, int strip, int gx, int gy, int ngroups, int& bx, int& by, int& bz)
{
    if (per_xcd >= (1 << 30)) {
        // BLOCKS (round 5): the tile grid cut into blocks of bw x bh tiles, block j of a frame group on XCD j % 8 (its bw * bh tiles
        // consecutive in that XCD's dispatch order): neighbouring tiles of a block share an L2 -- the lines at the ends of a rotated
        // tile's row pieces continue in the tile above / below / beside it -- while the eight XCDs stay in one neighbourhood of the image
        const int bw = per_xcd & 255, bh = (per_xcd >> 8) & 255, nbx = (gx + bw - 1) / bw, nby = (gy + bh - 1) / bh, per = bw * bh;
        const int q = (int)(blockIdx.x >> 3), blk8 = q / per, idx = q - blk8 * per;
        int blk = 8 * blk8 + (int)(blockIdx.x & 7);
        const int nb = nbx * nby;
        bz = blk / nb;
        blk -= bz * nb;
        const int byb = blk / nbx, bxb = blk - byb * nbx, iy = idx / bw;
        bx = bxb * bw + (idx - iy * bw);
        by = byb * bh + iy;
        return bz < ngroups && bx < gx && by < gy;
    }
    if (per_xcd >= 0) return wl_tile(per_xcd, strip, gx, gy, gx * gy * ngroups, bx, by, bz);
    const int pg = -per_xcd, q = (int)(blockIdx.x >> 3);
    bz = q / pg;
    const int rem = (int)(blockIdx.x & 7) * pg + (q - bz * pg);
    if (bz >= ngroups || rem >= gx * gy) return false;
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

// One wave's share of k_warp_resize_box: output pixel (x, y) of frame `frame` .
template <int S>
__device__ __forceinline__ void warp_resize_box_px(const View& s, const View& d, const Affine& A, const int frame, const int x, const int y)
{
    const uint8_t* sf = s.p + (size_t)frame * s.fstride;
    uint8_t* dfr = d.p + (size_t)frame * d.fstride;
    // A workgroup owns a 32 x 8 tile of output pixels, each wave a 16 x 4 sub-tile (lane = 16 * row + column): the taps of
    // a wave then fall into a compact source patch (64 x 16 px for S = 4, plus the rotation's drift) that its eight tap loads
    // reuse out of L1.  With one output ROW segment per wave (64 x 1) the same loads walked across 30 source rows at 7 degrees,
    // every line was used by two lanes only and fetched again by the waves of the neighbouring rows (1.66x the algorithmic
    // bytes from HBM; 64 % of the wave cycles waiting on memory).
    const int xq = min(x, d.cols - 1), yq = min(y, d.rows - 1);
    constexpr int o = S / 2 - 1;
    float sx[4], sy[4];
    bool inter = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float fxx = (float)(S * xq + o + (i & 1)), fyy = (float)(S * yq + o + (i >> 1));
        sx[i] = fmaf(A.m[0], fxx, fmaf(A.m[1], fyy, A.m[2]));
        sy[i] = fmaf(A.m[3], fxx, fmaf(A.m[4], fyy, A.m[5]));
        inter = inter && sx[i] >= 0.0f && sx[i] < (float)(s.cols - 3) && sy[i] >= 0.0f && sy[i] < (float)(s.rows - 1);
    }
    const bool small = ((uintptr_t)sf & 3) == 0 && (s.step & 3) == 0 && s.step < (1u << 24) && s.rows < (1 << 24) &&
                       (unsigned long long)s.rows * s.step < (1ull << 32);
    uint32_t p[4];
    if (small && __all(inter)) {   // wave-uniform: every tap of every lane inside the source (see k_warp_affine_bgr)
        struct U3 { uint32_t a, b, c; };
        U3 ta[4], tb[4];
        unsigned sh[4];
        float fx[4], fy[4];
        const unsigned sstep = (unsigned)s.step;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x0f = floorf(sx[i]), y0f = floorf(sy[i]);
            fx[i] = sx[i] - x0f;
            fy[i] = sy[i] - y0f;
            const unsigned off = __umul24((unsigned)(int)y0f, sstep) + 3u * (unsigned)(int)x0f;
            sh[i] = off & 3u;
            ta[i] = *(const U3*)(sf + (off & ~3u));
            tb[i] = *(const U3*)(sf + ((off & ~3u) + sstep));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t alo = __builtin_amdgcn_alignbyte(ta[i].b, ta[i].a, sh[i]), ahi = __builtin_amdgcn_alignbyte(ta[i].c, ta[i].b, sh[i]);
            const uint32_t blo = __builtin_amdgcn_alignbyte(tb[i].b, tb[i].a, sh[i]), bhi = __builtin_amdgcn_alignbyte(tb[i].c, tb[i].b, sh[i]);
            p[i] = bilerp_bgr<false>(alo, ahi, blo, bhi, f2{fx[i], fy[i]});
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            uint8_t o3[3];
            warp_px<3>(sf, s, A, (float)(S * xq + o + (i & 1)), (float)(S * yq + o + (i >> 1)), o3);
            p[i] = (uint32_t)o3[0] | ((uint32_t)o3[1] << 8) | ((uint32_t)o3[2] << 16);
        }
    }
    // (a+b+c+d+2)>>2 per channel: B and R ride in the two 16-bit halves of one dword, G in another
    const uint32_t br = (p[0] & 0x00ff00ffu) + (p[1] & 0x00ff00ffu) + (p[2] & 0x00ff00ffu) + (p[3] & 0x00ff00ffu) + 0x00020002u;
    const uint32_t gg = ((p[0] >> 8) & 0xffu) + ((p[1] >> 8) & 0xffu) + ((p[2] >> 8) & 0xffu) + ((p[3] >> 8) & 0xffu) + 2u;
    const uint32_t px = ((br >> 2) & 0x00ff00ffu) | ((gg >> 2) << 8);
    const uint32_t p1 = __builtin_amdgcn_update_dpp(0u, px, 0x101, 0xf, 0xf, true);
    const uint32_t p2 = __builtin_amdgcn_update_dpp(0u, px, 0x102, 0xf, 0xf, true);
    const uint32_t p3 = __builtin_amdgcn_update_dpp(0u, px, 0x103, 0xf, 0xf, true);
    if ((threadIdx.x & 3) == 0 && x < d.cols && y < d.rows) {
        struct U3 { uint32_t a, b, c; };
        *(U3*)(dfr + (size_t)y * d.step + (size_t)x * 3) =
            U3{__builtin_amdgcn_perm(p1, px, 0x04020100u), __builtin_amdgcn_perm(p2, p1, 0x05040201u), __builtin_amdgcn_perm(p3, p2, 0x06050402u)};
    }
}

#ifndef RCV_BOX_WW
#define RCV_BOX_WW 32
#endif
template <int S>
__global__ __launch_bounds__(kBlock) void k_warp_resize_box(View s, View d, Affine A, int per_xcd, int strip, int gx, int gy)
{
    int bx, by, bz;
    if (!wr_tile(per_xcd, strip, gx, gy, d.n, bx, by, bz)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WW = RCV_BOX_WW, WH = 64 / WW;   // wave sub-tile
    const int x = bx * kBoxTileW + (wave % (kBoxTileW / WW)) * WW + (lane % WW);   // d.cols % 4 == 0: quads never straddle the row end
    const int y = by * kBoxTileH + (wave / (kBoxTileW / WW)) * WH + (lane / WW);
    warp_resize_box_px<S>(s, d, A, bz, x, y);
}

// ---- the same launch on EXACT ROW PIECES staged through LDS (round 5, second design) ------------------------------------------------
// What the counters of the two gather kernels say (profiles/r05_warp_resize_pmc_*): 367 M L1 cache accesses per launch = 0.89 per
// CU and cycle (eight gathers of ~41 distinct lines each per wave: the tag pipeline is full), 33 M L1 -> L2 line requests of which
// 93 % miss, 287 M VALU instructions of which ~150 per pixel repeat for every frame.  Halving the gathers alone does not help (four
// 16-byte row windows per lane instead of eight 12-byte taps, measured: 0.78 against 0.68 ms, gathers only -- every window is an L1
// miss).  This kernel changes all three at once:
//  * a workgroup keeps ONE 64 x 4 output tile and walks the frames of its frame group; per tile, once: the sample coordinates, the
//    interior test, fractions, and a PLAN of the source bytes the tile reads -- for every source row the exact byte range
//    [3 x0_min, 3 x0_max + 5] over the tile's samples that tap it (LDS atomics), cut to 16-byte chunks: ~750 chunks = 47 B per output
//    pixel at 7 degrees (the bounding box of the rotated tile is 1.9x that);
//  * per frame the chunk list is fetched by three `buffer_load_dwordx4 ... lds` per lane (global -> LDS directly: chunk c lands at
//    byte 16 c of the buffer; the lane's source offset is the only register it needs, the frame base sits in the buffer resource) --
//    ~10 L1 accesses per wave instruction instead of 41, every line requested once per tile and frame;
//  * the taps come out of LDS: a row piece is contiguous there, so the tap pair (x0, y0) is the three dwords from tab[y0] + 3 x0 (hoisted;
//    lanes 12 bytes apart: no systematic bank conflict), read as ds_read2_b32 + ds_read_b32 and shifted into place as before;
//  * two buffers: frame f + 1 is in flight while frame f is computed; one barrier per frame.
// Per frame and pixel that leaves the tap shifts, the exact bilinear arithmetic (4 x 31), the box average and the store: ~160 VALU instructions.
// Tiles with a tap outside the source, with more than 64 source rows or more than kStageChunks chunks take k_warp_resize_box's path.
#ifndef RCV_STAGE_CHUNKS
#define RCV_STAGE_CHUNKS 832
#endif
#ifndef RCV_STAGE_WAIT
#define RCV_STAGE_WAIT 0x0f71   // vmcnt(1); 0x0f70: vmcnt(0)
#endif
// chunk slots (16 B) per buffer: three load instructions of 256 chunks + one of 64 (wave 0); 2 x 13 KB of LDS per workgroup
constexpr int kStageChunks = RCV_STAGE_CHUNKS, kStageLoads = (kStageChunks + 255) / 256, kStageBuf = kStageChunks * 16;

template <int S, int DBG = 0, int OCC = 5>   // OCC: waves per SIMD the register allocation aims at
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void k_warp_resize_stage(View s, View d, Affine A, int fpg, int gx, int gy, int ngroups, int per_xcd, int strip, int pad, int zfill)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t wrs_lds[];   // 2 x kStageBuf (+ whatever the host adds to cap the occupancy)
    int bx, by, bz;
    if (!wr_tile(per_xcd, strip, gx, gy, ngroups, bx, by, bz)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = bx * 64 + lane, y = by * 4 + wave;
    const int f0 = bz * fpg, f1 = min(f0 + fpg, d.n);
    const int xq = min(x & ~3, d.cols - 4) + (x & 3), yq = min(y, d.rows - 1);   // whole quads clamped into the image (d.cols % 4 == 0)
    constexpr int o = S / 2 - 1;
    float sx[4], sy[4];
    bool inter = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float fxx = (float)(S * xq + o + (i & 1)), fyy = (float)(S * yq + o + (i >> 1));
        sx[i] = fmaf(A.m[0], fxx, fmaf(A.m[1], fyy, A.m[2]));
        sy[i] = fmaf(A.m[3], fxx, fmaf(A.m[4], fyy, A.m[5]));
        inter = inter && sx[i] >= 0.0f && sx[i] < (float)(s.cols - 3) && sy[i] >= 0.0f && sy[i] < (float)(s.rows - 1);
    }
    // plan tables: alias the SECOND buffer (its first load is issued behind the barrier every table read has passed)
    int* const rowmin = (int*)(wrs_lds + kStageBuf);   // [64] first byte a row's taps need, then 16 * (first chunk in LDS - first chunk in the row)
    int* const rowmax = rowmin + 64;                   // [64] last byte; then the row's first chunk in the list
    int* const goff = rowmax + 64;                     // [64] the row's first chunk, counted from the row start (< 0: left of the image)
    int* const scal = goff + 64;                       // [0] lowest row, [1] highest row, [2] chunks in the list
    uint8_t* const map = (uint8_t*)(scal + 4);         // [kStageChunks] chunk -> row of the tile
    if (tid < 64) { rowmin[tid] = 0x7fffffff; rowmax[tid] = (int)0x80000000; }
    if (tid == 0) { scal[0] = 0x7fffffff; scal[1] = -0x7fffffff; scal[2] = 0; }
    // zfill (rows whose length is a multiple of 16 bytes: no chunk straddles a row end): tiles at the source border are staged too, on
    // a VIRTUAL source that is zero outside the image -- chunks outside the rows / outside a row are not fetched (buffer range: zeros
    // arrive in LDS), which is the constant border tap by tap; a sample the specification sets to 0 without reading taps (not
    // sx > -1 && sx < cols && sy > -1 && sy < rows; NaN coordinates included) gets zero weights on a pixel outside the image.
    bool staged = zfill != 0 || __syncthreads_and(inter) != 0;
    if (zfill != 0) __syncthreads();                   // (the barrier also publishes the table initialisation)
    int x0[4], y0[4];
    f2 fxy[4];
    if (staged) {
        int lo = 0x7fffffff, hi = -0x7fffffff;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x0f = floorf(sx[i]), y0f = floorf(sy[i]);
            fxy[i] = f2{sx[i] - x0f, sy[i] - y0f};
            x0[i] = (int)x0f;
            y0[i] = (int)y0f;
            const bool inx = sx[i] > -1.0f && sx[i] < (float)s.cols, iny = sy[i] > -1.0f && sy[i] < (float)s.rows;
            if (!(inx && iny)) {   // value 0: zero weights on a tap pair whose first pixel lies outside the image, next to the tile's own rows / columns
                if (!inx) x0[i] = sx[i] >= (float)s.cols ? s.cols : -2;   // (NaN: -2)
                if (!iny) y0[i] = sy[i] >= (float)s.rows ? s.rows : -2;
                fxy[i] = f2{0.0f, 0.0f};
            }
            lo = min(lo, y0[i]);
            hi = max(hi, y0[i] + 1);
        }
        // (an affine map is monotone along a wave's 64 x 1 pixels: the extremes of a wave sit in its first and last lane -- two lanes
        //  instead of 64 on one LDS address)
        //  (border tiles: samples pointed at row -2 break the order, so there a lane also reports when its value differs from its
        //  neighbour's)
        const int lol = (int)__builtin_amdgcn_update_dpp(0x7fffffffu, (unsigned)lo, 0x111, 0xf, 0xf, false);
        const int hil = (int)__builtin_amdgcn_update_dpp(0x80000001u, (unsigned)hi, 0x111, 0xf, 0xf, false);
        if (lane == 0 || lane == 63 || (zfill != 0 && (lol != lo || hil != hi))) {
            atomicMin(&scal[0], lo);
            atomicMax(&scal[1], hi);
        }
    }
    __syncthreads();
    const int r0 = scal[0];
    staged = staged && scal[1] - r0 < 64;
    if (staged) {
        // Byte range per source row.  Along a wave both x0 and y0 of a sample are monotone (affine map, one output row per wave), so
        // the lanes of a wave that tap one source row are a contiguous run and the run's extreme columns sit at its two ends: only
        // lanes whose row differs from a neighbour's (or that have no neighbour inside their 16-lane DPP row) touch the LDS -- at 0
        // degrees two lanes per wave and sample instead of 64 on one address (the plan of the first version cost 0.73 against 0.62 ms
        // for the gather kernel at 0 degrees; 8-way conflicts at 7 degrees)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = y0[i] - r0, b = 3 * x0[i];
            const int rl = (int)__builtin_amdgcn_update_dpp((unsigned)-1, (unsigned)r, 0x111, 0xf, 0xf, false);   // row_shr:1 (lane - 1; none: -1)
            const int rr = (int)__builtin_amdgcn_update_dpp((unsigned)-1, (unsigned)r, 0x101, 0xf, 0xf, false);   // row_shl:1 (lane + 1; none: -1)
            if (rl != r || rr != r) {
                atomicMin(&rowmin[r], b);
                atomicMax(&rowmax[r], b + 5);
                atomicMin(&rowmin[r + 1], b);
                atomicMax(&rowmax[r + 1], b + 5);
            }
        }
    }
    __syncthreads();
    if (staged && wave == 0) {   // one wave: chunk counts of the rows, their prefix sums, the chunk -> row map
        const int mn = rowmin[lane], mx = rowmax[lane];
        const int qs = mn >> 4, len = mn != 0x7fffffff ? (mx >> 4) - qs + 1 : 0;
        int inc = len;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            const int t = __shfl_up(inc, k);
            if (lane >= k) inc += t;
        }
        // One spare chunk slot behind every row piece when the buffer has room for it.  Along a wave the rows change every other lane at
        // nearly the same offset inside the piece, so with pieces of 16 chunks = 64 dwords the lanes l, l + 2, l + 4, ... meet in the same
        // few banks (SQ_LDS_BANK_CONFLICT: 72 % of the LDS cycles, the LDS 61 % busy); with the spare slot consecutive rows sit
        // 4 (+- 2) dwords further round the 64 banks.  pad = 0 (measurement): packed.
        const unsigned long long nonempty = __builtin_amdgcn_ballot_w64(len > 0);
        const int rows_used = __builtin_popcountll(nonempty), tot = __shfl(inc, 63);
        const int spare = (pad != 0 && tot + rows_used <= kStageChunks) ? 1 : 0;
        const int before = __builtin_popcountll(nonempty & ((1ull << lane) - 1ull));
        const int slots = len + (len > 0 ? spare : 0);
        const int cb = inc - len + spare * before;
        if (lane == 63) scal[2] = tot + spare * rows_used;
        if (tot + spare * rows_used <= kStageChunks)
            for (int i = 0; i < slots; ++i) map[cb + i] = (uint8_t)(i < len ? lane : 255);   // (255: a slot nothing is loaded into)
        rowmin[lane] = 16 * (cb - qs);
        rowmax[lane] = cb;
        goff[lane] = qs;   // (the row's first chunk, in chunks from the row start: may be negative)
    }
    __syncthreads();
    const int total = scal[2];
    staged = staged && total <= kStageChunks;
    if (!staged) {   // workgroup-uniform: a tap outside the source, or a footprint the buffers do not hold (steep or magnifying maps)
        if (x >= d.cols || y >= d.rows) return;
        for (int f = f0; f < f1; ++f) warp_resize_box_px<S>(s, d, A, f, x, y);
        return;
    }
    unsigned voff[kStageLoads];   // the lane's chunk of each load instruction: byte offset inside a frame (past the frame: no fetch)
#pragma unroll
    for (int j = 0; j < kStageLoads; ++j) {
        const int c = j * 256 + tid;
        voff[j] = 0xffffff00u;
        if (c < total) {
            const int r = map[c];
            if (r != 255) {
                const int row = r0 + r, q = goff[r] + (c - rowmax[r]);   // chunk q of source row `row`: fetched only if it lies inside the image
                if (row >= 0 && row < s.rows && q >= 0 && 16 * q < s.cols * 3) voff[j] = (unsigned)row * (unsigned)s.step + 16u * (unsigned)q;
            }
        }
    }
    const int nload = (total + 255) >> 8;   // load instructions that carry chunks (workgroup-uniform)
    unsigned la[4], lb[4], sh[4];   // LDS byte offset (buffer 0) of the dword below the upper / lower tap pair of each sample; the taps' byte in it
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = y0[i] - r0;
        sh[i] = (unsigned)(3 * x0[i]) & 3u;               // (row pieces start on 16-byte chunks: the same shift in both rows)
        la[i] = (unsigned)(rowmin[r] + 3 * x0[i]) & ~3u;
        lb[i] = (unsigned)(rowmin[r + 1] + 3 * x0[i]) & ~3u;
    }
    const int kq = min(lane & 3, 2);   // (the store scheme of k_warp_resize_loop: every lane one dword of its quad's 12 bytes)
    const unsigned doff = (unsigned)yq * (unsigned)d.step + (unsigned)(xq & ~3) * 3u + 4u * (unsigned)kq;
    const uint32_t psel = kq == 0 ? 0x04020100u : (kq == 1 ? 0x05040201u : 0x06050402u);
    constexpr int kRsrc = 0x00020000;
    // (the buffer range check is per dword -- tools/probe_buffer_range.hip: a chunk that runs past the end of the frame still delivers the
    //  dwords below it, and every tap byte lies in one)
    const unsigned sbytes = (unsigned)min((unsigned long long)s.cap, (unsigned long long)s.rows * s.step);
    const uint8_t* sf = s.p + (size_t)f0 * s.fstride;
    uint8_t* df = d.p + (size_t)f0 * d.fstride;
    auto issue = [&](const uint8_t* base, const int buf) {   // frame -> buffer `buf`: three direct-to-LDS loads per lane
        if constexpr (DBG == 2) return;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, sbytes, kRsrc);
#pragma unroll
        for (int j = 0; j < kStageLoads; ++j)
            if ((j < 2 || nload > j) && j * 256 + wave * 64 < kStageChunks)   // (the last instruction: only the waves whose 64 slots exist)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(wrs_lds + buf * kStageBuf + (j * 256 + wave * 64) * 16), 16,
                                                     voff[j], 0, 0, 0);
    };
    auto finish = [&](uint8_t* dbase, const int buf) {
        const __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc((void*)dbase, 0, 0xffffffff, kRsrc);
        if constexpr (DBG == 1) {   // staging and stores only
            __builtin_amdgcn_raw_buffer_store_b32(voff[0] + buf, w, doff, 0, 0);
            return;
        }
        // three dwords from the dword below the tap: ds_read2_b32 + ds_read_b32.  (A DS read wider than a dword that is not naturally
        // aligned -- b64 off 8, b96 / b128 off 16 -- is served one lane per cycle on gfx950: 62 cycles per wave instruction against 7-13
        // for this pair, tools/ubench_lds_taps.hip.)
        uint32_t p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t* qa = (const uint32_t*)(wrs_lds + (la[i] + buf * kStageBuf));
            const uint32_t* qb = (const uint32_t*)(wrs_lds + (lb[i] + buf * kStageBuf));
            const uint32_t a0 = qa[0], a1 = qa[1], a2 = qa[2], b0 = qb[0], b1 = qb[1], b2 = qb[2];
            p[i] = bilerp_bgr<true>(__builtin_amdgcn_alignbyte(a1, a0, sh[i]), __builtin_amdgcn_alignbyte(a2, a1, sh[i]),
                                    __builtin_amdgcn_alignbyte(b1, b0, sh[i]), __builtin_amdgcn_alignbyte(b2, b1, sh[i]), fxy[i]);
        }
        const uint32_t br = (p[0] & 0x00ff00ffu) + (p[1] & 0x00ff00ffu) + (p[2] & 0x00ff00ffu) + (p[3] & 0x00ff00ffu) + 0x00020002u;
        const uint32_t gg = ((p[0] >> 8) & 0xffu) + ((p[1] >> 8) & 0xffu) + ((p[2] >> 8) & 0xffu) + ((p[3] >> 8) & 0xffu) + 2u;
        const uint32_t px = ((br >> 2) & 0x00ff00ffu) | ((gg >> 2) << 8);
        const uint32_t pa = __builtin_amdgcn_update_dpp(0u, px, 0xA4, 0xf, 0xf, true);   // quad_perm [0,1,2,2]
        const uint32_t pb = __builtin_amdgcn_update_dpp(0u, px, 0xF9, 0xf, 0xf, true);   // quad_perm [1,2,3,3]
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(pb, pa, psel), w, doff, 0, 0);
    };
    const size_t sfs = s.fstride, dfs = d.fstride;
    issue(sf, 0);
    // gfx9 counts loads and stores in one in-order counter: with the frame's store as the YOUNGEST operation at the loop head, vmcnt(1)
    // waits for the staged loads and leaves the store in flight.  A first store (the lane's own output dword of frame f0, rewritten by
    // the frame itself) gives the entry edge the same shape as the back edge.
    __builtin_amdgcn_raw_buffer_store_b32(0u, __builtin_amdgcn_make_buffer_rsrc((void*)df, 0, 0xffffffff, kRsrc), doff, 0, 0);
    int f = f0;
#pragma unroll 1
    for (;;) {
        // frame f is in buffer 0 once every wave's loads have landed; all waves have left frame f - 1 (buffer 1 is free)
        // (scheduling barriers: the wait's count assumes that a half's loads are all OLDER than its store -- nothing may move a load
        //  below the store or the store above a load; tests/test_isa_waits.py checks the order in the ISA as well)
        __builtin_amdgcn_s_waitcnt(RCV_STAGE_WAIT);   // vmcnt(1) (expcnt / lgkmcnt untouched)
        __syncthreads();
        if (f + 1 < f1) issue(sf + sfs, 1);
        __builtin_amdgcn_sched_barrier(0);
        finish(df, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (++f >= f1) break;
        __builtin_amdgcn_s_waitcnt(RCV_STAGE_WAIT);
        __syncthreads();
        if (f + 1 < f1) issue(sf + 2 * sfs, 0);
        __builtin_amdgcn_sched_barrier(0);
        finish(df + dfs, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (++f >= f1) break;
        sf += 2 * sfs;
        df += 2 * dfs;
    }
}

// (Round 3 built the fused warp -> 4x down-scale on an LDS-staged source patch as well -- k_warp_resize_lds: 16 x 16 output tiles, the warp
//  kernel's staging plan and double buffer; bit-exact, 0.91 ms against this gather kernel's 0.715 ms on 32 x 8K -> 1080p because the samples
//  sit 4 pixels apart: <= 9 of 16 staged pixels are ever read and the tap reads are a 4-way bank conflict for every pitch.  Removed from the
//  product in round 4; the measurement is profiles/r03_warp_resize_lds.txt, the source is in the history at commit e87cdfd.)

bool wrl_ok(const View& s, const View& d)
{
    return ((uintptr_t)s.p & 3) == 0 && (s.step & 3) == 0 && (s.fstride & 3) == 0 && s.step < (1u << 24) && s.rows < (1 << 24) &&
           (unsigned long long)s.rows * s.step <= 0xffffff00ull && (unsigned long long)d.rows * d.step < (1ull << 32) && d.cols >= 4;
    // (<= 0xffffff00: the staged kernel's do-not-fetch offset must lie beyond the buffer descriptor's range, which is the frame's size)
}

// Does the staged kernel's plan hold for this map?  The plans of nine tiles spread over the output (a 3 x 3 grid at 1/4, 1/2, 3/4 of
// the tile grid), evaluated on the host with the kernel's own arithmetic: source rows a tile spans and 16-byte chunks of its row
// pieces.  Interior tiles of one map differ only by the fractional position of their corner (751 +- 11 chunks at 7 degrees), so the
// sample decides for the launch; tiles with a tap outside the source say nothing.  Maps that fail (steeper rotations, magnification)
// would send every workgroup down the staged kernel's fallback, which is much slower than the gather kernel itself (15 degrees:
// 1.83 against 1.06 ms), so they stay on the gather kernel.
bool wrs_fits(const View& s, const View& d, const Affine& A, int S)
{
    const int gx = (d.cols + 63) / 64, gy = (d.rows + 3) / 4, o = S / 2 - 1;
    int judged = 0;
    for (int t = 0; t < 9; ++t) {
        const int bx = gx * (1 + t % 3) / 4, by = gy * (1 + t / 3) / 4;
        int lo = INT_MAX, hi = INT_MIN, rmin[64], rmax[64];
        bool interior = true;
        for (int pass = 0; pass < 2 && interior; ++pass) {
            if (pass == 1) {
                if (hi - lo >= 64) return false;
                for (int r = 0; r <= hi - lo; ++r) { rmin[r] = INT_MAX; rmax[r] = -1; }
            }
            for (int ty = 0; ty < 4 && interior; ++ty)
                for (int tx = 0; tx < 64 && interior; ++tx) {
                    const int x = bx * 64 + tx, y = by * 4 + ty;
                    const int xq = min(x & ~3, d.cols - 4) + (x & 3), yq = min(y, d.rows - 1);
                    for (int i = 0; i < 4; ++i) {
                        const float fxx = (float)(S * xq + o + (i & 1)), fyy = (float)(S * yq + o + (i >> 1));
                        const float sx = fmaf(A.m[0], fxx, fmaf(A.m[1], fyy, A.m[2])), sy = fmaf(A.m[3], fxx, fmaf(A.m[4], fyy, A.m[5]));
                        if (!(sx >= 0.0f && sx < (float)(s.cols - 3) && sy >= 0.0f && sy < (float)(s.rows - 1))) { interior = false; break; }
                        const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
                        if (pass == 0) {
                            lo = min(lo, y0);
                            hi = max(hi, y0 + 1);
                        } else {
                            for (int k = 0; k < 2; ++k) {
                                rmin[y0 + k - lo] = min(rmin[y0 + k - lo], 3 * x0);
                                rmax[y0 + k - lo] = max(rmax[y0 + k - lo], 3 * x0 + 5);
                            }
                        }
                    }
                }
        }
        if (!interior) continue;
        int total = 0;
        for (int r = 0; r <= hi - lo; ++r)
            if (rmax[r] >= 0) total += (rmax[r] >> 4) - (rmin[r] >> 4) + 1;
        if (total > kStageChunks) return false;
        ++judged;
    }
    return judged > 0;
}

bool wrs_fits_cached(rcv_ctx* ctx, const View& s, const View& d, const Affine& A, int S)
{
    const int geom[5] = {s.rows, s.cols, d.rows, d.cols, S};
    for (const rcv_ctx::WrsEntry& e : ctx->wrs)
        if (e.valid && memcmp(e.M, A.m, sizeof(e.M)) == 0 && memcmp(e.geom, geom, sizeof(geom)) == 0) return e.ok;
    rcv_ctx::WrsEntry& e = ctx->wrs[ctx->wrs_next];
    ctx->wrs_next = (ctx->wrs_next + 1) & 3;
    e.ok = wrs_fits(s, d, A, S);
    memcpy(e.M, A.m, sizeof(e.M));
    memcpy(e.geom, geom, sizeof(geom));
    e.valid = true;
    return e.ok;
}

// host side of the staged kernel.  order: 0 raster grid, 1 XCD-contiguous runs of the whole list, 2 synchronous stripes
int wrs_launch(rcv_ctx* ctx, const View& s, const View& d, const Affine& A, int S, int fpg_, int order, int strip, unsigned extra_lds, int dbg, int occ = 5)
{
    const int gx = (d.cols + 63) / 64, gy = (d.rows + 3) / 4;
    const int fpg = fpg_ > 0 ? min(fpg_, d.n) : min(d.n, 16);
    const int groups = (d.n + fpg - 1) / fpg;
    const unsigned long long tiles = (unsigned long long)gx * gy * groups;
    if (tiles >= (1ull << 28)) return RCV_ERR_UNSUPPORTED;
    const int pg = (gx * gy + 7) / 8;
    int per = order == 1 ? (int)((tiles + 7) / 8) : (order == 2 ? -pg : 0);
    dim3 grid = order == 1 ? dim3((unsigned)(8 * per)) : (order == 2 ? dim3((unsigned)(8 * pg * groups)) : dim3((unsigned)gx, (unsigned)gy, (unsigned)groups));
    const int pad = (strip >> 16) & 1 ? 0 : 1;   // (measurement: + 65536 = row pieces packed without the spare slots)
    const int no_zfill = (strip >> 17) & 1;   // (measurement: + 131072 = border tiles on the gather path)
    strip &= 0xffff;
    if (order == 3) {   // blocks of bw x bh tiles (strip = bw + 256 * bh), dealt to the XCDs in turn
        const int bw = max(strip & 255, 1), bh = max((strip >> 8) & 255, 1);
        const long long nb = (long long)((gx + bw - 1) / bw) * ((gy + bh - 1) / bh) * groups;
        per = (1 << 30) | bw | (bh << 8);
        grid = dim3((unsigned)(((nb + 7) / 8) * 8 * bw * bh));
    }
    const unsigned lds = 2 * kStageBuf + extra_lds;
    const int zfill = (s.cols * 3) % 16 == 0 && !no_zfill ? 1 : 0;
#define WRS_GO(S_, D_) RCV_LAUNCH((k_warp_resize_stage<S_, D_>), grid, dim3(kBlock), lds, ctx->stream, s, d, A, fpg, gx, gy, groups, per, strip, pad, zfill)
    if (S == 2) WRS_GO(2, 0);
#ifdef RCV_WRL_BENCH
    else if (dbg == 1) WRS_GO(4, 1);
    else if (dbg == 2) WRS_GO(4, 2);
    else if (occ == 4) RCV_LAUNCH((k_warp_resize_stage<4, 0, 4>), grid, dim3(kBlock), lds, ctx->stream, s, d, A, fpg, gx, gy, groups, per, strip, pad, zfill);
    else if (occ == 6) RCV_LAUNCH((k_warp_resize_stage<4, 0, 6>), grid, dim3(kBlock), lds, ctx->stream, s, d, A, fpg, gx, gy, groups, per, strip, pad, zfill);
#endif
    else WRS_GO(4, 0);
#undef WRS_GO
    return rcv_launch_check(ctx);
}

#ifdef RCV_WRL_BENCH   // measurement build only (librustcv_hip_bench.so): measured, bit-exact, slower -- DESIGN.md 6.2, profiles/r05_warp_resize_*
// ---- the same launch as a FRAME LOOP (round 5) -----------------------------------------------------------------------------------
// One affine map serves every frame of a batch, so everything k_warp_resize_box computes before its first load -- the four sample
// coordinates, the interior test, floor / fraction, the tap offsets and alignment shifts: ~150 of its 278 VALU instructions per
// output pixel -- is the same for all frames.  Here a wave keeps one 64-pixel sub-tile and walks the frames [f0, f1) of its frame
// group: that state is computed once (18 registers), a frame costs its eight tap gathers (saddr form: the frame base is an SGPR
// pair, the lane offset never changes), 4 x 31 instructions of exact bilinear arithmetic and one dword store per lane.  The loop is
// software-pipelined by hand, two frames deep with two register sets (A, B): the gathers of frame f + 2 are issued before the
// arithmetic of frame f + 1, so a wave always has 8-16 gathers in flight and does not depend on occupancy to hide their latency
// (round 4's frame groups without the pipeline lost: 0.81-1.18 against 0.70 ms).
// Counting rules of gfx9's one in-order vmcnt the loop is shaped by (DESIGN_HISTORY.md 6; tests/test_isa_waits.py pins the numbers):
//  * no load or store of the loop sits behind a lane-dependent branch: every lane stores one dword of its quad's 12 bytes (lane 3
//    repeats lane 2's dword), coordinates are clamped into the image by whole quads (a clamped quad recomputes and rewrites its
//    neighbour's bytes with the same values), so ragged tiles need no bounds test;
//  * the prologue issues one store to the context's dump line between its two gather sets, so that the loop head sees the same
//    count of younger operations on the entry edge as on the back edge (set B's 8 gathers + 1 store) and waits with vmcnt(9+),
//    not for the stores just issued.
// Waves whose taps are not all inside the source (8 % at 7 degrees) run k_warp_resize_box's per-frame path.
// DBG (measurement build, -DRCV_WRL_BENCH): 1 = gathers and stores only (the taps are OR-ed into the stored dword), 2 = arithmetic and stores only
template <int S, int WW, int DBG = 0>
__global__ __launch_bounds__(kBlock) void k_warp_resize_loop(View s, View d, Affine A, int fpg, int gx, int gy, int ngroups, int tiles_per_xcd, int strip,
                                                             uint32_t* dump)
{
    int bx, by, bz;
    if (!wr_tile(tiles_per_xcd, strip, gx, gy, ngroups, bx, by, bz)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WH = 64 / WW;   // wave sub-tile WW x WH, workgroup tile 2 x 2 of them
    const int x = bx * (2 * WW) + (wave & 1) * WW + (lane % WW);
    const int y = by * (2 * WH) + (wave >> 1) * WH + (lane / WW);
    const int f0 = bz * fpg, f1 = min(f0 + fpg, d.n);
    // whole quads clamped into the image (d.cols % 4 == 0, d.cols >= 4)
    const int xq = min(x & ~3, d.cols - 4) + (x & 3), yq = min(y, d.rows - 1);
    constexpr int o = S / 2 - 1;
    float sx[4], sy[4];
    bool inter = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float fxx = (float)(S * xq + o + (i & 1)), fyy = (float)(S * yq + o + (i >> 1));
        sx[i] = fmaf(A.m[0], fxx, fmaf(A.m[1], fyy, A.m[2]));
        sy[i] = fmaf(A.m[3], fxx, fmaf(A.m[4], fyy, A.m[5]));
        inter = inter && sx[i] >= 0.0f && sx[i] < (float)(s.cols - 3) && sy[i] >= 0.0f && sy[i] < (float)(s.rows - 1);
    }
    if (!__all(inter)) {   // wave-uniform
        if (__all(x >= d.cols || y >= d.rows)) return;
        for (int f = f0; f < f1; ++f) warp_resize_box_px<S>(s, d, A, f, x, y);
        return;
    }
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    unsigned off[4], sh[4];
    f2 fxy[4];
    const unsigned sstep = (unsigned)s.step;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0f = floorf(sx[i]), y0f = floorf(sy[i]);
        fxy[i] = f2{sx[i] - x0f, sy[i] - y0f};
        const unsigned ob = __umul24((unsigned)(int)y0f, sstep) + 3u * (unsigned)(int)x0f;
        sh[i] = ob & 3u;
        off[i] = ob & ~3u;
    }
    if constexpr (DBG == 3) {   // the shared footprint: rows ymin .. ymin + 3 from column xmin, 16 aligned bytes each
        const float xm = floorf(fminf(fminf(sx[0], sx[1]), fminf(sx[2], sx[3]))), ym = floorf(fminf(fminf(sy[0], sy[1]), fminf(sy[2], sy[3])));
        const unsigned ob = __umul24((unsigned)(int)ym, sstep) + 3u * (unsigned)(int)xm;
        off[0] = ob & ~3u;
    }
    // lane k of a quad stores dword min(k, 2) of the quad's 12 bytes {b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3}: it needs the pixels
    // k' = min(k, 2) and k' + 1 of its quad (two quad_perm moves) and one v_perm with a per-lane selector
    const int kq = min(lane & 3, 2);
    const unsigned doff = (unsigned)yq * (unsigned)d.step + (unsigned)(xq & ~3) * 3u + 4u * (unsigned)kq;
    const uint32_t psel = kq == 0 ? 0x04020100u : (kq == 1 ? 0x05040201u : 0x06050402u);
    const uint8_t* sf = s.p + (size_t)f0 * s.fstride;
    uint8_t* df = d.p + (size_t)f0 * d.fstride;
    // raw buffer resources (base in SGPRs, the lane's offset in one VGPR, the second tap row through the scalar offset operand):
    // no per-frame address arithmetic on the VALU
    constexpr int kRsrc = 0x00020000;
    auto gather = [&](const uint8_t* base, u3 (&ta)[4], u3 (&tb)[4]) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xffffffff, kRsrc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (DBG == 2) {
                asm volatile("" : "=v"(ta[i]), "=v"(tb[i]));   // whatever the registers hold
                continue;
            }
            if constexpr (DBG == 3) {
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off[0], (unsigned)i * sstep, 0);
                ta[i] = u3{v.x, v.y, v.z};
                tb[i] = u3{v.w, v.w, v.w};
                continue;
            }
            ta[i] = __builtin_amdgcn_raw_buffer_load_b96(r, off[i], 0, 0);
            tb[i] = __builtin_amdgcn_raw_buffer_load_b96(r, off[i], sstep, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto finish = [&](const u3 (&ta)[4], const u3 (&tb)[4], uint8_t* dbase) {
        uint32_t p[4];
        if constexpr (DBG == 1 || DBG == 3) {
            uint32_t v = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) v |= ta[i].x | ta[i].y | ta[i].z | tb[i].x | tb[i].y | tb[i].z;
            __builtin_amdgcn_raw_buffer_store_b32(v, __builtin_amdgcn_make_buffer_rsrc((void*)dbase, 0, 0xffffffff, kRsrc), doff, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t alo = __builtin_amdgcn_alignbyte(ta[i].y, ta[i].x, sh[i]), ahi = __builtin_amdgcn_alignbyte(ta[i].z, ta[i].y, sh[i]);
            const uint32_t blo = __builtin_amdgcn_alignbyte(tb[i].y, tb[i].x, sh[i]), bhi = __builtin_amdgcn_alignbyte(tb[i].z, tb[i].y, sh[i]);
            p[i] = bilerp_bgr<true>(alo, ahi, blo, bhi, fxy[i]);
        }
        // (a+b+c+d+2)>>2 per channel: B and R ride in the two 16-bit halves of one dword, G in another
        const uint32_t br = (p[0] & 0x00ff00ffu) + (p[1] & 0x00ff00ffu) + (p[2] & 0x00ff00ffu) + (p[3] & 0x00ff00ffu) + 0x00020002u;
        const uint32_t gg = ((p[0] >> 8) & 0xffu) + ((p[1] >> 8) & 0xffu) + ((p[2] >> 8) & 0xffu) + ((p[3] >> 8) & 0xffu) + 2u;
        const uint32_t px = ((br >> 2) & 0x00ff00ffu) | ((gg >> 2) << 8);
        const uint32_t pa = __builtin_amdgcn_update_dpp(0u, px, 0xA4, 0xf, 0xf, true);   // quad_perm [0,1,2,2]
        const uint32_t pb = __builtin_amdgcn_update_dpp(0u, px, 0xF9, 0xf, 0xf, true);   // quad_perm [1,2,3,3]
        const __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc((void*)dbase, 0, 0xffffffff, kRsrc);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(pb, pa, psel), w, doff, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    u3 ta[4], tb[4], ua[4], ub_[4];
    const size_t sfs = s.fstride, dfs = d.fstride;
    gather(sf, ta, tb);
    __builtin_amdgcn_raw_buffer_store_b32(psel, __builtin_amdgcn_make_buffer_rsrc((void*)dump, 0, 0xffffffff, kRsrc), 4u * threadIdx.x, 0, 0);   // (see the counting rules above)
    __builtin_amdgcn_sched_barrier(0);
    gather(f0 + 1 < f1 ? sf + sfs : sf, ua, ub_);
    int f = f0;
#pragma unroll 1
    while (f + 3 < f1) {   // set A = frame f, set B = frame f + 1; frames f + 2 and f + 3 exist
        finish(ta, tb, df);
        gather(sf + 2 * sfs, ta, tb);
        finish(ua, ub_, df + dfs);
        gather(sf + 3 * sfs, ua, ub_);
        sf += 2 * sfs;
        df += 2 * dfs;
        f += 2;
    }
    // 1, 2 or 3 frames left: A = f, B = f + 1 (when it exists)
    finish(ta, tb, df);
    if (f + 1 < f1) {
        if (f + 2 < f1) gather(sf + 2 * sfs, ta, tb);
        finish(ua, ub_, df + dfs);
        if (f + 2 < f1) finish(ta, tb, df + 2 * dfs);
    }
}

// ---- host side of the frame-loop kernel ---------------------------------------------------------------------------------------
// fpg: frames per wave (0: by batch size); ww: wave sub-tile width; xcd: contiguous run of the tile list per XCD; strip: tile columns per
// vertical strip of that list (0: raster); lds: dynamic-LDS request that caps the workgroups per CU (0: none)
struct WrlPlan { int fpg = 0, ww = 32, xcd = 1, strip = 0; unsigned lds = 0; int dbg = 0; };

int wrl_launch(rcv_ctx* ctx, const View& s, const View& d, const Affine& A, int S, WrlPlan p)
{
    const int ww = p.ww == 16 ? 16 : (p.ww == 64 ? 64 : 32), wh = 64 / ww;
    const int gx = (d.cols + 2 * ww - 1) / (2 * ww), gy = (d.rows + 2 * wh - 1) / (2 * wh);
    int fpg = p.fpg > 0 ? min(p.fpg, d.n) : min(d.n, 16);
    const int groups = (d.n + fpg - 1) / fpg;
    const unsigned long long tiles = (unsigned long long)gx * gy * groups;
    if (tiles >= (1ull << 28)) return RCV_ERR_UNSUPPORTED;
    // xcd 1: contiguous runs of the whole list; 2: synchronous stripes (runs of every frame group's list)
    const int pg = (gx * gy + 7) / 8;
    const int tpx = p.xcd == 1 ? (int)((tiles + 7) / 8) : (p.xcd == 2 ? -pg : 0);
    const dim3 grid = p.xcd == 1 ? dim3((unsigned)(8 * tpx)) : (p.xcd == 2 ? dim3((unsigned)(8 * pg * groups)) : dim3((unsigned)gx, (unsigned)gy, (unsigned)groups));
    uint32_t* dump = (uint32_t*)(ctx->kconst + RCV_KC_SOBEL_DUMP);
    if (p.dbg == 1) { RCV_LAUNCH((k_warp_resize_loop<4, 32, 1>), grid, dim3(kBlock), p.lds, ctx->stream, s, d, A, fpg, gx, gy, groups, tpx, p.strip, dump); return rcv_launch_check(ctx); }
    if (p.dbg == 3) { RCV_LAUNCH((k_warp_resize_loop<4, 32, 3>), grid, dim3(kBlock), p.lds, ctx->stream, s, d, A, fpg, gx, gy, groups, tpx, p.strip, dump); return rcv_launch_check(ctx); }
    if (p.dbg == 2) { RCV_LAUNCH((k_warp_resize_loop<4, 32, 2>), grid, dim3(kBlock), p.lds, ctx->stream, s, d, A, fpg, gx, gy, groups, tpx, p.strip, dump); return rcv_launch_check(ctx); }
#define WRL_GO(S_, W_) RCV_LAUNCH((k_warp_resize_loop<S_, W_>), grid, dim3(kBlock), p.lds, ctx->stream, s, d, A, fpg, gx, gy, groups, tpx, p.strip, dump)
    if (S == 2) { if (ww == 16) WRL_GO(2, 16); else if (ww == 64) WRL_GO(2, 64); else WRL_GO(2, 32); }
    else { if (ww == 16) WRL_GO(4, 16); else if (ww == 64) WRL_GO(4, 64); else WRL_GO(4, 32); }
#undef WRL_GO
    return rcv_launch_check(ctx);
}

#endif   // RCV_WRL_BENCH

} // namespace

#ifndef RCV_WRL_BENCH
// resize(warp_affine(src -> mid_rows x mid_cols), dst) without materialising `mid` when mid = S * dst, S in {2, 4};
// any other shape runs the two ordinary kernels through the context workspace (same results either way).
// the one-launch forms (mid = S * dst, S in {2, 4}, BGR) on views; RCV_ERR_UNSUPPORTED (nothing enqueued) for every other shape
static int warp_resize_fused(rcv_ctx* ctx, const View& s, const View& d, const Affine& A, int mid_rows, int mid_cols);

extern "C" int rcv_warp_affine_resize_batch(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst, const float* M, int mid_rows, int mid_cols)
{
    // (as two halves on the context's two streams -- rcv_split_run -- the 32-frame launch measures +1.0 %: not split; tools/ab_split_ops.py)
    RCV_TRY(rcv_bind(ctx));
    if (!M || mid_rows < 0 || mid_cols < 0) return RCV_ERR_ARG;
    View s, d;
    RCV_TRY(check_geom(src, dst, &s, &d));
    if (d.rows == 0 || d.cols == 0 || d.n == 0) return RCV_OK;
    if (mid_rows == 0 || mid_cols == 0) return RCV_ERR_ARG;
    Affine A;
    for (int i = 0; i < 6; ++i) A.m[i] = M[i];
    {
        const int rc = warp_resize_fused(ctx, s, d, A, mid_rows, mid_cols);
        if (rc != RCV_ERR_UNSUPPORTED) return rc;
    }
    const size_t tstep = ((size_t)mid_cols * s.ch + 15) & ~(size_t)15, tfs = tstep * mid_rows;
    RCV_TRY(rcv_ws_reserve(ctx, tfs * s.n + 512));
    uint8_t* tmp;
    RCV_TRY(rcv_ws_alloc(ctx, tfs * s.n, &tmp));
    rcv_batch tb = *dst;
    tb.frame0.data = tmp;
    tb.frame0.cap = tfs;
    tb.frame0.step = tstep;
    tb.frame0.rows = mid_rows;
    tb.frame0.cols = mid_cols;
    tb.frame_stride = tfs;
    RCV_TRY(rcv_warp_affine_batch(ctx, src, &tb, M));
    return rcv_resize_batch(ctx, &tb, dst);
}

static int warp_resize_fused(rcv_ctx* ctx, const View& s, const View& d, const Affine& A, int mid_rows, int mid_cols)
{
    for (int S = 2; S <= 4; S += 2) {
        if (s.ch == 3 && s.cols >= 3 && mid_cols == S * d.cols && mid_rows == S * d.rows && d.cols % 4 == 0 && (uintptr_t)d.p % 4 == 0 &&
            d.step % 4 == 0 && d.fstride % 4 == 0 && mid_cols < (1 << 24) && mid_rows < (1 << 24)) {
            // batches of 8+ frames, 4x, maps whose tile footprints fit: the staged kernel (a tile's plan is paid once per frame group of
            // <= 12 frames: 32 frames = 3 groups of 11).  8K -> 1080p against the gather kernel, final version (tools/sweep_warp_resize_angles.sh,
            // profiles/r05_warp_resize_angles.txt, r05_warp_resize_small_batches.txt): 32 frames -12 % / -5.5 % / -17 % / -20 % at 0 / 3 / 7 / 10
            // degrees, 16 frames -13 %, 12 frames -16 %, 8 frames -10 .. -18 %, 4 frames -3 .. +4 % (the plan is not repaid: gather kernel).
            // 2x (8K -> 4K, 7 degrees): 8 frames 0.40 against 0.56 ms, 16 frames 0.80 / 1.11 (-27 %), raster order (blocks +2 %).
            if (d.n >= 8 && rcv_knobs().warp_lds != 0 && wrl_ok(s, d) && wrs_fits_cached(ctx, s, d, A, S)) {
                const int groups = (d.n + 11) / 12;
                // tiles in blocks of 2 x 4, dealt to the XCDs in turn: the lines at the ends of a tile's row pieces are hits in the L2 of the
                // XCD that runs its neighbours (FETCH 4.16 -> 3.4 GB; -1.5 .. -4 % at 0 / 3 / 7 / 10 degrees once the plan was cheap)
                return wrs_launch(ctx, s, d, A, S, (d.n + groups - 1) / groups, S == 4 ? 3 : 0, S == 4 ? 2 + 256 * 4 : 0, 0u, 0);
            }
            dim3 grid((unsigned)((d.cols + kBoxTileW - 1) / kBoxTileW), (unsigned)((d.rows + kBoxTileH - 1) / kBoxTileH), d.n);
            // occupancy cap (6 workgroups per CU through an untouched dynamic-LDS request): fewer concurrent tiles thrash the
            // rotated source footprint less -- measured 0.834 -> 0.706 ms on 32 x 8K -> 1080p (sweep: DESIGN_HISTORY.md 6)
            constexpr unsigned kLds = 27136;
            if (S == 2) RCV_LAUNCH(k_warp_resize_box<2>, grid, dim3(kBlock), kLds, ctx->stream, s, d, A, 0, 0, (int)grid.x, (int)grid.y);
            else RCV_LAUNCH(k_warp_resize_box<4>, grid, dim3(kBlock), kLds, ctx->stream, s, d, A, 0, 0, (int)grid.x, (int)grid.y);
            return rcv_launch_check(ctx);
        }
    }
    return RCV_ERR_UNSUPPORTED;
}

extern "C" int rcv_warp_affine_resize(rcv_ctx* ctx, const rcv_mat* src, rcv_mat* dst, const float* M, int mid_rows, int mid_cols)
{
    if (!src || !dst) return RCV_ERR_ARG;
    Stage st;
    RCV_TRY(stage_begin(&st, ctx));
    rcv_mat *ds, *dd;
    RCV_TRY(stage_in(&st, src, true, false, &ds));
    RCV_TRY(stage_in(&st, dst, true, true, &dd));
    rcv_batch bs = rcv_single(ds), bd = rcv_single(dd);
    return stage_finish(&st, rcv_warp_affine_resize_batch(ctx, &bs, &bd, M, mid_rows, mid_cols));
}
#endif

#ifdef RCV_WRL_BENCH
// Measurement entry (librustcv_hip_bench.so only): the fused warp -> down-scale launch with every plan parameter as an argument.
// variant 0: k_warp_resize_box (one launch per frame tile), 1: k_warp_resize_loop, 2: k_warp_resize_stage.
extern "C" int rcv__warp_resize_bench(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst, const float* M, int S, int variant, int fpg, int ww, int xcd,
                                      int strip, int lds)
{
    RCV_TRY(rcv_bind(ctx));
    View s, d;
    RCV_TRY(check_geom(src, dst, &s, &d));
    if (!M || (S != 2 && S != 4) || s.ch != 3 || d.cols % 4 || d.n == 0) return RCV_ERR_ARG;
    if (xcd & 256) s.fstride = 0;   // every frame reads frame 0: the launch without its HBM reads
    xcd &= 255;
    Affine A;
    for (int i = 0; i < 6; ++i) A.m[i] = M[i];
    if (variant == 0) {
        dim3 grid((unsigned)((d.cols + kBoxTileW - 1) / kBoxTileW), (unsigned)((d.rows + kBoxTileH - 1) / kBoxTileH), d.n);
        const int bgx = (int)grid.x, bgy = (int)grid.y, pg = (bgx * bgy + 7) / 8;
        const int ord = xcd & 3;
        const int per = ord == 1 ? (int)(((long long)bgx * bgy * d.n + 7) / 8) : (ord == 2 ? -pg : 0);
        if (ord == 1) grid = dim3((unsigned)(8 * per));
        if (ord == 2) grid = dim3((unsigned)(8 * pg * d.n));
        const unsigned l = lds < 0 ? 27136u : (unsigned)lds;
        if (S == 2) RCV_LAUNCH(k_warp_resize_box<2>, grid, dim3(kBlock), l, ctx->stream, s, d, A, per, strip, bgx, bgy);
        else RCV_LAUNCH(k_warp_resize_box<4>, grid, dim3(kBlock), l, ctx->stream, s, d, A, per, strip, bgx, bgy);
        return rcv_launch_check(ctx);
    }
    if (!wrl_ok(s, d)) return RCV_ERR_UNSUPPORTED;
    if (variant == 2) return wrs_launch(ctx, s, d, A, S, fpg, xcd & 3, strip, lds < 0 ? 0u : (unsigned)lds, (xcd >> 4) & 15, ww);   // (ww: waves per SIMD the build aims at, 4 / 6; else 5)
    WrlPlan p;
    p.fpg = fpg; p.ww = ww; p.xcd = xcd & 3; p.dbg = (xcd >> 4) & 15; p.strip = strip; p.lds = lds < 0 ? 0u : (unsigned)lds;   // (xcd + 16 * dbg)
    return wrl_launch(ctx, s, d, A, S, p);
}
#endif
