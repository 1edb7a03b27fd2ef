// rcv_filter7_mfma.hip -- filter2D with integer (i8) weights, ksize 3/5/7, u8 BGR -> u8 BGR, as a
// sliding-window stencil whose 49 MACs per sample run on the i8 matrix cores of gfx950.
//
// Why MFMA here (DESIGN_HISTORY.md §4): the op moves 6 algorithmic bytes per pixel but needs
// 147 MACs per pixel.  At the 70 %-of-HBM target (5.6 TB/s) that is 137 T MAC/s -- beyond the f32 VALU
// peak (78.6 T FMA/s) and right at the v_dot4 issue limit -- so the VALU cannot keep this kernel
// HBM-bound.  One v_mfma_i32_16x16x64_i8 per 16x16 output tile and per kernel-row PAIR does
// it at ~1/3 of the matrix pipe.  The stencil is NOT reshaped into an im2col GEMM: pixels are read once
// from HBM, staged once in LDS, and the "matrix" is a constant banded (Toeplitz) weight operand:
//
//   D[m][n] += sum_k A_p[m][k] * B_p[k][n]        (p = 0..3: kernel rows ky = 2p, 2p+1)
//     m : output x inside a 16-pixel tile            A_p[m][kyl*32 + j] = K[2p+kyl][j-m]  (0 <= j-m <= 6, else 0)
//     n : image row inside a 16-row step             B_p[kyl*32 + j][n] = plane[row n + 2p + kyl][x_tile - 3 + j]
//     lane l holds A[m=l&15][16 bytes at k=(l>>4)*16], B[same k][n=l&15], D[m=(l>>4)*4+r][n=l&15]
//   (operand layout verified on MI355X with a random 16x64x16 product, scratch probe).
//   u8 pixels enter the signed i8 MFMA as (p ^ 0x80); the accumulator starts at 128*sum(K) + round.
//
// Data movement (all global accesses are aligned vectors; all of them are unconditional -- see load_block):
//   * a workgroup (4 waves) owns a STRIP 256 px wide (16 tiles, 768 B = six whole 128-B lines per row) x a row
//     SEGMENT, and walks down it in 16-row steps; each source row is fetched from HBM once per strip (halo: 6 px per
//     256, 6 rows per segment).  A width that is not a multiple of 256 ends in a partial strip;
//   * staging: lane (row r, chunk q) loads its own aligned 48 bytes (16 pixels) as three 16-byte vectors; the chunk it
//     stages is shifted by the 3-pixel halo, so the 9 bytes in front come from the previous lane's last dwords by DPP
//     row_shr:1 (lane 0 / lane 15 of a row fetch the 12 bytes in front of / behind the strip with one shared side load).
//     v_alignbyte realigns, v_perm de-interleaves BGR -> three planar dwords x4, xor 0x80, one ds_write_b128 per plane.
//     Planar rows live in a 48-slot ring (3 blocks of 16 rows) with a 288-byte pitch (conflict-free ds_read_b128);
//     block k+3 is in flight in registers and block k+2 is written to the ring while step k is computed, one barrier
//     per step;
//   * the 6 pixels right of the strip's 16 chunks (the "halo piece") are made by lane 15 of each row from its own last
//     four pixels and the side load;
//   * BORDER_REFLECT_101 is resolved at staging: rows by picking the mirrored source row, the three left/right halo
//     pixels of the first/last strip by byte permutes in registers;
//   * epilogue: v_ashr_pk_u8_i32 does shift + saturate + pack, the 12 bytes a lane owns per tile go through a per-wave
//     LDS transpose so that every global store is a 16-byte vector in 192-byte contiguous runs.
// Measured on MI355X (4K, 64 frames, sustained clocks, DESIGN_HISTORY.md 4.1 / 5): 0.59-0.66 ms per launch = 60-67 % of 8 TB/s; the
// same launch with the MFMAs and LDS reads removed takes ~0.58 ms, loads alone ~0.27-0.31 ms, stores alone 0.29 ms.
#include "rcv_internal.h"
#include "rcv_kernels.h"
#include "rcv_device_utils.h"
#include <stdlib.h>
#include <string.h>


namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kTiles = 16;            // 16-px tiles per strip: 256 px = 768 B = six whole 128-B lines.  Strip seams that fall
                                      // inside a line make two workgroups write one line: a pure-store replica of this
                                      // traversal measured 3.5 TB/s with 720-B strips and 5.6 TB/s with 768-B strips.
constexpr int kPitch = 288;           // bytes per planar row in LDS: 18 x 16 B.  Row stride == 2 (mod 16) sixteen-byte
                                      // slots makes every ds_read_b128 lane group (8 rows at x, 8 rows at x+16) hit 16
                                      // distinct slots: conflict-free (272 B measured 46 % conflict cycles).
constexpr int kSlots = 48;            // ring of 3 blocks x 16 rows
constexpr int kPlane = kSlots * kPitch;
constexpr int kTilesG = 48;          // one-channel source: a strip is 768 px = 48 tiles = the same 768 row bytes as a BGR strip
constexpr int kPitchG = 800;         // its single LDS plane: 774 bytes used, 50 x 16 B (== 2 mod 16 like kPitch)
static_assert(kSlots * kPitchG <= 3 * kPlane, "the gray plane shares the BGR kernel's LDS");
constexpr int kLregs = 17;            // staging registers per block: 12 chunk dwords, 3 left-halo dwords, 2 halo-piece dwords
constexpr int kOutWave = 16 * 192;    // per-wave output transpose buffer: 16 rows x 4 tiles x 48 B, unpadded; the 16-B
                                      // chunks of row n are rotated by n>>2 (mod 12): dword writes and b128 reads <= 2-way

struct F7Args {
    const uint8_t* src;
    uint8_t* dst;
    const uint4* wtab;   // 4 MFMAs x 64 lanes x 16 B (A operands), device memory
    uint8_t* dump;       // 64 x 16 B scratch: masked-off lanes store here so that every store is unconditional
    size_t sstep, dstep, sfs, dfs;
    int rows, cols;
    int ntiles_total, nstrips, seg_rows, nsegs;
    int tps;             // 16-px tiles per strip: 16 (256 px, line-aligned seams) or 15 (240 px, when that divides the width evenly)
    int total_wgs, wgs_per_xcd;
    int shift, acc_init;
};

// 4 interleaved BGR pixels (3 dwords) -> planar B, G, R dwords
__device__ __forceinline__ void deint4(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t& pb, uint32_t& pg, uint32_t& pr)
{
    uint32_t t;
    t = __builtin_amdgcn_perm(d1, d0, 0x00060300u);  // b0(d0.0) b1(d0.3) b2(d1.2) x
    pb = __builtin_amdgcn_perm(d2, t, 0x05020100u);  // + b3(d2.1)
    t = __builtin_amdgcn_perm(d1, d0, 0x00070401u);  // g0(d0.1) g1(d1.0) g2(d1.3) x
    pg = __builtin_amdgcn_perm(d2, t, 0x06020100u);  // + g3(d2.2)
    t = __builtin_amdgcn_perm(d1, d0, 0x00000502u);  // r0(d0.2) r1(d1.1) x x
    pr = __builtin_amdgcn_perm(d2, t, 0x07040100u);  // + r2(d2.0) r3(d2.3)
}

struct U2 { uint32_t a, b; };
struct U3 { uint32_t a, b, c; };

// DBG: ablation bits (instantiated by hand when profiling; the product launches DBG = 0 only): 1 skip global stores, 2 skip global loads, 4 skip MFMA,
// 8 skip the staging realign/de-interleave, 16 skip the epilogue shift/saturate/pack
// DUAL: weights beyond the i8 range (integer GaussianBlur 7x7: taps up to 324) are split K = 4*Q + R with Q, R in i8;
// the same pixel operand feeds two MFMAs (tables A and A2) and the epilogue forms acc + (acc2 << 2).  LDS traffic is
// unchanged, only the matrix-pipe work doubles.
// SRC: 0 = BGR source; 1 = packed YUYV source (2 B/px): the BT.601 conversion of the reference
// (rustcv/src/videoio/mod.rs:356-363) runs at staging time, so the capture-side pipeline YUYV -> BGR -> filter2D is one
// launch and the intermediate BGR image never touches HBM (5 instead of 11 algorithmic bytes per pixel).
// SRC = 2: ONE-channel (gray) source and destination.  A gray row is the same bytes-per-strip problem as a BGR row, so the
// kernel keeps its shape: 768-byte strips, the same three aligned 16-byte loads per lane, 48 MFMAs per wave and step --
// but the strip is 768 PIXELS (48 tiles), the lane's 48 bytes go to LDS as they are (shifted by the 3 halo pixels, no
// de-interleave) into a single plane, and what the BGR kernel calls the three planes of a tile are three neighbouring tiles
// (so that dependent MFMAs still sit three issues apart).
// DUAL: 0 = one weight table; 1 = K = 4Q + R, two full tables; 2 = K = K1 + 2*T2 with T2 confined to kernel rows 2..5
// (the integer 7x7 Gaussian: only its 3x3 centre exceeds i8), so the second table costs 2 instead of 4 MFMAs per tile and plane
template <int DBG, int DUAL, int SRC = 0, bool LAT = false>
__global__ __launch_bounds__(kThreads) void k_filter7_mfma(F7Args a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[3 * kPlane + kWaves * kOutWave];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware order: hardware places block b on XCD b % 8 (speed only, never correctness).  Give each XCD a
    // contiguous run of (frame, segment, strip) ids so horizontally adjacent strips -- which share halo
    // columns and, at 720 B per strip row, partial 128-B lines -- are co-resident on one L2.
    int bid = (blockIdx.x & 7) * a.wgs_per_xcd + (blockIdx.x >> 3);
    if (bid >= a.total_wgs) return;
    const int strip = bid % a.nstrips;
    bid /= a.nstrips;
    const int seg = bid % a.nsegs;
    const int frame = bid / a.nsegs;

    const int tile0 = strip * a.tps;
    const int ntiles = min(a.tps, a.ntiles_total - tile0);
    const int x0 = tile0 * 16;
    const int ys = seg * a.seg_rows;
    const int ye = min(a.rows, ys + a.seg_rows);
    const int nsteps = (ye - ys + 15) >> 4;
    constexpr int PXB = SRC == 2 ? 1 : (SRC == 1 ? 2 : 3);   // source bytes per pixel
    constexpr int TPS = SRC == 2 ? kTilesG : kTiles;         // tiles of a full strip
    constexpr int PITCH = SRC == 2 ? kPitchG : kPitch;       // LDS row pitch
    const int rowbytes = a.cols * PXB;   // source row bytes

    const unsigned sstep24 = (unsigned)a.sstep, dstep24 = (unsigned)a.dstep;   // 24-bit multiplies run at full rate
    const uint8_t* sframe = a.src + (size_t)frame * a.sfs;
    uint8_t* dframe = a.dst + (size_t)frame * a.dfs;

    // A operands (banded weights), constant for the whole launch
    v4i A[4], A2[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        uint4 w = a.wtab[p * 64 + lane];
        A[p] = v4i{(int)w.x, (int)w.y, (int)w.z, (int)w.w};
        if (DUAL) {
            w = a.wtab[256 + p * 64 + lane];
            A2[p] = v4i{(int)w.x, (int)w.y, (int)w.z, (int)w.w};
        }
    }

    // ---- staging task of this thread: (row sr of a block, 16-pixel chunk sq of the strip) ----
    // The chunk's 16 pixels are image x = x0 - 3 + 16q .. +15, i.e. the 48 row bytes starting 9 bytes BEFORE the lane's own
    // aligned 48 bytes (row byte 3*x0 + 48q).  BGR source: the lane loads exactly its own 48 bytes as three aligned 16-byte
    // vectors -- no lane re-reads its neighbour's bytes -- and takes the 9 bytes in front from the previous lane's last three
    // dwords with DPP row_shr:1; lane 0 of a row (sq == 0) has no such neighbour and fetches the 12 bytes in front of the
    // strip itself (one dwordx3; every other lane of that instruction reads one fixed cached address).  Shifted, mutually
    // overlapping 52-byte windows per lane measured 5 % slower end to end (loads-only 5.2 vs 6.0 TB/s).
    // EVERY load below is unconditional (no branch may enclose a VMEM op in the main loop, otherwise the
    // compiler can only wait with vmcnt(0) and the register prefetch pipeline collapses): vectors that fall
    // outside the row are clamped into it -- their bytes are either reflected halo (patched in registers
    // below) or multiplied by zero weights -- and rows past the segment re-read its last row (cache hits).
    const int sr = tid >> 4, sq = tid & 15;
    // (YUYV source: the 16 pixels + 1 on each side are the 9 macropixels = 36 bytes at row byte 2*x0 - 8 + 32q.)
    const int hi = rowbytes - 16;
    const unsigned ya0 = (unsigned)min(2 * x0 + 32 * sq, hi), ya1 = (unsigned)min(2 * x0 + 32 * sq + 16, hi);   // YUYV: own 32 bytes
    const unsigned ylh = (unsigned)max(2 * x0 - 8, 0);                                                            // YUYV: 8 bytes in front of the strip
    const int xa = x0 - 3 + (SRC == 2 ? 48 : 16) * sq;   // image x of the chunk's first pixel (gray: 48-pixel chunks)
    const bool xleft = xa < 0;                          // chunk 0 of the first strip: x = -3..-1 are reflected
    // chunk `ntiles` of the last strip: x = cols..cols+2 reflected (gray: the lane whose 48-pixel chunk holds them, at tile
    // ntiles % 3 of the chunk; a full last strip has them in the halo piece instead)
    const bool xright = SRC == 2 ? (x0 + 16 * ntiles == a.cols && ntiles < TPS && sq == ntiles / 3) : xa + 3 == a.cols;
    const int ry_last = ye + 2;                         // last source row (before reflection) this segment needs
    // A full strip (16 tiles) needs 6 more pixels than its 16 chunks hold: xx = 256..261 <- x = x0+253 .. x0+258.
    // Their 18 bytes lie in the 32 bytes at row offset 3*x0 + 752; every wave fetches them for all 16 rows as one
    // dwordx2 per lane (row lane>>2, piece lane&3) -- unconditional like all VMEM here; wave 0 then plants them.
    const bool fullstrip = ntiles == TPS;
    const int wave0 = (__builtin_amdgcn_readfirstlane(wave) == 0 && fullstrip) ? 1 : 0;   // scalar
    const int er = lane >> 2, ep = lane & 3;
    const int eoff = min(max((SRC == 1 ? 2 * x0 + 504 : 3 * x0 + 752) + 8 * ep, 0), rowbytes - 8);
    const bool lastfull = fullstrip && x0 + 16 * TPS == a.cols;   // right image border inside the halo piece

    constexpr int OB = SRC == 2 ? 1 : 3;   // (the BGR / gray staging below addresses the strip in bytes: OB per pixel)
    const unsigned oa0 = (unsigned)min(OB * x0 + 48 * sq, hi), oa1 = (unsigned)min(OB * x0 + 48 * sq + 16, hi), oa2 = (unsigned)min(OB * x0 + 48 * sq + 32, hi);
    const unsigned olh = (unsigned)max(OB * x0 - 12, 0);   // the 12 bytes in front of the strip (first strip: unused, x = -3..-1 are reflected)
    const unsigned orh = (unsigned)min(OB * x0 + 768, rowbytes - 12);   // the 12 bytes behind a full strip (last strip: unused, mirrored)

    auto load_block = [&](int b, uint32_t (&L)[kLregs]) {
        const int ry = min(ys - 3 + 16 * b + sr, ry_last);
        const int srow = ry < 0 ? -ry : (ry >= a.rows ? 2 * a.rows - 2 - ry : ry);
        const unsigned ro = __umul24((unsigned)srow, sstep24);   // row offset: frame bytes < 2^32, step < 2^24 (host check)
        if (DBG & 2) {
#pragma unroll
            for (int i = 0; i < kLregs; ++i) L[i] = 0;
            return;
        }
        // YUYV source: only wave 0 plants the halo piece.  The load has to stay unconditional (a branch around it, even a
        // scalar one, collapses the counted vmcnt waits), so the other waves fetch one fixed, always-cached 8 bytes instead.
        if constexpr (SRC == 1) {
            const int rye = min(ys - 3 + 16 * b + er, ry_last);
            const int erow = rye < 0 ? -rye : (rye >= a.rows ? 2 * a.rows - 2 - rye : rye);
            const U2 e = *(const U2*)(sframe + (wave0 ? __umul24((unsigned)erow, sstep24) + (unsigned)eoff : 0u));
            L[15] = e.a;
            L[16] = e.b;
        } else {
            L[15] = L[16] = 0;
        }
        // (plain loads: the four vectors of neighbouring lanes share 128-B lines, and non-temporal loads lose
        //  that L1/L2 reuse -- measured 0.77 -> 1.03 ms)
        if constexpr (SRC == 1) {
            // YUYV: the lane's own 16 pixels are 8 macropixels = two aligned 16-byte vectors; the two macropixels in front
            // (pixels -4..-1 of the chunk) come from the previous lane by DPP, for lane 0 of a row from a side load
            const uint4 v0 = *(const uint4*)(sframe + (ro + ya0));
            const uint4 v1 = *(const uint4*)(sframe + (ro + ya1));
            const U2 lh = *(const U2*)(sframe + (sq == 0 ? ro + ylh : 0u));
            L[0] = v0.x; L[1] = v0.y; L[2] = v0.z; L[3] = v0.w;
            L[4] = v1.x; L[5] = v1.y; L[6] = v1.z; L[7] = v1.w;
            L[8] = L[9] = L[10] = L[11] = L[14] = 0;
            L[12] = lh.a; L[13] = lh.b;
        } else {
            const uint4 v0 = *(const uint4*)(sframe + (ro + oa0));
            const uint4 v1 = *(const uint4*)(sframe + (ro + oa1));
            const uint4 v2 = *(const uint4*)(sframe + (ro + oa2));
            // side load: lane 0 of a row fetches the 12 bytes in front of the strip, lane 15 the 12 bytes behind it (the
            // right halo piece), every other lane one fixed cached address
            const U3 lh = *(const U3*)(sframe + (sq == 0 ? ro + olh : (sq == 15 ? ro + orh : 0u)));
            L[0] = v0.x; L[1] = v0.y; L[2] = v0.z; L[3] = v0.w;
            L[4] = v1.x; L[5] = v1.y; L[6] = v1.z; L[7] = v1.w;
            L[8] = v2.x; L[9] = v2.y; L[10] = v2.z; L[11] = v2.w;
            L[12] = lh.a; L[13] = lh.b; L[14] = lh.c;
        }
    };

    // one YUYV macropixel [Y0 U Y1 V] -> the six pre-shift BT.601 sums b0 g0 r0 b1 g1 r1 (>> 8 and saturate follow)
    auto mp_sums = [](uint32_t m, int* o) __attribute__((always_inline)) {
        const int y0 = (int)(m & 0xff), u = (int)((m >> 8) & 0xff) - 128, y1 = (int)((m >> 16) & 0xff), v = (int)(m >> 24) - 128;
        const int c0 = 298 * (y0 - 16) + 128, c1 = 298 * (y1 - 16) + 128;
        const int db = 516 * u, dg = -100 * u - 208 * v, dr = 409 * v;
        o[0] = c0 + db; o[1] = c0 + dg; o[2] = c0 + dr;
        o[3] = c1 + db; o[4] = c1 + dg; o[5] = c1 + dr;
    };

    auto store_block = [&](int b, const uint32_t (&L)[kLregs]) {
        // ---- halo piece: the 6 pixels x0+253 .. x0+258 right of the strip's 16 chunks, planted at xx = 256..261 ----
        if constexpr (SRC == 2) {
            // gray: the row's last lane holds pixels x0+764..767 in its dword 11 and has fetched x0+768..779 with the side load
            if (fullstrip && sq == 15 && ys - 3 + 16 * b + sr <= ry_last) {
                const int hslot = 16 * (b % 3) + sr;
                const uint32_t g1 = L[11], g2 = L[12];
                const uint32_t lo = lastfull ? __builtin_amdgcn_perm(g1, g1, 0x02030201u)   // 765 766 767 | 766
                                             : __builtin_amdgcn_perm(g2, g1, 0x04030201u);  // 765 766 767 | 768
                const uint32_t hi2 = lastfull ? __builtin_amdgcn_perm(g1, g1, 0x0c0c0001u)  // 765 764
                                              : (g2 >> 8);                                   // 769 770
                *(U2*)(lds + hslot * PITCH + 768) = U2{lo ^ 0x80808080u, hi2 ^ 0x80808080u};
            }
        } else if constexpr (SRC == 0) {
            // The last lane of each row (sq == 15) holds pixels x0+252..255 in its own dwords 9..11 and has fetched the 12
            // bytes behind the strip (pixels x0+256..259) with the side load: two de-interleaves, no extra traffic.
            if (fullstrip && sq == 15 && ys - 3 + 16 * b + sr <= ry_last) {   // (no VMEM inside: LDS ops may be conditional)
                uint32_t g1[3], g2[3];
                deint4(L[9], L[10], L[11], g1[0], g1[1], g1[2]);     // pixels x0+252 .. 255
                deint4(L[12], L[13], L[14], g2[0], g2[1], g2[2]);    // pixels x0+256 .. 259
                const int hslot = 16 * (b % 3) + sr;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint32_t lo = lastfull ? __builtin_amdgcn_perm(g1[c], g1[c], 0x02030201u)   // 253 254 255 | 254
                                                 : __builtin_amdgcn_perm(g2[c], g1[c], 0x04030201u);  // 253 254 255 | 256
                    const uint32_t hi = lastfull ? __builtin_amdgcn_perm(g1[c], g1[c], 0x0c0c0001u)   // 253 252
                                                 : (g2[c] >> 8);                                      // 257 258
                    *(U2*)(lds + c * kPlane + hslot * kPitch + 256) = U2{lo ^ 0x80808080u, hi ^ 0x80808080u};
                }
            }
        } else if (fullstrip && wave == 0) {
            // YUYV source (wave 0 only): the four lanes of a row hold bytes [0,32) of the piece; lanes ep = 0, 1 of the quad
            // hold macropixels (252,253)(254,255) | (256,257)(258,259).  Lane ep == 0 collects them with DPP row_shl,
            // converts, and plants them (mirrored at the right image border).
            uint32_t g1[3], g2[3];
            const uint32_t m2 = __builtin_amdgcn_update_dpp(0u, L[15], 0x101, 0xf, 0xf, true);
            const uint32_t m3 = __builtin_amdgcn_update_dpp(0u, L[16], 0x101, 0xf, 0xf, true);
            int q[24];
            mp_sums(L[15], q);
            mp_sums(L[16], q + 6);
            mp_sums(m2, q + 12);
            mp_sums(m3, q + 18);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                g1[c] = rcv_ashr_sat_pk4(q[c], q[3 + c], q[6 + c], q[9 + c], 8);
                g2[c] = rcv_ashr_sat_pk4(q[12 + c], q[15 + c], q[18 + c], q[21 + c], 8);
            }
            if (ep == 0 && ys - 3 + 16 * b + er <= ry_last) {
                const int hslot = 16 * (b % 3) + er;   // ring slot of source row 16b + er (the block base is scalar)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint32_t lo = lastfull ? __builtin_amdgcn_perm(g1[c], g1[c], 0x02030201u)   // 253 254 255 | 254
                                                 : __builtin_amdgcn_perm(g2[c], g1[c], 0x04030201u);  // 253 254 255 | 256
                    const uint32_t hi = lastfull ? __builtin_amdgcn_perm(g1[c], g1[c], 0x0c0c0001u)   // 253 252
                                                 : (g2[c] >> 8);                                      // 257 258
                    *(U2*)(lds + c * kPlane + hslot * kPitch + 256) = U2{lo ^ 0x80808080u, hi ^ 0x80808080u};
                }
            }
        }
        if constexpr (SRC == 2) {
            // s[i] = bytes [4i - 3, 4i + 1) relative to the lane's own 48 bytes: the previous lane's last dword by DPP (lane 0 of
            // a row: the last dword of its side load), one v_alignbyte per dword, no de-interleave
            const uint32_t pv = __builtin_amdgcn_update_dpp(L[14], L[11], 0x111, 0xf, 0xf, false);
            uint32_t g[12];
            g[0] = __builtin_amdgcn_alignbyte(L[0], pv, 1);
#pragma unroll
            for (int i = 1; i < 12; ++i) g[i] = __builtin_amdgcn_alignbyte(L[i], L[i - 1], 1);
            // BORDER_REFLECT_101 in x: left as in the BGR kernel; right: x = cols..cols+2 are bytes 3 | 0, 1 of dwords 4m | 4m+1
            // (m = ntiles % 3, uniform) and mirror x = cols-2, cols-3 (bytes 1, 0 of dword 4m) and cols-4 (byte 3 of the dword in
            // front: the previous lane's last one for m == 0)
            const uint32_t ng = __builtin_amdgcn_update_dpp(0u, g[11], 0x111, 0xf, 0xf, true);
            if (xleft) g[0] = __builtin_amdgcn_perm(g[1], g[0], 0x03040506u);
            if (xright) {
                const int m = ntiles % 3;
#pragma unroll
                for (int mm = 0; mm < 3; ++mm)
                    if (m == mm) {
                        const uint32_t prev = mm == 0 ? ng : g[4 * mm - 1];
                        const uint32_t t = __builtin_amdgcn_perm(prev, g[4 * mm], 0x00000700u);
                        g[4 * mm + 1] = __builtin_amdgcn_perm(g[4 * mm + 1], t, 0x07060100u);
                        g[4 * mm] = __builtin_amdgcn_perm(g[4 * mm], g[4 * mm], 0x01020100u);
                    }
            }
            if (ys - 3 + 16 * b + sr > ry_last) return;   // row not needed (LDS ops may be conditional)
            uint8_t* dstp = lds + (16 * (b % 3) + sr) * PITCH + 48 * sq;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                *(uint4*)(dstp + 16 * j) = make_uint4(g[4 * j] ^ 0x80808080u, g[4 * j + 1] ^ 0x80808080u, g[4 * j + 2] ^ 0x80808080u, g[4 * j + 3] ^ 0x80808080u);
            return;
        }
        uint32_t pb[4], pg[4], pr[4];
        if constexpr (SRC == 1) {
            // 9 macropixels = pixels x0-4+16q .. x0+13+16q (two from the previous lane, seven of the lane's own eight); the
            // chunk is pixels 1..16 of those 18
            const uint32_t pm0 = __builtin_amdgcn_update_dpp(L[12], L[6], 0x111, 0xf, 0xf, false);
            const uint32_t pm1 = __builtin_amdgcn_update_dpp(L[13], L[7], 0x111, 0xf, 0xf, false);
            int q[54];
            mp_sums(pm0, q);
            mp_sums(pm1, q + 6);
#pragma unroll
            for (int m = 2; m < 9; ++m) mp_sums(L[m - 2], q + 6 * m);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 3 * (1 + 4 * i);
                pb[i] = rcv_ashr_sat_pk4(q[e], q[e + 3], q[e + 6], q[e + 9], 8);
                pg[i] = rcv_ashr_sat_pk4(q[e + 1], q[e + 4], q[e + 7], q[e + 10], 8);
                pr[i] = rcv_ashr_sat_pk4(q[e + 2], q[e + 5], q[e + 8], q[e + 11], 8);
            }
        } else {
            if (DBG & 8) {   // ablation: no realign / de-interleave
#pragma unroll
                for (int i = 0; i < 4; ++i) { pb[i] = L[i]; pg[i] = L[4 + i]; pr[i] = L[8 + i]; }
            } else {
                // the previous lane's dwords 9..11 (its last 12 bytes); lane 0 of each row keeps `old` = its own left-halo load
                const uint32_t pv0 = __builtin_amdgcn_update_dpp(L[12], L[9], 0x111, 0xf, 0xf, false);
                const uint32_t pv1 = __builtin_amdgcn_update_dpp(L[13], L[10], 0x111, 0xf, 0xf, false);
                const uint32_t pv2 = __builtin_amdgcn_update_dpp(L[14], L[11], 0x111, 0xf, 0xf, false);
                uint32_t s[12];   // s[i] = bytes [4i - 9, 4i - 5) relative to the lane's own 48 bytes
                s[0] = __builtin_amdgcn_alignbyte(pv1, pv0, 3);
                s[1] = __builtin_amdgcn_alignbyte(pv2, pv1, 3);
                s[2] = __builtin_amdgcn_alignbyte(L[0], pv2, 3);
#pragma unroll
                for (int i = 3; i < 12; ++i) s[i] = __builtin_amdgcn_alignbyte(L[i - 2], L[i - 3], 3);
#pragma unroll
                for (int i = 0; i < 4; ++i) deint4(s[3 * i], s[3 * i + 1], s[3 * i + 2], pb[i], pg[i], pr[i]);
            }
        }
        // BORDER_REFLECT_101 in x, in registers (first / last strip only; branch-free selects elsewhere).
        //  left : chunk 0 holds x = -3..12 ; x = -3,-2,-1 mirror x = 3,2,1 = bytes 6,5,4 of the same chunk.
        //  right: chunk `ntiles` holds x = cols-3..cols+12 ; x = cols,cols+1,cols+2 (bytes 3,4,5) mirror
        //         x = cols-2,cols-3,cols-4 = bytes 1, 0 of this chunk and byte 15 of the previous chunk (lane - 1).
        const uint32_t nb = __builtin_amdgcn_update_dpp(0u, pb[3], 0x111, 0xf, 0xf, true);  // row_shr:1 -> lane-1's dword
        const uint32_t ng = __builtin_amdgcn_update_dpp(0u, pg[3], 0x111, 0xf, 0xf, true);
        const uint32_t nr = __builtin_amdgcn_update_dpp(0u, pr[3], 0x111, 0xf, 0xf, true);
        if (xleft) {
            pb[0] = __builtin_amdgcn_perm(pb[1], pb[0], 0x03040506u);
            pg[0] = __builtin_amdgcn_perm(pg[1], pg[0], 0x03040506u);
            pr[0] = __builtin_amdgcn_perm(pr[1], pr[0], 0x03040506u);
        }
        if (xright) {
            uint32_t t;
            t = __builtin_amdgcn_perm(nb, pb[0], 0x00000700u);
            pb[1] = __builtin_amdgcn_perm(pb[1], t, 0x07060100u);
            pb[0] = __builtin_amdgcn_perm(pb[0], pb[0], 0x01020100u);
            t = __builtin_amdgcn_perm(ng, pg[0], 0x00000700u);
            pg[1] = __builtin_amdgcn_perm(pg[1], t, 0x07060100u);
            pg[0] = __builtin_amdgcn_perm(pg[0], pg[0], 0x01020100u);
            t = __builtin_amdgcn_perm(nr, pr[0], 0x00000700u);
            pr[1] = __builtin_amdgcn_perm(pr[1], t, 0x07060100u);
            pr[0] = __builtin_amdgcn_perm(pr[0], pr[0], 0x01020100u);
        }
        if (sq > ntiles || ys - 3 + 16 * b + sr > ry_last) return;  // chunk / row not needed (LDS ops may be conditional)
        const int slot = 16 * (b % 3) + sr;
        uint8_t* dstp = lds + slot * kPitch + 16 * sq;
        *(uint4*)(dstp) = make_uint4(pb[0] ^ 0x80808080u, pb[1] ^ 0x80808080u, pb[2] ^ 0x80808080u, pb[3] ^ 0x80808080u);
        *(uint4*)(dstp + kPlane) = make_uint4(pg[0] ^ 0x80808080u, pg[1] ^ 0x80808080u, pg[2] ^ 0x80808080u, pg[3] ^ 0x80808080u);
        *(uint4*)(dstp + 2 * kPlane) = make_uint4(pr[0] ^ 0x80808080u, pr[1] ^ 0x80808080u, pr[2] ^ 0x80808080u, pr[3] ^ 0x80808080u);
    };

    const int n = lane & 15, kb = lane >> 4;
    const int kyl = kb >> 1, xh = (kb & 1) * 16;
    uint8_t* const dumpp = a.dump + lane * 16;
    uint8_t* const obuf = lds + 3 * kPlane + wave * kOutWave;

    // Output transpose through LDS (per wave, no barrier: LDS ops of one wave execute in order).
    // After the MFMAs lane (n, kb) holds, for tile i, the 12 bytes [48i + 12kb, +12) of output row n of the wave's
    // 192-byte row segment.  They are written as dwords, then each lane reads back three 16-B chunks (chunk
    // f = lane + 64j: row f/12, column chunk f%12) so that every global store is a full 16-B vector and a wave
    // instruction covers 192-B contiguous runs (48-B runs measured 3.5 TB/s of pure store rate).
    int woff[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int d = SRC == 2 ? 12 * i + 4 * j + kb : 12 * i + 3 * kb + j;   // gray: dword kb of tile j of group i
            woff[i][j] = n * 192 + (((d >> 2) + (n >> 2)) % 12) * 16 + (d & 3) * 4;
        }
    int roff[3], grow[3], gcol[3];
    bool gtile[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int f = lane + 64 * j, row = f / 12, c = f % 12;
        roff[j] = row * 192 + ((c + (row >> 2)) % 12) * 16;
        grow[j] = row;
        gcol[j] = SRC == 2 ? x0 + 192 * wave + 16 * c : 3 * (x0 + 64 * wave) + 16 * c;
        gtile[j] = (SRC == 2 ? 12 * wave + c : 4 * wave + c / 3) < ntiles;
    }

    // accumulator start value 128*sum(K) + round as a resident VGPR quad (opaque to the compiler, which would otherwise
    // rebuild it from the scalar before each of the 12 first-MFMAs of a step: 24 moves)
    v4i initv = v4i{a.acc_init, a.acc_init, a.acc_init, a.acc_init};
    asm volatile("" : "+v"(initv));

    auto compute = [&](int k) {
        // ring slot of source row (16k + n + 2p + kyl): the block base is wave-uniform (scalar), the rest needs one
        // conditional subtract; 24-bit multiply (full rate) instead of a 32-bit one (quarter rate)
        const int sbase = (16 * k) % kSlots;
        int off[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int t = sbase + n + 2 * p + kyl;
            t = t >= kSlots ? t - kSlots : t;
            off[p] = (int)__umul24((unsigned)t, (unsigned)PITCH) + xh + (SRC == 2 ? 192 : 64) * wave;
        }
        // 48 MFMAs per step (4 tiles x 4 row-pairs x 3 planes), B operands read kAhead items ahead into a
        // static register ring so LDS latency is covered inside the wave; the three planes' accumulators are
        // interleaved so dependent MFMAs sit three issues apart.
        constexpr int kAhead = DUAL ? 5 : 9;   // (the dual-table kernel carries 28 more registers: a shorter read-ahead keeps it at 3 waves per SIMD)
        v4i Bq[kAhead];
        // p == 3 pairs kernel row 6 with the non-existent row 7: the upper 32 lanes (kyl == 1) would read pixels that
        // only meet zero weights, so they skip the LDS read (half the LDS cycles of that instruction).
        const bool lowhalf = kyl == 0;
        // (the idle upper lanes keep whatever the ring register held before -- `old` -- so the masked read needs no
        //  zero-fill moves: garbage times zero weights is still zero in integer arithmetic)
        auto rd = [&](int it, const v4i& old) -> v4i {
            const int i = it / 12, r = it % 12, p = r / 3, c = r % 3;
            const v4i* src = (const v4i*)(lds + (SRC == 2 ? 16 * (3 * i + c) : c * kPlane + 16 * i) + off[p]);
            if (p == 3) return lowhalf ? *src : old;
            return *src;
        };
        // (wave-uniform, LDS-only region -- no VMEM inside, so the vmcnt bookkeeping stays exact)
        // narrow last strips: tiles past the strip are skipped (gray: groups of three tiles)
        const int nmf = 12 * min(max(SRC == 2 ? (ntiles - 12 * wave + 2) / 3 : ntiles - 4 * wave, 0), 4);
        v4i acc[3], acc2[3];
        const v4i zerov = v4i{0, 0, 0, 0};
        // the MFMA phase is the longest dependent chain of a step: give it issue priority over the staging phases of the
        // other workgroups' waves on this SIMD (measured -2 %)
        __builtin_amdgcn_s_setprio(3);
        if (!(DBG & 4)) {
#pragma unroll
            for (int it = 0; it < kAhead; ++it) Bq[it] = rd(it, zerov);
        }
#pragma unroll
        for (int it = 0; it < 48; ++it) {
            const int i = it / 12, r = it % 12, p = r / 3, c = r % 3;
            if (r == 0 && it >= nmf) break;
            if (DBG & 4) {
                if (r < 3) acc[c] = initv;
            } else {
                // the first MFMA of each accumulator takes the constant (128*sum(K) + round) vector as its C operand
                acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[p], Bq[it % kAhead], r < 3 ? initv : acc[c], 0, 0, 0);
                if (DUAL == 1 || (DUAL == 2 && (p == 1 || p == 2)))
                    acc2[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A2[p], Bq[it % kAhead], (DUAL == 1 ? r < 3 : p == 1) ? zerov : acc2[c], 0, 0, 0);
                if (it + kAhead < 48) Bq[it % kAhead] = rd(it + kAhead, Bq[it % kAhead]);
            }
            if (r == 11) {
                if (DUAL) {
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) acc[cc] += acc2[cc] << (DUAL == 1 ? 2 : 1);
                }
                // lane holds x = x0 + 16t + 4*kb + {0,1,2,3} of row n (t = 4*wave + i): 12 interleaved bytes
                if (DBG & 16) {   // ablation: no shift / saturate / pack
                    *(uint32_t*)(obuf + woff[i][0]) = acc[0][0];
                    *(uint32_t*)(obuf + woff[i][1]) = acc[1][1];
                    *(uint32_t*)(obuf + woff[i][2]) = acc[2][2];
                    continue;
                }
                if constexpr (SRC == 2) {   // three tiles, four consecutive pixels each
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) *(uint32_t*)(obuf + woff[i][cc]) = rcv_ashr_sat_pk4(acc[cc][0], acc[cc][1], acc[cc][2], acc[cc][3], a.shift);
                    continue;
                }
                *(uint32_t*)(obuf + woff[i][0]) = rcv_ashr_sat_pk4(acc[0][0], acc[1][0], acc[2][0], acc[0][1], a.shift);
                *(uint32_t*)(obuf + woff[i][1]) = rcv_ashr_sat_pk4(acc[1][1], acc[2][1], acc[0][2], acc[1][2], a.shift);
                *(uint32_t*)(obuf + woff[i][2]) = rcv_ashr_sat_pk4(acc[2][2], acc[0][3], acc[1][3], acc[2][3], a.shift);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        const int ybase = ys + 16 * k;
        uint4 ov[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) ov[j] = *(const uint4*)(obuf + roff[j]);   // three reads in flight, then three stores
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint4 v = ov[j];
            const int y = ybase + grow[j];
            // rows past the segment and tiles past the strip go to the dump line instead (the store stays unconditional)
            uint8_t* dp = (y < ye && gtile[j]) ? dframe + (__umul24((unsigned)y, dstep24) + (unsigned)gcol[j]) : dumpp;
            if (DBG & 1) {
                if (v.x == 0x12345678u && v.y == 0x9abcdef0u) *(uint4*)dumpp = v;
            } else {
                *(uint4*)dp = v;
            }
        }
    };

    // ---- software pipeline: block k+3 is in flight in registers and block k+2 is written to the LDS ring while
    // step k computes from blocks k, k+1.  Blocks 0..nsteps are needed; loads past that are harmless re-reads.
    // Two register sets, loop unrolled by two so that the set index is static. ----
    uint32_t LA[kLregs], LB[kLregs];
    if constexpr (LAT) {
        // Latency variant for launches that fill the GPU at most once (a single 1080p frame is 544 workgroups of 16 rows):
        // segments of <= 32 rows, so blocks 0..2 are all a workgroup ever needs -- requested together (ONE memory latency
        // instead of two dependent ones), staged, and computed without any pipeline.
        uint32_t LC[kLregs];
        load_block(0, LA);
        load_block(1, LB);
        load_block(2, LC);
        store_block(0, LA);
        store_block(1, LB);
        store_block(2, LC);
        __syncthreads();
        compute(0);
        if (nsteps > 1) compute(1);
        return;
    }
    load_block(0, LA);
    load_block(1, LB);
    store_block(0, LA);
    store_block(1, LB);
    load_block(2, LA);
    // Three dummy stores: inside the loop the loads of a block are followed by the three output stores of the step
    // before the next loads are issued.  Reproducing that order here makes the vmcnt bookkeeping of the loop-entry
    // path identical to the back-edge path; without it the compiler merges the two conservatively and the first
    // half-step of every iteration waits for its own output stores to be acknowledged (vmcnt(5) instead of (8)).
#pragma unroll
    for (int j = 0; j < 3; ++j) *(uint4*)(dumpp + 1024 * j) = make_uint4(0u, 0u, 0u, 0u);   // dump area: 3 KiB at kconst + 16 KiB
    __syncthreads();

    for (int k = 0; k < nsteps; k += 2) {
        load_block(k + 3, LB);
        store_block(k + 2, LA);   // block k+2's ring slots are free during step k: stage first, compute after
        compute(k);
        __syncthreads();
        load_block(k + 4, LA);
        store_block(k + 3, LB);
        compute(k + 1);   // (when nsteps is odd this last half-iteration computes a step past the segment: all dumped)
        __syncthreads();
    }
}

// host: banded A operands.  K7 is the kernel embedded (centred) in 7x7; `part` selects R (0), Q (1) of K = 4Q + R, the
// weights themselves (2, all within i8), or K1 (3), T2 (4) of the centre split K = K1 + 2*T2.
void build_wtab(const int16_t* k, int ksize, int part, int8_t* tab /*4*64*16*/)
{
    int K7[7][7];
    memset(K7, 0, sizeof(K7));
    int o = (7 - ksize) / 2;
    for (int y = 0; y < ksize; ++y)
        for (int x = 0; x < ksize; ++x) {
            int w = k[y * ksize + x];
            int q = w >> 2;  // floor
            // centre split: T2 takes half of what lies beyond the i8 range (rounded away from zero), K1 = w - 2*T2 stays inside
            int t2 = w > 127 ? (w - 127 + 1) / 2 : (w < -128 ? -((-w - 128 + 1) / 2) : 0);
            K7[y + o][x + o] = part == 2 ? w : (part == 1 ? q : (part == 0 ? w - 4 * q : (part == 4 ? t2 : w - 2 * t2)));
        }
    for (int p = 0; p < 4; ++p)
        for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 16; ++i) {
                int m = lane & 15, kb = lane >> 4, kyl = kb >> 1, j = (kb & 1) * 16 + i;
                int ky = 2 * p + kyl, tap = j - m;
                tab[(p * 64 + lane) * 16 + i] = (int8_t)((ky < 7 && tap >= 0 && tap <= 6) ? K7[ky][tap] : 0);
            }
}

} // namespace

extern "C" int rcv__debug_occupancy(void)
{
    int nb = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_filter7_mfma<0, 0>, kThreads, 0) != hipSuccess) return -1;
    return nb;
}

// weights k16 in [-512, 511]; those within [-128, 127] run the single-MFMA kernel
int rcv_filter_i16_fast(rcv_ctx* ctx, const View& s, const View& d, const int16_t* k, int ksize, int shift, int src_yuyv)
{
    if (ksize != 3 && ksize != 5 && ksize != 7) return RCV_ERR_UNSUPPORTED;
    // one channel: the strip kernel's gray variant (SRC = 2) where the shape allows, else the dot4 streaming kernel
    const bool gray = !src_yuyv && s.ch == 1;
    if (gray) {
        const bool ok = d.ch == 1 && s.cols % 16 == 0 && s.cols >= 16 && s.rows >= 4 && (uintptr_t)s.p % 16 == 0 && s.step % 16 == 0 &&
                        (s.n <= 1 || s.fstride % 16 == 0) && (uintptr_t)d.p % 16 == 0 && d.step % 16 == 0 && (d.n <= 1 || d.fstride % 16 == 0) &&
                        s.step < (1u << 24) && d.step < (1u << 24) && s.rows < (1 << 24) && (unsigned long long)s.rows * s.step < (1ull << 32) &&
                        (unsigned long long)s.rows * d.step < (1ull << 32) && !rcv_knobs().f7_no_gray;
        if (!ok) return rcv_filter_i16_gray(ctx, s, d, k, ksize, shift);
    } else {
        if (s.ch != (src_yuyv ? 2 : 3) || d.ch != 3) return RCV_ERR_UNSUPPORTED;
    }
    if (!gray && !src_yuyv && (s.cols % 16 != 0 || (uintptr_t)s.p % 16 || s.step % 16 || (s.n > 1 && s.fstride % 16) || (uintptr_t)d.p % 16 || d.step % 16 ||
                               (d.n > 1 && d.fstride % 16)))
        // BGR widths that are a multiple of 4 with 4-byte aligned rows (a packed 1080-pixel-wide frame): the row-streaming kernel
        // takes them whatever the size of the launch (one 1080 x 1920 frame: 0.015 ms against 0.088 ms on the streaming VALU
        // kernel); this strip kernel does not
        return rcv_filter_i16_rows(ctx, s, d, k, ksize, shift, 0, true);
    if (s.cols % 16 != 0 || s.cols < 16 || s.rows < 4) return RCV_ERR_UNSUPPORTED;
    if ((uintptr_t)s.p % 16 || s.step % 16 || (s.n > 1 && s.fstride % 16)) return RCV_ERR_UNSUPPORTED;
    if ((uintptr_t)d.p % 16 || d.step % 16 || (d.n > 1 && d.fstride % 16)) return RCV_ERR_UNSUPPORTED;
    // in-frame byte offsets are formed with 24-bit multiplies and kept in 32 bits
    if (s.step >= (1u << 24) || d.step >= (1u << 24) || s.rows >= (1 << 24)) return RCV_ERR_UNSUPPORTED;
    if ((unsigned long long)s.rows * s.step >= (1ull << 32) || (unsigned long long)s.rows * d.step >= (1ull << 32)) return RCV_ERR_UNSUPPORTED;
    bool dual = false, centre = true;   // centre: every weight beyond i8 sits in (embedded) kernel rows 2..5 and splits as K1 + 2*T2
    long long ksum = 0;
    for (int i = 0; i < ksize * ksize; ++i) {
        if (k[i] < -512 || k[i] > 511) return RCV_ERR_UNSUPPORTED;
        if (k[i] < -128 || k[i] > 127) {
            dual = true;
            const int ky = i / ksize + (7 - ksize) / 2;
            if (ky < 2 || ky > 5 || k[i] > 127 + 2 * 127 || k[i] < -128 - 2 * 128) centre = false;
        }
        ksum += k[i];
    }
    const int mode = !dual ? 0 : ((centre && !rcv_knobs().f7_dual_full) ? 2 : 1);   // (the knob keeps the full-table kernel testable)
    // BGR, enough rows to fill the GPU: the row-streaming kernel (rcv_filter_rows_mfma.hip), one or two weight tables
    {
        const int rc = rcv_filter_i16_rows(ctx, s, d, k, ksize, shift, gray ? 2 : src_yuyv);
        if (rc != RCV_ERR_UNSUPPORTED) return rc;
    }

    // weight tables: rebuilt/uploaded only when the kernel changes (the upload is stream-ordered)
    const uint8_t* wtab = ctx->kconst + RCV_KC_F7_TAB;
    if (!ctx->f7_valid || ctx->f7_ksize != ksize || ctx->f7_mode != mode || memcmp(ctx->f7_k, k, (size_t)ksize * ksize * sizeof(int16_t)) != 0) {
        int8_t tab[2 * 4 * 64 * 16];
        build_wtab(k, ksize, mode == 0 ? 2 : (mode == 1 ? 0 : 3), tab);
        if (dual) build_wtab(k, ksize, mode == 1 ? 1 : 4, tab + 4096);
        ctx->f7_valid = false;
        RCV_TRY(rcv_upload_const(ctx, tab, dual ? sizeof(tab) : sizeof(tab) / 2, RCV_KC_F7_TAB));
        RCV_HIP(hipStreamSynchronize(ctx->stream)); // `tab` is on this stack frame
        memcpy(ctx->f7_k, k, (size_t)ksize * ksize * sizeof(int16_t));
        ctx->f7_ksize = ksize;
        ctx->f7_mode = mode;
        ctx->f7_valid = true;
    }

    F7Args a;
    a.src = s.p;
    a.dst = d.p;
    a.wtab = (const uint4*)wtab;
    a.dump = ctx->kconst + RCV_KC_F7_DUMP;
    a.sstep = s.step;
    a.dstep = d.step;
    a.sfs = s.fstride;
    a.dfs = d.fstride;
    a.rows = s.rows;
    a.cols = s.cols;
    a.ntiles_total = s.cols / 16;
    // strips of 256 px: line-aligned seams.  (240-px strips split widths such as 1920 evenly, but measured 3.5 % slower there
    // than 7 full strips + one half strip.)
    a.tps = gray ? kTilesG : kTiles;
    a.nstrips = (a.ntiles_total + a.tps - 1) / a.tps;
    // row segments: a few waves of 3 workgroups per CU with little tail (total close to a multiple of 3 * CUs), each
    // segment a multiple of 16 rows (>= 32); the per-segment constant models the prologue (two synchronous blocks).
    // The plan depends on the launch geometry only and is cached in the context (a 1080p frame is a 6-us launch).
    int seg_rows;
    bool lat = false;
    const bool lat_ok = !src_yuyv && !dual && !rcv_knobs().f7_no_lat;
    if (ctx->f7_plan_rows == s.rows && ctx->f7_plan_nstrips == a.nstrips && ctx->f7_plan_n == s.n && ctx->f7_plan_lat_ok == lat_ok) {
        seg_rows = ctx->f7_plan_seg_rows;
        lat = ctx->f7_plan_lat;
    } else {
        seg_rows = (s.rows + 15) & ~15;
        const long long slots = 3LL * ctx->cu_count;
        double best = 1e30;
        for (int ns = 1; ns <= 64; ++ns) {
            int sr = ((s.rows + ns - 1) / ns + 15) & ~15;
            if (ns > 1 && sr < 32) break;
            int nsegs = (s.rows + sr - 1) / sr;
            long long tot = (long long)a.nstrips * nsegs * s.n;
            long long rounds = (tot + slots - 1) / slots;
            // cost model: time ~ rounds * rows-per-segment (+6 halo rows), slight penalty per segment
            double cost = (double)rounds * (sr + 6 + 40);
            if (cost < best) { best = cost; seg_rows = sr; }
        }
        // small launches (everything resident in one round even with the shortest segments): the latency variant
        if (lat_ok) {
            for (int sr = 16; sr <= 32 && !lat; sr += 16) {
                const long long tot = (long long)a.nstrips * ((s.rows + sr - 1) / sr) * s.n;
                if (tot <= 3LL * ctx->cu_count) {
                    seg_rows = sr;
                    lat = true;
                }
            }
        }
        ctx->f7_plan_rows = s.rows;
        ctx->f7_plan_nstrips = a.nstrips;
        ctx->f7_plan_n = s.n;
        ctx->f7_plan_lat_ok = lat_ok;
        ctx->f7_plan_seg_rows = seg_rows;
        ctx->f7_plan_lat = lat;
    }
    a.seg_rows = seg_rows;
    a.nsegs = (s.rows + seg_rows - 1) / seg_rows;
    a.shift = shift;
    a.acc_init = (int)(128 * ksum + (shift > 0 ? (1 << (shift - 1)) : 0));
    long long total = (long long)a.nstrips * a.nsegs * s.n;
    if (total > 0x3fffffffLL) return RCV_ERR_UNSUPPORTED;
    a.total_wgs = (int)total;
    a.wgs_per_xcd = (int)((total + 7) / 8);
    const dim3 grid((unsigned)(a.wgs_per_xcd * 8)), block(kThreads);
    if (gray) {
        if (lat) RCV_LAUNCH((k_filter7_mfma<0, 0, 2, true>), grid, block, 0, ctx->stream, a);
        else if (mode == 2) RCV_LAUNCH((k_filter7_mfma<0, 2, 2>), grid, block, 0, ctx->stream, a);
        else if (mode == 1) RCV_LAUNCH((k_filter7_mfma<0, 1, 2>), grid, block, 0, ctx->stream, a);
        else RCV_LAUNCH((k_filter7_mfma<0, 0, 2>), grid, block, 0, ctx->stream, a);
        return rcv_launch_check(ctx);
    }
    if (src_yuyv) {
        if (dual) return RCV_ERR_UNSUPPORTED;
        RCV_LAUNCH((k_filter7_mfma<0, 0, 1>), grid, block, 0, ctx->stream, a);
        return rcv_launch_check(ctx);
    }
    if (dual) {
        if (mode == 2) RCV_LAUNCH((k_filter7_mfma<0, 2>), grid, block, 0, ctx->stream, a);
        else RCV_LAUNCH((k_filter7_mfma<0, 1>), grid, block, 0, ctx->stream, a);
        return rcv_launch_check(ctx);
    }
    if (lat) {
        RCV_LAUNCH((k_filter7_mfma<0, 0, 0, true>), grid, block, 0, ctx->stream, a);
        return rcv_launch_check(ctx);
    }
    RCV_LAUNCH((k_filter7_mfma<0, 0>), grid, block, 0, ctx->stream, a);
    return rcv_launch_check(ctx);
}

int rcv_filter_i8_fast(rcv_ctx* ctx, const View& s, const View& d, const int8_t* k, int ksize, int shift)
{
    if (ksize < 1 || ksize > 7) return RCV_ERR_UNSUPPORTED;
    int16_t k16[49];
    for (int i = 0; i < ksize * ksize; ++i) k16[i] = k[i];
    return rcv_filter_i16_fast(ctx, s, d, k16, ksize, shift, 0);
}

// fused capture pipeline: packed-or-strided YUYV (2 channels) -> BGR -> integer filter2D, one launch
int rcv_filter_i8_yuyv_fast(rcv_ctx* ctx, const View& s, const View& d, const int8_t* k, int ksize, int shift)
{
    if (ksize < 1 || ksize > 7) return RCV_ERR_UNSUPPORTED;
    int16_t k16[49];
    for (int i = 0; i < ksize * ksize; ++i) k16[i] = k[i];
    return rcv_filter_i16_fast(ctx, s, d, k16, ksize, shift, 1);
}
