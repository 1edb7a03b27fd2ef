// rcv_filter_rows_mfma.hip -- filter2D with integer (i8) weights, ksize 3/5/7, u8 BGR -> u8 BGR, as a ROW-STREAMING stencil
// on the i8 matrix cores whose data operands never leave the register file: no LDS, no barrier, nothing shared between waves.
// (Round 2's north-star kernel; the round-1 strip kernel, rcv_filter7_mfma.hip, keeps the shapes this one does not take.)
//
// Why MFMA at all: 147 MACs per pixel against 6 bytes (DESIGN_HISTORY.md 4.1) -- the vector ALU cannot keep the op HBM-bound.
//
// The matrix product.  Per colour plane a 16-pixel output tile needs a 22-pixel source window, so the 64-deep data operand of
// v_mfma_i32_16x16x64_i8 holds the 32-pixel windows of TWO source rows:
//
//   D[m][n] += sum_k A[m][k] * B[k][n]
//     n : one of the 16 WINDOWS of the wave's strip: window n = pixels [16n - 4, 16n + 28) of the strip, output tile = [16n, 16n + 16)
//     k : (row half h = k >> 5, window pixel j = k & 31): plane[row 2i + h][16n - 4 + j]
//     m : output pixel 16n + m
//     A[m][32h + j] = K[ky(h)][j - 4 - m + r]   (banded, constant; lane l holds A[l & 15][16 (l >> 4) .. +15])
//   lane (n, q = l >> 4) holds B[16q .. 16q + 15][n]: 16 consecutive pixels of ONE plane of row 2i + (q >> 1), chunk (q & 1) of
//   the window -- i.e. a function of 48 consecutive SOURCE BYTES.  The lane loads those 48 bytes itself (three dword-aligned
//   dwordx4; the half-waves q < 2 / q >= 2 fetch the two rows of a pair with one instruction), de-interleaves them with 24
//   v_perm into the B, G and R operands and xors in the sign bit (u8 enters the signed MFMA as p ^ 0x80; the accumulator
//   starts at 128 * sum(K) + round).  Every source byte is loaded by two lanes (window overlap): the second read is an L1 hit.
//
// Traversal.  A wave (= a 64-thread workgroup) owns a strip of 256 pixels = 768 bytes = six whole 128-byte lines per row and a
// BAND of rows, and walks down it two rows at a time.  Source rows live in a register ring of row PAIRS (2i, 2i + 1); a step
// needs NP = (ksize + 1) / 2 pairs and produces two output rows from them:
//     even row 2u     : pairs u .. u + NP - 1 with kernel rows [K0 K1] [K2 K3] [K4 K5] [K6  0]
//     odd  row 2u + 1 : the same pairs with                    [ 0 K0] [K1 K2] [K3 K4] [K5 K6]
// = 2 * NP * 3 MFMAs per two rows of 256 pixels (4/7 of one-kernel-row-per-MFMA), accumulators chained through the C operand.
// PP further pairs are in flight (requested PP steps ahead into the same ring; the loop is unrolled NP + PP times so that ring
// slots are static registers).  The three planes' accumulators interleave directly into the lane's 12 output bytes (4 pixels x
// BGR): v_ashr_pk_u8_i32 shift + saturate + pack, one non-temporal dwordx3 store per lane and row -- a wave instruction
// writes 768 contiguous bytes.
// BORDER_REFLECT_101: rows by reading the mirrored source row (scalar index math); columns in the first / last strip by
// reading the chunk shifted into the row and repairing the planar registers (EDGE instantiation of the row loop, wave-uniform).
//
// Work split.  The batch's n * rows frame-rows are cut into equal bands; (band, strip) waves are dispatched so that each XCD
// gets a contiguous run of bands and the strips of a band -- which share the 128-byte lines at their seams -- are neighbours
// on one L2.  Occupancy: with 2 row pairs in flight the kernel needs 162 VGPRs (12 waves per CU), with 3 pairs 174 (8 waves per
// CU).  Measured on 64 4K frames: 2 pairs at 12 / 10 / 8 waves per CU (capped with an untouched dynamic-LDS request) 0.581 /
// 0.563 / 0.573 ms; 3 pairs (8 waves) 0.557-0.566; 6 / 4 waves 0.583 / 0.62 -- more streams in flight do not help on this
// memory system, deeper streams do (the memory-only variant of the kernel moves the same way).  Default: 3 pairs.
//
// What was tried on the way (DESIGN_HISTORY.md 4.1 has the numbers): byte-space operands straight from global memory with one kernel
// row per MFMA (taps every 3rd byte; 7 MFMAs per tile, no VALU de-interleave) -- correct, but 1.75x the matrix work pulls the
// clock from 2.25 to 1.95 GHz and the per-CU memory path with it; the same with v_smfmac_i32_16x16x128_i8 (the byte-space band
// IS 2:4 sparse; layout probed with tools/probe_smfmac.hip) -- half the instructions, each twice as long.
#include "rcv_internal.h"
#include "rcv_kernels.h"
#include "rcv_device_utils.h"
#include <string.h>

// This file is compiled TWICE: into librustcv_hip.so (the product: the plain instantiations and the launch plan below, nothing else)
// and, with -DRCV_ROWS_BENCH, into librustcv_hip_bench.so, where every plan parameter is an argument of rcv__filter_rows_bench
// (RowsTune) and the measurement instantiations exist -- memory-only variants, ablations, the per-wave timeline.  bench.py's
// memory_only leg and tools/ablate_*.py call that entry; the product library carries no experiment and reads no tuning knob here
// beyond the four dispatch overrides its tests need (RCV_F7_ROWS, RCV_F7_DUAL_FULL, RCV_FR_CHAIN, RCV_FR_CHAIN_ROWS).
struct RowsTune {
    int f7_rows = -1;     // 1 every eligible shape, 0 never, -1 by size
    int dual_full = 0;    // large weights: K = 4Q + R even where the centre split applies
    int chain = -1;       // chained-band kernel: 0 never, 1 every eligible launch, -1 launches that fill the GPU
    int chain_rows = 0;   // rows per chained band (0 = 32)
    // ---- measurement builds only (the product passes the defaults) ----
    int dbg = 0;          // instantiation: 4 memory-only, 1 / 2 / ... ablations (launch_rows_dbg), 24 wave timeline
    int wpc = 0;          // waves per CU the bands are planned for / (chain) resident per CU
    int rounds = 0;       // bands per wave slot (0 = 8)
    int pp = 0;           // row pairs in flight (0 = 3)
    int order = 0;        // 1: bands dealt round-robin to the XCDs
    int bpf = 0;          // bands per frame
    int band_rows = 0;    // rows per band on launches of a few frames
    int taper = -1;       // 0 equal bands, -1 / 1 tapered tail, n > 1: n % of a round
    int wpb = 0;          // waves per workgroup (2 / 4 / 8)
    int edge_pct = 0;     // (chain) weight of an edge strip against an interior one, % (0 = 115)
    int var = 0;          // (chain, 7x7) code variant under test, a bit mask: see CV_* below
    int lane = 0;         // lane of ticket counters: 0, or 1 for the second half of a split call (which runs on the half stream)
    bool must_chain = false;   // RCV_ERR_UNSUPPORTED (nothing enqueued) unless the launch takes the chained kernel
    void* trace = nullptr;   // device buffer of the per-wave timeline
};

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v3i __attribute__((ext_vector_type(3)));

struct FRArgs {
    const uint8_t* src;
    uint8_t* dst;
    const uint4* wtab;   // 2 x NP tables x 64 lanes x 16 B (A operands), device memory
    uint8_t* dump;       // 64 x 16 B scratch for masked-off lanes (EDGE path only)
    size_t sstep, dstep, sfs, dfs;
    int rows, cols;
    int nstrips, nframes;
    int nbands, bands_per_xcd;   // the n * rows frame-rows of the batch are cut into nbands equal bands; a wave = (band, strip)
    int order;                   // 0: a contiguous eighth of the bands per XCD (default); 1: bands dealt round-robin to the XCDs
    int wpb;                     // waves per workgroup (1, 2, 4 or 8): independent waves, neighbouring strips of a band on one CU
    int shift, acc_init;
    int dual_shift;              // DMASK != 0: result = acc + (acc2 << dual_shift), the second tables follow the first 2 x NP
    uint8_t *gdx, *gdy;          // SOB: the i16 gradient planes (one channel), row step / frame stride in bytes
    // tn > 0: tapered bands (order 0, n % 8 == 0).  Every XCD's eighth of the frame-rows is cut into tn runs of equal bands, the
    // last runs into shorter ones, so that the waves of the launch's last round finish together (see the host code).  Rows relative
    // to the XCD's first row.
    int tn, tnb[4];
    long long tstart[4], tlen[4];
    unsigned long long* trace;   // (DBG & 1024, profiling builds) per-wave {start, end} of the 100 MHz counter
    size_t gstep, gfs;
    // chained bands (k_filter_rows_chain, round 4): every frame is cut into bpf bands; the batch's n * bpf bands (frame-major) are dealt
    // to the XCDs in eight contiguous runs -- XCD x owns bands [gb0[x], gb0[x] + nbx[x]) (round 5: any frame count; a frame may be shared
    // by two XCDs) -- and its waves draw (band, strip) items from that run
    int bpf;
    unsigned cbands;             // n * bpf: XCD x owns bands [cbands * x / 8, cbands * (x + 1) / 8)
    unsigned bmul;               // ceil(rows * 2^20 / bpf): band j of a frame = rows [(j * bmul) >> 20, ((j + 1) * bmul) >> 20)
    int n_edge, edge_waves;          // strips whose windows stick out of the row (the first + the last one or two); waves per XCD that start on them
    unsigned long long inv_edge, inv_int, inv_bpf;   // ceil(2^32 / n_edge), ceil(2^32 / (nstrips - n_edge)), ceil(2^32 / bpf): exact quotients for a launch's item counts
    unsigned long long* tickets;     // this launch's SET of 16 counters (XCD x {interior, edge}), 16 x 8 bytes apart (one 128-byte line each), all zero at its start
    unsigned long long* tickets_next;   // the set of the launch after the next one: every wave zeroes its own XCD's two counters of it
    // tapered tail: the last tp2 bands of every XCD's run are drawn as quarter-height items, the tp1 bands before them as half-height
    // ones, so that the waves of a launch's last round finish close together (fr_queue_items has the item count)
    unsigned tp1, tp2;
    // completion check (see k_chain_check): the counter set of the context's PREVIOUS chained launch and that launch's plan; the fault word
    // lives in pinned host memory and is only ever written when a queue was left short
    const unsigned long long* prev_tickets;
    unsigned prev_cbands, prev_kint, prev_kedge, prev_tp1, prev_tp2;
    unsigned* fault;
    int drop_xcd;                    // (fault injection, RCV_FR_CHAIN_DROP_XCD; -1 = none) waves that run on this XCD leave at once
};

// items of one queue (interior or edge strips: `kstrips` of them) of XCD x: its run of bands, the tapered ones counted 2 / 4 times
__host__ __device__ __forceinline__ unsigned fr_queue_items(unsigned cbands, unsigned x, unsigned kstrips, unsigned tp1, unsigned tp2)
{
    const unsigned gb0 = (unsigned)(((unsigned long long)cbands * x) >> 3), gb1 = (unsigned)(((unsigned long long)cbands * (x + 1)) >> 3);
    const unsigned nb = gb1 - gb0, t2 = tp2 < nb ? tp2 : nb, t1 = tp1 < nb - t2 ? tp1 : nb - t2;
    return (nb + t1 + 3 * t2) * kstrips;
}

// Did the chained launch that drew from `set` finish its lists?  Every queue's counter ends at items + (waves that found it empty); a
// counter below its item count means that nobody drew the rest -- an XCD that received no waves (a CU mask, a partitioned device that
// still reports 256 CUs, a dispatcher that does not place block b on XCD b % 8).  16 lanes, one counter each.
__device__ __forceinline__ void fr_check_set(const unsigned long long* set, unsigned cbands, unsigned kint, unsigned kedge, unsigned tp1, unsigned tp2, unsigned* fault,
                                             int lane)
{
    if (lane < 16) {
        const unsigned need = fr_queue_items(cbands, (unsigned)lane >> 1, (lane & 1) ? kedge : kint, tp1, tp2);
        const unsigned long long got = __builtin_nontemporal_load(set + 16 * lane);
        if (got < need) __hip_atomic_store(fault, 1u + (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
#ifndef RCV_ROWS_BENCH
__global__ void k_chain_check(const unsigned long long* set, unsigned cbands, unsigned kint, unsigned kedge, unsigned tp1, unsigned tp2, unsigned* fault)
{
    fr_check_set(set, cbands, kint, kedge, tp1, tp2, fault, (int)threadIdx.x);
}
#endif

// packed i16 arithmetic on two pixels (the Sobel stage of the SOB instantiation; same forms as rcv_harris_fused.hip)
typedef short fr_s2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t fr_pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(fr_s2v, a) - __builtin_bit_cast(fr_s2v, b)); }
__device__ __forceinline__ uint32_t fr_pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(fr_s2v, a) + __builtin_bit_cast(fr_s2v, b)); }
__device__ __forceinline__ uint32_t fr_pk_add2x(uint32_t a, uint32_t b)   // a + 2 * b on both halves
{
    uint32_t d;
    asm("v_pk_mad_i16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(b), "v"(a));
    return d;
}

struct U3w { uint32_t a, b, c; };

// swap the odd 16-lane rows of x with the even rows of y / the upper 32 lanes of x with the lower 32 of y
__device__ __forceinline__ void sw16(uint32_t& x, uint32_t& y)
{
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}
__device__ __forceinline__ void sw32(uint32_t& x, uint32_t& y)
{
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}

// 4 interleaved BGR pixels (3 dwords) -> planar B, G, R dwords
__device__ __forceinline__ void deint4w(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t& pb, uint32_t& pg, uint32_t& pr)
{
    uint32_t t;
    t = __builtin_amdgcn_perm(d1, d0, 0x00060300u);  // b0(d0.0) b1(d0.3) b2(d1.2) x
    pb = __builtin_amdgcn_perm(d2, t, 0x05020100u);  // + b3(d2.1)
    t = __builtin_amdgcn_perm(d1, d0, 0x00070401u);  // g0(d0.1) g1(d1.0) g2(d1.3) x
    pg = __builtin_amdgcn_perm(d2, t, 0x06020100u);  // + g3(d2.2)
    t = __builtin_amdgcn_perm(d1, d0, 0x00000502u);  // r0(d0.2) r1(d1.1) x x
    pr = __builtin_amdgcn_perm(d2, t, 0x07040100u);  // + r2(d2.0) r3(d2.3)
}

// DBG (measurement build of this file, -DRCV_ROWS_BENCH): 1 skip stores, 2 skip loads, 4 skip MFMAs, 8 plain instead of non-temporal stores
// DMASK: weights beyond the i8 range are split K = M + (S << dual_shift) with M, S inside i8 (integer GaussianBlur 7x7: taps up to
// 324 = K1 + 2 * T2 with T2 confined to the centre rows; any |w| <= 511 as 4Q + R).  The same data operand feeds a second MFMA
// with the S table into a second accumulator set; bit (parity * NP + p) of DMASK says which row pairs have a non-zero S table
// (compile time, so that the accumulator chains stay static): matrix work +5/8 for the Gaussian, memory traffic unchanged.
// SRC = 1: the source is packed YUYV (2 B/px): the lane's 16 pixels are 8 macropixels = 32 consecutive bytes, converted with the
// reference's BT.601 integer formula (rustcv/src/videoio/mod.rs:356-363) straight into the three planar operands -- the capture
// chain YUYV -> BGR -> filter2D in one launch, 5 instead of 11 algorithmic bytes per pixel and no BGR intermediate.
// SRC = 2: ONE-channel (gray) source and destination.  A strip is 768 PIXELS = three neighbouring blocks of 256; what the BGR
// kernel calls the three planes of a window are the same window in the three blocks, loaded as one dwordx4 each -- no
// de-interleave at all.  The lane's four output pixels of the three blocks are three dwords 256 bytes apart: a 4 x 4 dword
// transpose across the four 16-lane rows (2 v_permlane16_swap + 2 v_permlane32_swap) turns them into 16 consecutive bytes per lane.
// SRC = 3: BGR with rows of ANY alignment and ANY width >= 16 (an odd width of a packed image: step = cols * 3).  A lane's 48
// source bytes are fetched as the 13 ALIGNED dwords that contain them and shifted into place with v_alignbyte (the shift is the
// row's own misalignment plus the chunk's: a per-lane value per row pair); the chunk that holds pixel `cols` is repaired at byte
// granularity (its rv = 0..15 valid pixels moved to the front, three mirrored pixels behind them); stores are unaligned 12-byte
// stores, the row's last 1-3 pixels byte by byte.
// SOB = 1 (BGR source, SRC = 0): filter2D -> BGR2GRAY -> Sobel in ONE launch; the filtered image never exists.  The row that
// `finish` would store goes through the gray formula (weights times 4: the value lands in byte 2 of its dword, as in
// rcv_harris_fused.hip) and a Sobel stage in packed i16 that keeps the horizontal parts of the two previous filtered rows in
// registers: filtered row y completes gradient row y - 1.  The horizontal neighbours of a lane's four pixels live in lanes
// l -+ 16 (or l +- 47 across windows): two ds_bpermute per row.
// Layout (round 6; rounds 3-5 laid single waves 240 pixels apart and stored 240 pixels each -- 480-byte row pieces whose seams fall inside
// the 128-byte lines of the i16 planes: the two halves of such a line are written by two waves at unrelated times, and those stores, not
// the arithmetic, were 40 % of the launch: profiles/r06_sob_store_pattern.txt).  A WORKGROUP of four waves owns a group of 960 output pixels
// = 15 whole lines of each gradient plane and walks a band of rows in step (one s_barrier per row pair).  Its 64 windows are the 60 regular
// ones of the group (waves 0-2: 16 each, wave 3: 12) plus two SEAM windows in wave 3's spare slots -- the 16 pixels left of the group and
// the 16 right of it, of which only the nearest pixel is used: the horizontal neighbour of the group's first / last pixel.  (A lane loads
// its window's bytes itself, so a wave's windows need not be neighbours.)  The gray values that cross a wave seam -- 2 per wave and row --
// go through LDS.  Matrix work 64 / 60 of the output as before, but every store is a whole line, written once, non-temporally.
// A band of gradient rows [oys, oye) runs the filter over rows oys - 1 .. oye (clamped to the image); at the image's top / bottom the missing
// row is the mirror image (gradient row 0 is formed from rows 1, 0, 1; row rows - 1 after the loop from rows - 2, rows - 1, rows - 2); the
// row's first pixel takes pixel -1 := pixel 1, the lane whose last pixel is the row's last pixel cols := cols - 2, from itself.
template <int KS, int PP, bool EDGE, int DBG, int DMASK = 0, int SRC = 0, int SOB = 0>
__device__ __forceinline__ void fr_segment(const FRArgs& a, const int lane, const int X, const int ys, const int ye, const uint8_t* sframe,
                                           uint8_t* dframe, const int oys = 0, const int oye = 0, uint8_t* dxf = nullptr, uint8_t* dyf = nullptr,
                                           const int tw = 0, uint32_t* const seam = nullptr)
{
    constexpr int RAD = KS / 2, NP = (KS + 1) / 2;
    constexpr int RP = NP + PP;   // ring of row pairs: the NP-pair window + PP pairs requested ahead
    const int n = lane & 15, q = lane >> 4, h = q >> 1, c = q & 1;
    constexpr bool GRAY = SRC == 2, RAGB = SRC == 3;
    constexpr int SB = GRAY ? 1 : (SRC == 1 ? 2 : 3), CB = 16 * SB;   // source bytes per pixel / per 16-pixel chunk
    const unsigned fmis = RAGB ? (unsigned)((uintptr_t)sframe & 3) : 0u;   // (RAGB) the frame base aligned down: offsets stay non-negative
    const uint8_t* const sfa = sframe - fmis;
    const int rb = a.cols * (GRAY ? 1 : 3), rbs = a.cols * SB;        // destination / source row bytes

    v4i A[2][NP];
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const uint4 w = a.wtab[(par * NP + p) * 64 + lane];
            A[par][p] = v4i{(int)w.x, (int)w.y, (int)w.z, (int)w.w};
        }
    v4i A2[2][NP];
    if constexpr (DMASK != 0) {
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if ((DMASK >> (par * NP + p)) & 1) {
                    const uint4 w = a.wtab[(2 * NP + par * NP + p) * 64 + lane];
                    A2[par][p] = v4i{(int)w.x, (int)w.y, (int)w.z, (int)w.w};
                }
    }

    // the lane's 48 source bytes per pair: pixels [16n - 4 + 16c, +16) of row 2i + h.  Chunks that stick out of the row are read
    // shifted into it and repaired after the de-interleave (EDGE); without EDGE the clamp is a no-op.
    // (gray: X counts pixels = bytes, and the chunk of block pl starts 256 * pl further)
    // SOB: X = 3 * (first pixel of wave tw's 256 pixels of its group); the lane's window begins wpx pixels from there: 16 n, except the two
    // seam windows of wave 3 (n = 12: the 16 pixels left of the group; n = 13 .. 15: the 16 right of it)
    const int wpx = (SOB && tw == 3 && n >= 12) ? (n == 12 ? -784 : 192) : 16 * n;
    const int cb = SOB ? X + 3 * wpx - 12 + 48 * c : (GRAY ? X : (X / 3) * SB) + CB * n - 4 * SB + CB * c;
    const unsigned cbo = (unsigned)min(max(cb, 0), rbs - CB);
    // The chunk that holds pixel `cols` (the first one past the row) has `rv` valid pixels in front of it: 4 when the width is a
    // multiple of 16; BGR also takes widths that are a multiple of 4 (a 1080-pixel portrait frame): rv = 0, 8, 12 then (the same in
    // every such lane of the launch: a scalar)
    const int rv = (SRC == 0 || RAGB) ? ((a.cols & 15) + 4) & 15 : 4;   // (RAGB: any value 0..15; every strip starts at a multiple of 16 pixels)
    const bool fl = EDGE && cb < 0, fr = EDGE && cb == rbs - rv * SB;
    // (RAGB, rv = 14 / 15: the third / second and third mirrored pixel are the first pixels of the NEXT chunk)
    const bool frn = EDGE && RAGB && rv >= 14 && cb == rbs - rv * SB + CB;
    unsigned cbo1 = 0, cbo2 = 0;
    bool fr1 = false, fr2 = false;
    if constexpr (GRAY) {
        cbo1 = (unsigned)min(cb + 256, rbs - CB);
        cbo2 = (unsigned)min(cb + 512, rbs - CB);
        fr1 = EDGE && cb + 256 == rbs - 4;
        fr2 = EDGE && cb + 512 == rbs - 4;
    }
    // the lane's output bytes: BGR: 12 bytes = pixels 16n + 4q .. +3; gray (after the transpose): 16 bytes = window n of block q
    const int so = GRAY ? X + 256 * min(q, 2) + 16 * n : ((DBG & 16) ? X + 12 * lane : X + 48 * n + 12 * q);
    const int ordl = ((lane >> 2) + 16 * (lane & 3)) << 2;   // (DBG & 16) lane-ordered stores: lane l takes the 12 bytes of lane (n = l >> 2, q = l & 3)
    uint8_t* const dumpp = a.dump + lane * 16;

    // accumulator start value as a resident register quad (opaque to the compiler, which would rebuild it before every chain)
    v4i initv = v4i{a.acc_init, a.acc_init, a.acc_init, a.acc_init};
    asm volatile("" : "+v"(initv));

    v4i W[RP][3];   // raw source bytes until prepare() turns them into the B, G, R operands
    int Wx[RP], Wm[RP];   // RAGB: the 13th aligned dword of the lane's run and the byte shift of the run inside the 13
    auto request = [&](int pr, int slot) {
        v4i(&dst)[3] = W[slot];
        // rows 2*pr (lanes h = 0) and 2*pr + 1 (lanes h = 1) of the segment's window, mirrored at the image border (scalar math);
        // rows past the window re-read its last row (cache hits, never used)
        const int XR = (DBG & 128) ? 0 : RAD;   // (experiment: no halo rows)
        const int y0 = min(ys - XR + 2 * pr, ye - 1 + XR), y1 = min(ys - XR + 2 * pr + 1, ye - 1 + XR);
        const int s0 = y0 < 0 ? -y0 : (y0 >= a.rows ? 2 * a.rows - 2 - y0 : y0), s1 = y1 < 0 ? -y1 : (y1 >= a.rows ? 2 * a.rows - 2 - y1 : y1);
        const unsigned o0 = (unsigned)s0 * (unsigned)a.sstep, o1 = (unsigned)s1 * (unsigned)a.sstep;   // < 2^32 (host check)
        const unsigned off = (h ? o1 : o0) + cbo;
        if (DBG & 2) {
            dst[0] = dst[1] = dst[2] = v4i{(int)off, (int)cbo, s0, lane};
            return;
        }
        if constexpr (GRAY) {
            const unsigned ro = h ? o1 : o0;
            dst[0] = *(const v4i*)(sframe + ro + cbo);
            dst[1] = *(const v4i*)(sframe + ro + cbo1);
            dst[2] = *(const v4i*)(sframe + ro + cbo2);
            return;
        }
        if constexpr ((DBG & 32) != 0) {   // (experiment: every source byte loaded by exactly one lane: 24 bytes per lane)
            const unsigned xo = (h ? o1 : o0) + (unsigned)(X + 48 * n + 24 * c);
            typedef int v2i_ __attribute__((ext_vector_type(2)));
            v4i t0;
            v2i_ t1;
            if (DBG & 64) {
                t0 = __builtin_nontemporal_load((const v4i*)(sframe + xo));
                t1 = __builtin_nontemporal_load((const v2i_*)(sframe + xo + 16));
            } else {
                t0 = *(const v4i*)(sframe + xo);
                t1 = *(const v2i_*)(sframe + xo + 16);
            }
            dst[0] = t0;
            dst[1] = v4i{t1.x, t1.y, t0.x, t0.y};
            dst[2] = t0;
            return;
        }
        if constexpr (RAGB) {
            const unsigned offp = off + fmis, ao = offp & ~3u, mis = offp & 3u;
            dst[0] = *(const v4i*)(sfa + ao);
            dst[1] = *(const v4i*)(sfa + ao + 16);
            dst[2] = *(const v4i*)(sfa + ao + 32);
            Wx[slot] = *(const int*)(sfa + ao + (mis ? 48u : 44u));   // (an aligned run needs no 13th dword: never read past it)
            Wm[slot] = (int)mis;
            return;
        }
        dst[0] = *(const v4i*)(sframe + off);
        dst[1] = *(const v4i*)(sframe + off + 16);
        if constexpr (SRC == 0) dst[2] = *(const v4i*)(sframe + off + 32);
    };
    auto prepare = [&](int slot) {
        v4i(&w)[3] = W[slot];
        uint32_t pb[4], pg[4], prr[4];
        if constexpr (GRAY) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pb[i] = (uint32_t)w[0][i];
                pg[i] = (uint32_t)w[1][i];
                prr[i] = (uint32_t)w[2][i];
            }
        } else if constexpr (SRC == 1) {
            // 8 macropixels [Y0 U Y1 V] -> the six pre-shift BT.601 sums of their two pixels; >> 8, saturate and pack four pixels
            // of a plane per dword (v_ashr_pk_u8_i32)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int sb[4], sg[4], sr[4];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const uint32_t mp = (uint32_t)w[i >> 1][2 * (i & 1) + m];
                    const int y0 = (int)(mp & 0xff), u = (int)((mp >> 8) & 0xff) - 128, y1 = (int)((mp >> 16) & 0xff), v = (int)(mp >> 24) - 128;
                    const int c0 = 298 * (y0 - 16) + 128, c1 = 298 * (y1 - 16) + 128;
                    const int db = 516 * u, dg = -100 * u - 208 * v, dr = 409 * v;
                    sb[2 * m] = c0 + db; sg[2 * m] = c0 + dg; sr[2 * m] = c0 + dr;
                    sb[2 * m + 1] = c1 + db; sg[2 * m + 1] = c1 + dg; sr[2 * m + 1] = c1 + dr;
                }
                pb[i] = rcv_ashr_sat_pk4(sb[0], sb[1], sb[2], sb[3], 8);
                pg[i] = rcv_ashr_sat_pk4(sg[0], sg[1], sg[2], sg[3], 8);
                prr[i] = rcv_ashr_sat_pk4(sr[0], sr[1], sr[2], sr[3], 8);
            }
        } else {
            uint32_t r[13] = {(uint32_t)w[0][0], (uint32_t)w[0][1], (uint32_t)w[0][2], (uint32_t)w[0][3], (uint32_t)w[1][0], (uint32_t)w[1][1],
                              (uint32_t)w[1][2], (uint32_t)w[1][3], (uint32_t)w[2][0], (uint32_t)w[2][1], (uint32_t)w[2][2], (uint32_t)w[2][3], 0u};
            if constexpr (RAGB) {
                r[12] = (uint32_t)Wx[slot];
#pragma unroll
                for (int i = 0; i < 12; ++i) r[i] = __builtin_amdgcn_alignbyte(r[i + 1], r[i], (uint32_t)Wm[slot]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) deint4w(r[3 * i], r[3 * i + 1], r[3 * i + 2], pb[i], pg[i], prr[i]);
        }
        if (EDGE) {
            // left border: the lane read pixels 0..15 instead of -4..11: shift by one dword; pixels -3..-1 mirror 3, 2, 1
            if (fl) {
#pragma unroll
                for (int pl = 0; pl < (GRAY ? 1 : 3); ++pl) {
                    uint32_t* pp = pl == 0 ? pb : (pl == 1 ? pg : prr);
                    pp[3] = pp[2];
                    pp[2] = pp[1];
                    pp[1] = pp[0];
                    pp[0] = __builtin_amdgcn_perm(pp[0], pp[0], 0x01020300u);   // [x, px3, px2, px1]
                }
            }
            if (frn) {   // the lane read pixels cols-16..cols-1 (clamped): pixel 0 (and 1) of its chunk := cols-4 (cols-3, cols-4)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    uint32_t* pp = pl == 0 ? pb : (pl == 1 ? pg : prr);
                    pp[0] = rv == 14 ? pp[3] : __builtin_amdgcn_perm(pp[3], pp[3], 0x00000001u);
                }
            }
            // right border: the lane read pixels cols-16..cols-1 instead of cols-rv..cols-rv+15: its rv valid pixels move down to
            // the front, pixels cols..cols+2 mirror cols-2..cols-4
            if (GRAY ? (fr || fr1 || fr2) : fr) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    if (GRAY && !(pl == 0 ? fr : (pl == 1 ? fr1 : fr2))) continue;
                    uint32_t* pp = pl == 0 ? pb : (pl == 1 ? pg : prr);
                    const uint32_t mir = __builtin_amdgcn_perm(pp[3], pp[3], 0x00000102u);   // [cols-2, cols-3, cols-4, x]
                    if (RAGB && (rv & 3) != 0) {
                        // rv is any value: the plane's 16 bytes move down by 16 - rv bytes (whole dwords + a byte shift), then the
                        // three mirrored bytes go in at byte rv (all positions are the same in every lane: scalars)
                        const int sh16 = 16 - rv, sft = sh16 >> 2, a4 = rv >> 2;
                        const uint32_t bs = (uint32_t)(sh16 & 3), b8 = 8u * (uint32_t)(rv & 3);
                        const uint32_t L0 = pp[0], L1 = pp[1], L2 = pp[2], L3 = pp[3];
                        auto pick = [&](int j) -> uint32_t { return j == 0 ? L0 : (j == 1 ? L1 : (j == 2 ? L2 : (j == 3 ? L3 : 0u))); };
                        uint32_t T[5];
#pragma unroll
                        for (int i = 0; i < 4; ++i) T[i] = __builtin_amdgcn_alignbyte(pick(i + sft + 1), pick(i + sft), bs);
                        T[4] = 0u;
                        const uint32_t m0 = 0x00ffffffu << b8, m1 = 0x00ffffffu >> (32u - b8);   // (rv & 3 != 0: b8 = 8, 16, 24)
                        const uint32_t v0 = mir << b8, v1 = mir >> (32u - b8);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            uint32_t t = T[i];
                            t = i == a4 ? ((t & ~m0) | (v0 & m0)) : t;
                            t = i == a4 + 1 ? ((t & ~m1) | (v1 & m1)) : t;
                            pp[i] = t;
                        }
                    } else if (SRC == 1 || GRAY || rv == 4) {
                        pp[0] = pp[3];
                        pp[1] = mir;
                    } else if (rv == 8) {
                        pp[0] = pp[2];
                        pp[1] = pp[3];
                        pp[2] = mir;
                    } else if (rv == 12) {
                        pp[0] = pp[1];
                        pp[1] = pp[2];
                        pp[2] = pp[3];
                        pp[3] = mir;
                    } else {   // rv == 0: the chunk starts at pixel cols
                        pp[0] = mir;
                    }
                }
            }
        }
        w[0] = v4i{(int)(pb[0] ^ 0x80808080u), (int)(pb[1] ^ 0x80808080u), (int)(pb[2] ^ 0x80808080u), (int)(pb[3] ^ 0x80808080u)};
        w[1] = v4i{(int)(pg[0] ^ 0x80808080u), (int)(pg[1] ^ 0x80808080u), (int)(pg[2] ^ 0x80808080u), (int)(pg[3] ^ 0x80808080u)};
        w[2] = v4i{(int)(prr[0] ^ 0x80808080u), (int)(prr[1] ^ 0x80808080u), (int)(prr[2] ^ 0x80808080u), (int)(prr[3] ^ 0x80808080u)};
    };

#pragma unroll
    for (int i = 0; i < RP - 1; ++i) request(i, i);
#pragma unroll
    for (int i = 0; i < NP - 1; ++i) prepare(i);

    const int nrows = ye - ys;
    // SOB: Sobel state (horizontal parts of filtered rows y-2, y-1 as packed i16 pairs) and the lane's place in the tile
    uint32_t sh1a[2] = {0u, 0u}, sh1b[2] = {0u, 0u}, sh2a[2] = {0u, 0u}, sh2b[2] = {0u, 0u};
    const int tpx = (SOB ? wpx : 16 * n) + 4 * q, gpx = X / 3 + tpx;         // the lane's first output pixel: in the wave's 256 / in the row
    const bool g_first = gpx == 0 && lane == 0, g_last = gpx + 4 == a.cols;  // the lane holds the row's first / last pixel
    const int addrL = 4 * (q > 0 ? lane - 16 : (n > 0 ? lane + 47 : lane)), addrR = 4 * (q < 3 ? lane + 16 : (n < 15 ? lane - 47 : lane));
    // gray values that cross a wave seam (dword slots per row: 0-3 the first pixel of wave 0-3, 4-6 the last pixel of wave 0-2, 7 the pixel
    // left of the group, 8 the pixel right of it): what this lane writes (its first pixel: sel 0, its last: sel 1) and what it reads instead
    // of the bpermute's value
    int wr_slot = -1, wr_sel = 0, rdL = -1, rdR = -1;
    if (SOB) {
        if (lane == 0) wr_slot = tw, rdL = tw > 0 ? 3 + tw : 7;
        if (lane == 63 && tw < 3) wr_slot = 4 + tw, wr_sel = 1, rdR = tw + 1;
        if (tw == 3 && lane == 60) wr_slot = 7, wr_sel = 1;   // (n = 12, q = 3): the last pixel of the left seam window
        if (tw == 3 && lane == 13) wr_slot = 8;               // (n = 13, q = 0): the first pixel of the right seam window
        if (tw == 3 && lane == 59) rdR = 8;                   // (n = 11, q = 3): the group's last pixel
    }
    // stores: lanes (n, q) and (n, q + 1), q even, hold eight consecutive pixels; after one v_permlane16_swap per dword the even
    // lane has both lanes' dx and the odd lane both lanes' dy: ONE 16-byte store per lane and row.  Valid: the regular windows' pixels
    // inside the group's 960 and inside the row; a width that is not a multiple of 8 ends on a half pair.
    const int tp2 = (SOB ? wpx : 16 * n) + 8 * (q >> 1), gp2 = X / 3 + tp2;
    const int glim = min(X / 3 - 256 * tw + 960, a.cols);                  // end of the group's output pixels
    const bool pair_ok = SOB && !(tw == 3 && n >= 12);
    const bool g_full = pair_ok && gp2 + 8 <= glim, g_half = pair_ok && gp2 + 8 > glim && gp2 + 4 <= glim;
    uint8_t* const gplane = (q & 1) ? dyf : dxf;
    typedef uint32_t fr_u2 __attribute__((ext_vector_type(2)));
    typedef uint32_t fr_u4 __attribute__((ext_vector_type(4)));
    // gradient row gy from the horizontal parts of rows gy-1 (p1, p2), gy (c1) and gy+1 (n1, n2)
    auto emit = [&](const uint32_t(&p1)[2], const uint32_t(&c1)[2], const uint32_t(&n1)[2], const uint32_t(&p2)[2], const uint32_t(&n2)[2], int gy) {
        uint32_t x0 = fr_pk_add2x(fr_pk_add(p1[0], n1[0]), c1[0]), x1 = fr_pk_add2x(fr_pk_add(p1[1], n1[1]), c1[1]);
        uint32_t y0 = fr_pk_sub(n2[0], p2[0]), y1 = fr_pk_sub(n2[1], p2[1]);
        sw16(x0, y0);   // even q: (x0, x1) own dx, (y0, y1) the dx of lane + 16; odd q: (x0, x1) the dy of lane - 16, (y0, y1) own dy
        sw16(x1, y1);
        uint8_t* const o = gplane + (size_t)gy * a.gstep + 2 * (size_t)gp2;
        // whole 128-byte lines of each plane, written once: non-temporal (the launch never reads its output back)
        if constexpr ((DBG & 1) != 0) {   // (measurement: no gradient stores)
            if (x0 == 0x12345678u && y1 == 0x9abcdef0u) *(fr_u4*)o = fr_u4{x0, x1, y0, y1};
        } else if (g_full) __builtin_nontemporal_store(fr_u4{x0, x1, y0, y1}, (fr_u4*)o);
        else if (g_half) *(fr_u2*)o = fr_u2{x0, x1};
    };
    // SOB, per filtered row: the lane's four gray values (byte 2 of each dword) ...
    auto gray_row = [&](v4i(&acc)[3], uint32_t(&g)[4]) {
        uint32_t oa, ob, oc;
        {
            const int v[12] = {acc[0][0], acc[1][0], acc[2][0], acc[0][1], acc[1][1], acc[2][1], acc[0][2], acc[1][2], acc[2][2], acc[0][3], acc[1][3], acc[2][3]};
            rcv_ashr_sat_pk12_mfma(v, a.shift, oa, ob, oc);
        }
        if constexpr ((DBG & 2048) != 0) {   // (measurement: no gray arithmetic)
            g[0] = oa; g[1] = ob; g[2] = oc; g[3] = oa;
            return;
        }
        const uint32_t px[4] = {oa, __builtin_amdgcn_alignbyte(ob, oa, 3), __builtin_amdgcn_alignbyte(oc, ob, 2), oc >> 8};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t hi8 = __builtin_amdgcn_udot4(px[i], 0x004c961du, 0u, false);      // 4 * (1868, 9617, 4899) = 256 * {29,150,76} + {48,68,140}
            const uint32_t lo8 = __builtin_amdgcn_udot4(px[i], 0x008c4430u, 32768u, false);  // (+ 4 * 8192): gray = bits 16..23
            g[i] = (hi8 << 8) + lo8;
        }
    };
    // ... and, once the gray values of the wave seams are in LDS (row slot rs of this step's buffer), the row's horizontal parts and the gradient
    // row above it
    auto sobel_row = [&](const uint32_t(&g)[4], const uint32_t* sx, int y) {
        uint32_t gl = (uint32_t)__builtin_amdgcn_ds_bpermute(addrL, (int)g[3]), gr = (uint32_t)__builtin_amdgcn_ds_bpermute(addrR, (int)g[0]);
        if (rdL >= 0) gl = sx[rdL];
        if (rdR >= 0) gr = sx[rdR];
        if (g_first) gl = g[1];   // pixel -1 := pixel 1
        if (g_last) gr = g[2];    // pixel cols := pixel cols - 2
        constexpr uint32_t kPair = 0x0c060c02u;   // (byte 2 of the low source, byte 2 of the high source) as two u16
        const uint32_t Pa = __builtin_amdgcn_perm(g[0], gl, kPair), Pb = __builtin_amdgcn_perm(g[2], g[1], kPair), Pc = __builtin_amdgcn_perm(gr, g[3], kPair);
        const uint32_t C0 = __builtin_amdgcn_perm(g[1], g[0], kPair), C1 = __builtin_amdgcn_perm(g[3], g[2], kPair);
        const uint32_t h1[2] = {fr_pk_sub(Pb, Pa), fr_pk_sub(Pc, Pb)};
        const uint32_t h2[2] = {fr_pk_add2x(fr_pk_add(Pa, Pb), C0), fr_pk_add2x(fr_pk_add(Pb, Pc), C1)};
        const int gy = y - 1;
        if (gy >= oys && gy < oye) {   // (scalar conditions)
            if (gy == 0) emit(h1, sh1b, h1, h2, h2, gy);   // row -1 := row 1
            else emit(sh1a, sh1b, h1, sh2a, h2, gy);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            sh1a[j] = sh1b[j];
            sh1b[j] = h1[j];
            sh2a[j] = sh2b[j];
            sh2b[j] = h2[j];
        }
    };
    auto finish = [&](v4i(&acc)[3], const v4i(&acc2)[3], int y) {
        if constexpr (DMASK != 0) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) acc[pl] += acc2[pl] << a.dual_shift;
        }
        if constexpr (GRAY) {
            // lane (q, n): four consecutive pixels of window n in each of the three blocks -> transpose -> window n of block q
            uint32_t D[4];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) D[pl] = 0u;
            {
                const int v[12] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3], acc[2][0], acc[2][1], acc[2][2], acc[2][3]};
                rcv_ashr_sat_pk12_mfma(v, a.shift, D[0], D[1], D[2]);
            }
            D[3] = D[2];
            sw16(D[0], D[1]);
            sw16(D[2], D[3]);
            sw32(D[0], D[2]);
            sw32(D[1], D[3]);
            uint8_t* drow = dframe + (size_t)y * a.dstep;
            if (DBG & 1) {
                if (D[0] == 0x12345678u && D[1] == 0x9abcdef0u) *(uint4*)dumpp = make_uint4(D[0], D[1], D[2], D[3]);
            } else if (EDGE) {
                *(uint4*)(so < rb ? drow + so : dumpp) = make_uint4(D[0], D[1], D[2], D[3]);
            } else {
                __builtin_nontemporal_store(v4i{(int)D[0], (int)D[1], (int)D[2], (int)D[3]}, (v4i*)(drow + so));
            }
            return;
        }
        // lane (q, n) holds pixels 16n + 4q .. +3 of the three planes: 12 interleaved output bytes
        U3w o;
        {
            const int v[12] = {acc[0][0], acc[1][0], acc[2][0], acc[0][1], acc[1][1], acc[2][1], acc[0][2], acc[1][2], acc[2][2], acc[0][3], acc[1][3], acc[2][3]};
            rcv_ashr_sat_pk12_mfma(v, a.shift, o.a, o.b, o.c);
        }
        if (DBG & 16) {
            o.a = (uint32_t)__builtin_amdgcn_ds_bpermute(ordl, (int)o.a);
            o.b = (uint32_t)__builtin_amdgcn_ds_bpermute(ordl, (int)o.b);
            o.c = (uint32_t)__builtin_amdgcn_ds_bpermute(ordl, (int)o.c);
        }
        uint8_t* drow = dframe + (size_t)y * a.dstep;
        if (DBG & 1) {
            if (o.a == 0x12345678u && o.b == 0x9abcdef0u) *(U3w*)dumpp = o;
        } else if (EDGE && RAGB) {
            // any width: a lane's four pixels may end past the row (its last 1-3 pixels byte by byte) or lie past it altogether
            typedef uint32_t v3u1 __attribute__((ext_vector_type(3), aligned(1)));
            if (so + 12 <= rb) *(v3u1*)(drow + so) = v3u1{o.a, o.b, o.c};
            else if (so < rb) {
                const uint32_t ow[3] = {o.a, o.b, o.c};
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    if (so + i < rb) drow[so + i] = (uint8_t)(ow[i >> 2] >> (8 * (i & 3)));
            }
        } else if (EDGE) {
            *(U3w*)(so < rb ? drow + so : dumpp) = o;   // windows past the row end (partial last strip) go to the dump line
        } else if (DBG & 8) {
            *(U3w*)(drow + so) = o;
        } else {
            // non-temporal: this launch never reads its output back, and the source lines that neighbouring strips share
            // stay in L2 (measured -3 % end to end against plain stores)
            __builtin_nontemporal_store(v3i{(int)o.a, (int)o.b, (int)o.c}, (v3i*)(drow + so));
        }
    };
    const int nsteps = (nrows + 1) >> 1;
    if constexpr (SOB != 0) __builtin_amdgcn_s_barrier();   // (the previous segment's last step may have used the seam buffer this one begins with)
    for (int u0 = 0; u0 < nsteps; u0 += RP) {
#pragma unroll
        for (int s = 0; s < RP; ++s) {
            const int u = u0 + s;
            if (u >= nsteps) break;
            request(u + RP - 1, (s + RP - 1) % RP);
            if constexpr ((DBG & 256) != 0) {
                // the kernel's memory pattern alone: the loads of the real kernel, and its stores fed with loaded bytes (no
                // de-interleave, no MFMA, no epilogue) -- the ceiling of THIS access pattern (bench.py: copy_ceiling)
                const v4i& w0 = W[s % RP][0];
                const v4i& w1 = W[s % RP][1];
                const int y = ys + 2 * u;
                if ((DBG & 512) != 0 && (u & 1) != oys) continue;   // (this experiment passes the wave's parity in `oys`, unused without SOB)
                __builtin_nontemporal_store(v3i{w0[0], w0[1], w0[2]}, (v3i*)(dframe + (size_t)y * a.dstep + so));
                if (2 * u + 1 < nrows) __builtin_nontemporal_store(v3i{w1[0], w1[1], w1[2]}, (v3i*)(dframe + (size_t)(y + 1) * a.dstep + so));
                continue;
            }
            prepare((s + NP - 1) % RP);
            v4i acc[2][3], acc2[2][3];
            const v4i zerov = v4i{0, 0, 0, 0};
#pragma unroll
            for (int par = 0; par < 2; ++par)
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        if (DBG & 4) acc[par][pl] = p == 0 ? W[(s + p) % RP][pl] : acc[par][pl] + W[(s + p) % RP][pl];
                        else acc[par][pl] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[par][p], W[(s + p) % RP][pl], p == 0 ? initv : acc[par][pl], 0, 0, 0);
                        if constexpr (DMASK != 0) {
                            const int bits = (DMASK >> (par * NP)) & ((1 << NP) - 1);   // this parity's pairs (constant after unrolling)
                            if ((bits >> p) & 1) {
                                const bool first = (bits & ((1 << p) - 1)) == 0;   // no earlier pair of this parity has an S table
                                acc2[par][pl] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A2[par][p], W[(s + p) % RP][pl], first ? zerov : acc2[par][pl], 0, 0, 0);
                            }
                        }
                    }
            if constexpr (SOB != 0) {
                // both rows' gray values, the wave seams' through LDS (two buffers by step parity: a wave that is a step ahead writes the
                // other one), ONE barrier per step, then the Sobel stage of both rows
                uint32_t gA[4], gB[4];
                gray_row(acc[0], gA);
                gray_row(acc[1], gB);
                uint32_t* const sx = seam + 32 * (u & 1);
                if (wr_slot >= 0) {
                    sx[wr_slot] = wr_sel ? gA[3] : gA[0];
                    sx[16 + wr_slot] = wr_sel ? gB[3] : gB[0];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if constexpr ((DBG & 4096) == 0) __builtin_amdgcn_s_barrier();   // (DBG & 4096, measurement: what keeping the four waves in step costs)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                sobel_row(gA, sx, ys + 2 * u);
                if (2 * u + 1 < nrows) sobel_row(gB, sx + 16, ys + 2 * u + 1);
            } else {
                finish(acc[0], acc2[0], ys + 2 * u);
                if (2 * u + 1 < nrows) finish(acc[1], acc2[1], ys + 2 * u + 1);
            }
        }
    }
    if constexpr (SOB != 0) {
        if (oye == a.rows) emit(sh1a, sh1b, sh1a, sh2a, sh2a, a.rows - 1);   // the image's last row: row `rows` := row rows - 2
    }
}

// ---- chained bands (round 4) ---------------------------------------------------------------------------------------------------
// What the memory system wants (tools/ablate_walk*.py, ablate_bands.py; DESIGN_HISTORY.md 4.1 round 4): the strip-walker copy -- this
// kernel's access pattern without the kernel -- runs 6-10 % faster when the bands are 16-32 rows instead of 103: what one XCD's 256
// waves touch at a time is then a window of ~500 rows (6 MB) instead of a whole frame (25 MB).  The one-band-per-wave kernel cannot
// use short bands: every band costs a wave launch, the weight tables, 2 * RAD halo rows and a pipeline fill (32-row bands: 0.587
// against 0.574 ms while its memory-only variant GAINS 4 %).  Here a wave is PERSISTENT and CHAINS items: it draws (band, strip)
// items from its XCD's ticket counter -- band-major, so the 256 waves of an XCD work on ~17 neighbouring bands, and dynamic, so a
// slow wave simply takes fewer items (a STATIC assignment of bands to persistent waves was measured first: 6-9 % slower than hardware
// dispatch even for the plain walker copy, and the EDGE strips' slower waves set the launch time) -- and walks them as ONE continuous
// stream of row pairs through the register ring.  While the last rows of an item are computed, the first rows of the next one (its
// halo included) are already in flight; tables and lane constants are set up once per wave.  Three cursors run over the same item
// sequence: requests (RP - 1 pairs ahead), operand preparation (NP - 1 ahead), output.  The NP - 1 steps whose window straddles
// two items compute garbage that is never stored.  The strip changes from item to item, so the border repair of the first / last
// strip sits under a wave-uniform branch in `prepare` (VALU only) and every store is exec-masked by `byte offset < row bytes`.
// Tickets (round 5: every launch self-contained).  One 64-bit counter per XCD and queue, 128 bytes apart, in FOUR sets (ctx->kconst +
// RCV_KC_FR_TICKETS): launch i of a context draws from set i % 4, which is all zero when it starts, and every wave of launch i zeroes
// the two counters of ITS XCD in set (i + 2) % 4 -- the set launch i - 2 used, idle since then, next used by launch i + 2 (launches of
// a context run in stream order).  No host-side count of what a launch draws exists any more, so nothing can drift.  A counter is only
// ever touched by waves of one XCD: the XCD is read from the hardware (HW_REG_XCC_ID), not inferred from the block index, so the
// L2-scope atomic is coherent whatever the dispatcher does; that every XCD receives waves (block b -> XCD b % 8) is still needed for
// the launch to finish its lists and holds on an unpartitioned MI355X (the host gates on 256 CUs).
struct FRItem {                // (all scalar)
    int X;                     // byte offset of the strip in a destination row
    int ys, nrows, P;          // rows [ys, ys + nrows) of the frame, P = ceil(nrows / 2) + NP - 1 pairs
    bool interior, done;       // no source row of the item needs mirroring; past the end of the list
    const uint8_t* sf;
    uint8_t* df;
};

// One queue of one XCD: the (band, strip) items of the EDGE strips (the first strip of a row and the last one or two, whose windows
// stick out of the row) or of the interior strips, band-major.  EDGE is a compile-time property of the loop -- the border repair
// and the masked stores exist in the edge loop only -- so a wave runs the loop of its own kind until that queue is empty and then
// helps with the other one (kernel below).  Returns when the queue is empty.
// Code variants of the chained loop (a bit mask; the product instantiates kChainVar, the measurement build also 0 and the traced form):
//   CV_PACK2   the four accumulators of an output dword packed by TWO v_ashr_pk_u8_i32 (the second with op_sel[3]: writes D[31:16]) instead of
//              2 + shift + or: 6 instead of 12 VALU per row (rcv_ashr_sat_pk12_mfma)
//   CV_TRACE   (measurement) every wave records the chip-wide 100 MHz counter at its start and after its last store (FRArgs::trace)
// Round 6, same-process A/B on four boxes (profiles/r06_chain_variants.txt): CV_PACK2 -0.9 ... -1.2 % on three, +0.2 % on one -- with it the
// filter runs level with its own memory-only form.  Built, measured and removed again: twelve more VALU instructions per step +0.2 % (an
// instruction costs next to nothing here); a second, discarded ticket draw per item from a line of the same XCD and queue +0.1 % (a draw costs
// nothing measurable); the next item's ticket requested three steps ahead and waited for at the item's end 0 ... +0.4 % (and the pending
// scalar register pair is invisible to the compiler: not worth the hazard).
constexpr int CV_PACK2 = 1, CV_TRACE = 32;
constexpr int kChainVar = CV_PACK2;   // the product's form
template <int KS, int PP, bool EDGE, int DBG, int DMASK, int VAR>
__device__ __forceinline__ void fr_chain_run(const FRArgs& a, const int lane, const int xcd, const v4i (&A)[2][(KS + 1) / 2], const v4i (&A2)[2][(KS + 1) / 2],
                                             const v4i& initv)
{
    constexpr int RAD = KS / 2, NP = (KS + 1) / 2;
    constexpr int RP = NP + PP;
    const int n = lane & 15, q = lane >> 4, h = q >> 1, c = q & 1;
    const int rb = a.cols * 3;
    const int lane_cb = 48 * n - 12 + 48 * c, lane_so = 48 * n + 12 * q;
    const int rv = ((a.cols & 15) + 4) & 15;
    const unsigned sstep = (unsigned)a.sstep, dstep = (unsigned)a.dstep;   // (in-frame offsets are 32-bit: host check)
    const int kstrips = EDGE ? a.n_edge : a.nstrips - a.n_edge;            // strips of this kind
    // (computed, not looked up: a dynamic index into the by-value argument struct would move the whole struct to scratch memory)
    const unsigned gb0 = (unsigned)(((unsigned long long)a.cbands * (unsigned)xcd) >> 3), gb1 = (unsigned)(((unsigned long long)a.cbands * (unsigned)(xcd + 1)) >> 3);
    const unsigned nb = gb1 - gb0, t2 = min(a.tp2, nb), t1 = min(a.tp1, nb - t2), nbn = nb - t1 - t2;   // bands of the run: ordinary, halved, quartered
    const unsigned n0 = nbn * (unsigned)kstrips, n1 = n0 + 2u * t1 * (unsigned)kstrips;
    const unsigned nitems = n1 + 4u * t2 * (unsigned)kstrips;             // of this XCD and kind (= fr_queue_items)
    if (nitems == 0) return;
    unsigned long long* const tick = a.tickets + 16 * (2 * xcd + (EDGE ? 1 : 0));
    const unsigned long long inv_k = EDGE ? a.inv_edge : a.inv_int;

    // A ticket is drawn with a SCALAR atomic (s_atomic_add_x2, returns the old value): its latency is counted by lgkmcnt, which
    // nothing else in the loop uses, so waiting for it on the spot stalls this wave's issue for one L2 round trip per item (the other
    // wave of the SIMD runs on) but leaves vmcnt alone -- the row pairs in flight stay in flight.  (A vector atomic drawn one item
    // ahead was built first: the compiler's atomic optimizer broadcasts the result at the draw site behind an s_waitcnt vmcnt(0),
    // which drains the prefetch ring once per item.)
    auto draw = [&]() -> unsigned {
        unsigned long long t = 1ull;
        asm volatile("s_atomic_add_x2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(tick) : "memory");
        return (unsigned)t;   // items per launch < 2^31 (host check)
    };
    auto make_item = [&](unsigned li, FRItem& it) {   // item li of this queue: band-major, the strips of a band neighbours
        if (li >= nitems) {
            it.done = true;
            return;
        }
        // ordinary items: band li / kstrips of the run; in the tapered tail 2 / 4 consecutive rows of strips share a band
        const unsigned l2 = li < n0 ? li : (li < n1 ? li - n0 : li - n1);
        const unsigned sb = (unsigned)(((unsigned long long)l2 * inv_k) >> 32);
        const unsigned ks = l2 - sb * (unsigned)kstrips;
        const unsigned lg = li < n0 ? 0u : (li < n1 ? 1u : 2u);          // pieces per band = 1 << lg
        const unsigned bl = li < n0 ? sb : (li < n1 ? nbn + (sb >> 1) : nbn + t1 + (sb >> 2));
        const unsigned piece = sb & ((1u << lg) - 1u);
        const unsigned band = gb0 + bl;                                  // of the batch, frame-major
        const unsigned f = (unsigned)(((unsigned long long)band * a.inv_bpf) >> 32);
        const unsigned j = band - f * (unsigned)a.bpf;
        const int frame = (int)f;
        // edge strips: 0, then the trailing ones; interior strips: 1 .. nstrips - n_edge
        const int strip = EDGE ? (ks == 0 ? 0 : a.nstrips - a.n_edge + (int)ks) : 1 + (int)ks;
        it.X = strip * 768;
        const int bys = (int)(((unsigned long long)j * a.bmul) >> 20), bnr = (int)(((unsigned long long)(j + 1) * a.bmul) >> 20) - bys;
        it.ys = bys + (int)(((unsigned)bnr * piece) >> lg);
        it.nrows = bys + (int)(((unsigned)bnr * (piece + 1u)) >> lg) - it.ys;
        it.P = (it.nrows + 1) / 2 + NP - 1;
        it.interior = it.ys >= RAD && it.ys + it.nrows + RAD + 1 <= a.rows;   // (+ 1: an odd band's last pair holds one row more)
        it.sf = a.src + (size_t)frame * a.sfs;
        it.df = a.dst + (size_t)frame * a.dfs;
        it.done = false;
    };

    // Three cursors over the same item sequence: rc requests row pairs, pc prepares operands, oc computes / stores.  `pend` is the item
    // rc entered last; pc and oc take it over when they reach the end of theirs (they follow within RP - 1 < P steps).  Per cursor,
    // what the hot loop needs is kept as running values: a scalar row offset advanced per step and the lane's column offset as a
    // register set once per item.
    FRItem pend;
    pend.X = 0; pend.ys = 0; pend.nrows = 0; pend.P = 1; pend.interior = false; pend.done = false; pend.sf = a.src; pend.df = a.dst;
    // (one item per ticket: two / four neighbouring strips per ticket -- fewer draws -- measured 6 / 12 % SLOWER, tools/ablate_chain_misc.py:
    //  the strips of a band must be in flight together)
    auto next_item = [&]() { make_item(draw(), pend); };
    next_item();
    if (pend.done) return;               // (every wave draws its items + 1 tickets of every queue it visits: the host's accounting)

    // request cursor
    const uint8_t* r_sf;
    int ri, r_P, r_y;                    // pair ri of r_P; r_y = first row of the pair (may lie outside the frame: mirrored)
    int r_last;                          // the item's last source row (ys + nrows - 1 + RAD): rows past it re-read it
    unsigned r_off;                      // (interior items) r_y * sstep
    bool r_int, r_done = false;
    unsigned r_cb;                       // lane: clamped chunk offset
    auto enter_rc = [&]() {
        r_sf = pend.sf; ri = 0; r_P = pend.P; r_y = pend.ys - RAD; r_last = pend.ys + pend.nrows - 1 + RAD; r_int = pend.interior;
        if constexpr ((DBG & 128) != 0) {   // (experiment: no halo rows -- the band's own rows only, the last one re-read)
            r_y = pend.ys; r_last = pend.ys + pend.nrows - 1; r_int = false;
        }
        r_off = (unsigned)r_y * sstep;
        // (EDGE: chunks that stick out of the row are read shifted into it)
        r_cb = EDGE ? (unsigned)min(max(pend.X + lane_cb, 0), rb - 48) : (unsigned)(pend.X + lane_cb);
    };
    // prepare cursor
    int pi, p_P;
    bool p_fl = false, p_fr = false;
    auto enter_pc = [&]() {
        pi = 0; p_P = pend.P;
        if (EDGE) {
            const int cb = pend.X + lane_cb;
            p_fl = cb < 0;
            p_fr = cb == rb - rv * 3;
        }
    };
    // output cursor
    uint8_t* o_df;
    int oi, o_P, o_nrows;
    unsigned o_off;                      // row offset of the step's first output row
    unsigned o_so;                       // lane: byte offset in the row
    bool o_in = true;                    // lane: inside the row (a partial last strip)
    auto enter_oc = [&]() {
        o_df = pend.df; oi = 0; o_P = pend.P; o_nrows = pend.nrows; o_off = (unsigned)pend.ys * dstep;
        o_so = (unsigned)(pend.X + lane_so);
        if (EDGE) o_in = (int)o_so < rb;
    };
    enter_rc();
    enter_pc();
    enter_oc();

    v4i W[RP][3];
    auto request = [&](int slot) {
        v4i(&dst)[3] = W[slot];
        unsigned off;
        if (r_int) {
            off = r_off + (h ? sstep : 0u) + r_cb;
        } else {   // items at the top / bottom of a frame: mirrored row indices (BORDER_REFLECT_101)
            const int y0 = min(r_y, r_last), y1 = min(r_y + 1, r_last);
            const int s0 = y0 < 0 ? -y0 : (y0 >= a.rows ? 2 * a.rows - 2 - y0 : y0), s1 = y1 < 0 ? -y1 : (y1 >= a.rows ? 2 * a.rows - 2 - y1 : y1);
            off = (h ? (unsigned)s1 * sstep : (unsigned)s0 * sstep) + r_cb;
        }
        dst[0] = *(const v4i*)(r_sf + off);
        dst[1] = *(const v4i*)(r_sf + off + 16);
        dst[2] = *(const v4i*)(r_sf + off + 32);
        r_y += 2;
        r_off += 2 * sstep;
        if (++ri == r_P) {
            if (!r_done) next_item();
            if (pend.done) {                         // past the last one: keep re-reading its last pair (cache hits, never used)
                r_done = true;
                ri = r_P - 1;
                r_y -= 2;
                r_off -= 2 * sstep;
            } else {
                enter_rc();
            }
        }
    };
    auto prepare = [&](int slot) {
        v4i(&w)[3] = W[slot];
        uint32_t pb[4], pg[4], prr[4];
        const uint32_t r[12] = {(uint32_t)w[0][0], (uint32_t)w[0][1], (uint32_t)w[0][2], (uint32_t)w[0][3], (uint32_t)w[1][0], (uint32_t)w[1][1],
                                (uint32_t)w[1][2], (uint32_t)w[1][3], (uint32_t)w[2][0], (uint32_t)w[2][1], (uint32_t)w[2][2], (uint32_t)w[2][3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) deint4w(r[3 * i], r[3 * i + 1], r[3 * i + 2], pb[i], pg[i], prr[i]);
        if (EDGE) {
            if (p_fl) {   // the lane read pixels 0..15 instead of -4..11: shift by one dword; pixels -3..-1 mirror 3, 2, 1
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    uint32_t* pp = pl == 0 ? pb : (pl == 1 ? pg : prr);
                    pp[3] = pp[2];
                    pp[2] = pp[1];
                    pp[1] = pp[0];
                    pp[0] = __builtin_amdgcn_perm(pp[0], pp[0], 0x01020300u);
                }
            }
            if (p_fr) {   // the lane read pixels cols-16..cols-1 instead of cols-rv..: its rv valid pixels move to the front, then the mirror
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    uint32_t* pp = pl == 0 ? pb : (pl == 1 ? pg : prr);
                    const uint32_t mir = __builtin_amdgcn_perm(pp[3], pp[3], 0x00000102u);
                    if (rv == 4) {
                        pp[0] = pp[3];
                        pp[1] = mir;
                    } else if (rv == 8) {
                        pp[0] = pp[2];
                        pp[1] = pp[3];
                        pp[2] = mir;
                    } else if (rv == 12) {
                        pp[0] = pp[1];
                        pp[1] = pp[2];
                        pp[2] = pp[3];
                        pp[3] = mir;
                    } else {
                        pp[0] = mir;
                    }
                }
            }
        }
        w[0] = v4i{(int)(pb[0] ^ 0x80808080u), (int)(pb[1] ^ 0x80808080u), (int)(pb[2] ^ 0x80808080u), (int)(pb[3] ^ 0x80808080u)};
        w[1] = v4i{(int)(pg[0] ^ 0x80808080u), (int)(pg[1] ^ 0x80808080u), (int)(pg[2] ^ 0x80808080u), (int)(pg[3] ^ 0x80808080u)};
        w[2] = v4i{(int)(prr[0] ^ 0x80808080u), (int)(prr[1] ^ 0x80808080u), (int)(prr[2] ^ 0x80808080u), (int)(prr[3] ^ 0x80808080u)};
    };
    auto store_row = [&](v4i(&acc)[3], const v4i(&acc2)[3], unsigned roff, bool ok) {
        if constexpr (DMASK != 0) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) acc[pl] += acc2[pl] << a.dual_shift;
        }
        U3w o;
        if constexpr ((VAR & CV_PACK2) != 0) {
            const int v[12] = {acc[0][0], acc[1][0], acc[2][0], acc[0][1], acc[1][1], acc[2][1], acc[0][2], acc[1][2], acc[2][2], acc[0][3], acc[1][3], acc[2][3]};
            rcv_ashr_sat_pk12_mfma(v, a.shift, o.a, o.b, o.c);
        } else {
            o.a = rcv_ashr_sat_pk4(acc[0][0], acc[1][0], acc[2][0], acc[0][1], a.shift);
            o.b = rcv_ashr_sat_pk4(acc[1][1], acc[2][1], acc[0][2], acc[1][2], a.shift);
            o.c = rcv_ashr_sat_pk4(acc[2][2], acc[0][3], acc[1][3], acc[2][3], a.shift);
        }
        // non-temporal: the launch never reads its output back; (EDGE) windows past the row end (a partial last strip) are masked off
        if (ok && o_in) __builtin_nontemporal_store(v3i{(int)o.a, (int)o.b, (int)o.c}, (v3i*)(o_df + (size_t)(roff + o_so)));
    };

#pragma unroll
    for (int i = 0; i < RP - 1; ++i) request(i);
#pragma unroll
    for (int i = 0; i < NP - 1; ++i) {
        prepare(i);
        ++pi;
    }

    for (;;) {
#pragma unroll
        for (int s = 0; s < RP; ++s) {
            request((s + RP - 1) % RP);
            const bool valid = oi <= o_P - NP;          // the window's NP pairs belong to one item
            const bool two = 2 * oi + 1 < o_nrows;      // ... and its second output row exists (odd band heights)
            if constexpr ((DBG & 256) != 0) {
                // the chain's memory pattern alone: its loads, and its stores fed with loaded bytes
                const v4i& w0 = W[s % RP][0];
                const v4i& w1 = W[s % RP][1];
                if (valid && o_in) __builtin_nontemporal_store(v3i{w0[0], w0[1], w0[2]}, (v3i*)(o_df + (size_t)(o_off + o_so)));
                if (valid && two && o_in) __builtin_nontemporal_store(v3i{w1[0], w1[1], w1[2]}, (v3i*)(o_df + (size_t)(o_off + dstep + o_so)));
            } else if (valid) {   // (wave-uniform)
                prepare((s + NP - 1) % RP);
                v4i acc[2][3], acc2[2][3];
                const v4i zerov = v4i{0, 0, 0, 0};
#pragma unroll
                for (int par = 0; par < 2; ++par)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            acc[par][pl] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[par][p], W[(s + p) % RP][pl], p == 0 ? initv : acc[par][pl], 0, 0, 0);
                            if constexpr (DMASK != 0) {   // second weight table (fr_segment has the scheme)
                                const int bits = (DMASK >> (par * NP)) & ((1 << NP) - 1);
                                if ((bits >> p) & 1) {
                                    const bool first = (bits & ((1 << p) - 1)) == 0;
                                    acc2[par][pl] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A2[par][p], W[(s + p) % RP][pl], first ? zerov : acc2[par][pl], 0, 0, 0);
                                }
                            }
                        }
                store_row(acc[0], acc2[0], o_off, true);
                store_row(acc[1], acc2[1], o_off + dstep, two);
            } else {              // the NP - 1 windows that straddle two items: operands only
                prepare((s + NP - 1) % RP);
            }
            o_off += 2 * dstep;
            if (++pi == p_P) enter_pc();
            if (++oi == o_P) {
                if (pend.done) return;   // (rc found the list empty while oc was in this item: it was the wave's last one)
                enter_oc();
            }
        }
    }
}

template <int KS, int PP, int DBG, int DMASK = 0, int VAR = 0>
__global__ __launch_bounds__(64, 2) void k_filter_rows_chain(FRArgs a)
{
    constexpr int NP = (KS + 1) / 2;
    const int lane = threadIdx.x;
    // the XCD this wave RUNS on (bits 3:0 of the XCC_ID register), not the one its block index suggests: the ticket counters are
    // L2-scope and one XCD's L2 is only coherent with itself
    const int xcd = __builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u));   // hwreg(HW_REG_XCC_ID, 0, 4); wave-uniform
    const int slot = (int)(blockIdx.x >> 3);   // of this XCD (with the usual placement)
    unsigned long long t_start = 0;
    if constexpr ((VAR & CV_TRACE) != 0) t_start = __builtin_amdgcn_s_memrealtime();
    // the first wave of the launch checks that the context's previous chained launch drew all of its items (the launch boundary has made
    // every XCD's counters visible); the last launch before a host-side wait is checked by k_chain_check (rcv_chain_flush)
    if (blockIdx.x == 0 && a.prev_tickets) fr_check_set(a.prev_tickets, a.prev_cbands, a.prev_kint, a.prev_kedge, a.prev_tp1, a.prev_tp2, a.fault, lane);
    if (xcd == a.drop_xcd) return;             // (fault injection for the completion check's test)
    {   // (see "Tickets" above; uniform addresses: a lane-dependent index here made the compiler keep the counter addresses in VGPRs)
        int xz = xcd;
        asm volatile("" : "+s"(xz));   // (its own copy: shared with the draw addresses, the vector store pulled the whole address chain into VGPRs)
        unsigned long long* const nx = a.tickets_next + 32 * xz;
        if (lane == 0) {
            nx[0] = 0ull;
            nx[16] = 0ull;
        }
    }
    v4i A[2][NP], A2[2][NP];
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const uint4 w = a.wtab[(par * NP + p) * 64 + lane];
            A[par][p] = v4i{(int)w.x, (int)w.y, (int)w.z, (int)w.w};
            if constexpr (DMASK != 0) {
                if ((DMASK >> (par * NP + p)) & 1) {
                    const uint4 w2 = a.wtab[(2 * NP + par * NP + p) * 64 + lane];
                    A2[par][p] = v4i{(int)w2.x, (int)w2.y, (int)w2.z, (int)w2.w};
                }
            }
        }
    v4i initv = v4i{a.acc_init, a.acc_init, a.acc_init, a.acc_init};
    asm volatile("" : "+v"(initv));
    // the first edge_waves slots of every XCD start on the edge queue (their share of the work, a little more because they are the
    // slower kind), the others on the interior queue; whoever finds its queue empty helps with the other one
    bool edge = slot < a.edge_waves;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {   // (one copy of each loop in the binary)
        if (edge) fr_chain_run<KS, PP, true, DBG, DMASK, VAR>(a, lane, xcd, A, A2, initv);
        else fr_chain_run<KS, PP, false, DBG, DMASK, VAR>(a, lane, xcd, A, A2, initv);
        edge = !edge;
    }
    if constexpr ((VAR & CV_TRACE) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wave's last stores have left
        if (lane == 0 && a.trace) {
            a.trace[2 * (size_t)blockIdx.x] = t_start;
            a.trace[2 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

template <int KS, int PP, int DBG, int DMASK = 0, int SRC = 0, int SOB = 0>
__global__ __launch_bounds__(SOB ? 256 : 512, (SOB && PP <= 2) ? 3 : 2) void k_filter_rows_mfma(FRArgs a)   // (SOB: see kSobPP)
{
    // A workgroup is `wpb` independent waves (nothing is shared, no barrier): wave w of workgroup b takes item slot
    // (b >> 3) * wpb + w of XCD b & 7, so that the waves of one workgroup -- one CU -- are neighbouring strips of one band: their
    // seam lines meet in that CU's L1 and the CU streams wpb * 768 contiguous bytes per row (RCV_FR_WPB; DESIGN_HISTORY.md 4.1).
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned long long t_start = 0;
    if constexpr ((DBG & 1024) != 0) t_start = __builtin_amdgcn_s_memrealtime();
    // XCD-aware order (speed only): hardware places block b on XCD b % 8; each XCD gets a contiguous run of bands, and the strips
    // of one band -- which share the 128-B lines at their seams -- are neighbours in dispatch order on one L2
    const int xcd = blockIdx.x & 7;
    int slot = (int)(blockIdx.x >> 3) * a.wpb + wave;
    // (DBG & 512, with 256: the memory pattern of TWO waves per (band, strip) that take alternate output row pairs -- both read every
    //  input row, wave phi stores the row pairs u with (u & 1) == phi: half as many bands in flight at the same occupancy)
    const int phi = (DBG & 512) ? (slot & 1) : 0;
    if constexpr ((DBG & 512) != 0) slot >>= 1;
    // Band order: every XCD works through its own contiguous eighth of the bands, so that what ONE XCD has in flight is a
    // compact piece of the batch (~21 neighbouring bands = one frame).  Measured on 64 4K frames (same box, same run): this
    // order 0.551 ms; bands dealt round-robin to the XCDs (order 1: each XCD's waves spread over eight frames) 0.600 ms; one
    // XCD's concurrent bands spread over its whole eighth 0.639 against 0.579 ms -- the wider the address range an XCD touches
    // at one time, the slower its memory path (translation reach per XCD is the likely cause).  RCV_FR_ORDER keeps it measurable.
    // (Round 3, DESIGN_HISTORY.md 4.1 "sweep orders": order 1 with bands down to 8 rows -- the whole GPU inside one moving window of a
    // few MB -- and a plain raster over (band, strip) were measured too: never better than this order, short bands much worse.)
    const int strip = slot % a.nstrips;
    const int bi = slot / a.nstrips;   // dispatch order of this band on its XCD
    if (bi >= a.bands_per_xcd) return;
    const int band = a.order == 0 ? xcd * a.bands_per_xcd + bi : bi * 8 + xcd;
    if (a.tn == 0 && band >= a.nbands) return;
    // byte offset of the strip's TILE in a destination row (gray: also the pixel offset); SOB: the workgroup's four waves are `strips`
    // 4 g .. 4 g + 3 = the four waves of group g, 960 output pixels apart, wave t beginning 256 t pixels into the group (fr_segment)
    const int X = SOB ? 3 * (960 * (strip >> 2) + 256 * (strip & 3)) : strip * 768;
    __shared__ uint32_t seam_lds[SOB ? 64 : 1];   // (SOB: two buffers x two rows x 16 dword slots, nine of them used)
    // the last chunk a strip touches ends at destination byte X + 804 (gray: at pixel X + 780)
    const bool edge = X == 0 || (SRC == 2 ? X + 780 > a.cols : X + 804 > a.cols * 3);
    const long long G = (long long)a.nframes * a.rows;
    long long g0 = G * band / a.nbands;
    long long g1 = G * (band + 1) / a.nbands;
    if (a.tn > 0) {   // tapered: run r of this XCD's list, band b of the run (scalar)
        int b = bi, r = 0;
        while (r < a.tn - 1 && b >= a.tnb[r]) b -= a.tnb[r++];
        const long long x0 = G / 8 * xcd + a.tstart[r];
        g0 = x0 + a.tlen[r] * b / a.tnb[r];
        g1 = x0 + a.tlen[r] * (b + 1) / a.tnb[r];
    }
    while (g0 < g1) {   // (a band that crosses a frame boundary is two segments)
        const int frame = (int)(g0 / a.rows), ys = (int)(g0 - (long long)frame * a.rows);
        const int ye = (int)min((long long)a.rows, ys + (g1 - g0));
        const uint8_t* sframe = a.src + (size_t)frame * a.sfs;
        uint8_t* dframe = a.dst + (size_t)frame * a.dfs;
        if constexpr (SOB != 0) {
            // gradient rows ys .. ye-1 need filtered rows ys-1 .. ye
            const int fys = max(ys - 1, 0), fye = min(ye + 1, a.rows);
            uint8_t* dxf = a.gdx + (size_t)frame * a.gfs;
            uint8_t* dyf = a.gdy + (size_t)frame * a.gfs;
            if (edge) fr_segment<KS, PP, true, DBG, DMASK, SRC, SOB>(a, lane, X, fys, fye, sframe, dframe, ys, ye, dxf, dyf, strip & 3, seam_lds);
            else fr_segment<KS, PP, false, DBG, DMASK, SRC, SOB>(a, lane, X, fys, fye, sframe, dframe, ys, ye, dxf, dyf, strip & 3, seam_lds);
        } else if (edge) fr_segment<KS, PP, true, DBG, DMASK, SRC>(a, lane, X, ys, ye, sframe, dframe, phi);
        else fr_segment<KS, PP, false, DBG, DMASK, SRC>(a, lane, X, ys, ye, sframe, dframe, phi);
        g0 += ye - ys;
    }
    if constexpr ((DBG & 1024) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wave's last stores have left
        if (lane == 0 && a.trace) {
            const size_t w = (size_t)blockIdx.x * a.wpb + wave;
            a.trace[2 * w] = t_start;
            a.trace[2 * w + 1] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

// host: A tables.  Table (parity, p): lane (m, qa) holds k = 16 qa + i: row half h = qa >> 1, window pixel j = 16 (qa & 1) + i
// (the window starts 4 pixels left of the tile).  parity 0: pair p carries kernel rows 2p, 2p + 1; parity 1: 2p - 1, 2p.
static void build_rows_wtab(const int8_t* k, int ksize, int8_t* tab)
{
    const int rad = ksize / 2, np = (ksize + 1) / 2;
    for (int par = 0; par < 2; ++par)
        for (int p = 0; p < np; ++p)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 16; ++i) {
                    const int m = lane & 15, qa = lane >> 4, h = qa >> 1, j = 16 * (qa & 1) + i;
                    const int ky = par == 0 ? 2 * p + h : 2 * p - 1 + h;
                    const int t = j - 4 - m + rad;
                    tab[(((par * np) + p) * 64 + lane) * 16 + i] = (int8_t)((ky >= 0 && ky < ksize && t >= 0 && t < ksize) ? k[ky * ksize + t] : 0);
                }
}

template <int KS, int PP>
void launch_rows_dbg(const FRArgs& a, const dim3 grid, unsigned lds, hipStream_t st, int dbg)
{
#ifdef RCV_ROWS_BENCH
    // (the ablation instantiations exist for the default prefetch depth only: 21 variants x 6 depths made this file a four-minute build)
    if constexpr (PP == 3) switch (dbg & 255) {
    case 32: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 32>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 96: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 96>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 128: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 128>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 224: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 224>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 36: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 36>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 100: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 100>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 132: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 132>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 228: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 228>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 37: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 37>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 101: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 101>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 229: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 229>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 1: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 1>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 2: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 2>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 3: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 3>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 5: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 5>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 6: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 6>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 8: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 8>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 16: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 16>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 18: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 18>), grid, dim3(64 * a.wpb), lds, st, a); return;
    case 24: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 1024>), grid, dim3(64 * a.wpb), lds, st, a); return;   // wave timeline (rcv__debug_trace_buffer)
    case 68: RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 4>), grid, dim3(64 * a.wpb), lds, st, a); return;   // (64 + 4: adds instead of MFMAs)
    default: break;
    }
#endif
    // the kernel's own memory-only variant (its loads and its stores, nothing in between): the ceiling of THIS access pattern (the
    // output is not a filtered image); 7: the same with two waves per (band, strip): twice the workgroups for the same band plan
#ifdef RCV_ROWS_BENCH
    if constexpr (KS == 7) {
        if ((dbg & 255) == 7) {
            RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 772>), dim3(grid.x * 2), dim3(64 * a.wpb), lds, st, a);
            return;
        }
        if ((dbg & 255) == 4) {
            RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 260>), grid, dim3(64 * a.wpb), lds, st, a);
            return;
        }
    }
#endif
    (void)dbg;
    RCV_LAUNCH((k_filter_rows_mfma<KS, PP, 0>), grid, dim3(64 * a.wpb), lds, st, a);
}

// The fused filter2D -> gray -> Sobel launch is bound by instruction issue, not by its streams (tools/ablate_sob.py: adds instead of the matrix
// instructions -10 ... -18 %, no stores -11 ... -16 %): TWO row pairs in flight instead of three fit three waves per SIMD (168 VGPRs, four
// of them spilled outside the row loop) -- 64 x 4K: 0.80 -> 0.72 ms on the same box, -9 ... -10 % on three boxes; one pair in flight: -3 %
// (profiles/r06_3f_ab.txt).  The plain filter goes the other way (three waves per SIMD: +6 %, round 3): its streams bind it.
constexpr int kSobPP = 2;

// which row pairs of which parity carry kernel rows [lo, hi]: the DMASK of a second table confined to those rows
constexpr int rows_dmask(int ksize, int lo, int hi)
{
    const int np = (ksize + 1) / 2;
    int m = 0;
    for (int par = 0; par < 2; ++par)
        for (int p = 0; p < np; ++p)
            for (int h = 0; h < 2; ++h) {
                const int ky = par == 0 ? 2 * p + h : 2 * p - 1 + h;
                if (ky >= lo && ky <= hi) m |= 1 << (par * np + p);
            }
    return m;
}
constexpr int kCentre7 = rows_dmask(7, 2, 4);   // second table in kernel rows 2..4 (the integer 7x7 Gaussian): 2 of 4 pairs per parity

template <int KS>
void launch_rows(const FRArgs& a, int pp, unsigned lds, int dmask, int src_yuyv, hipStream_t st, int dbg)
{
    const dim3 grid((unsigned)(((long long)a.bands_per_xcd * a.nstrips + a.wpb - 1) / a.wpb * 8));
    if (a.gdx) {   // filter2D -> gray -> Sobel (BGR source, one weight table: the caller checked)
#ifdef RCV_ROWS_BENCH
        if constexpr (KS == 7) {   // (measurement forms of the fused launch: rcv__filter_rows_sobel_bench)
            switch (dbg) {
            case 1: RCV_LAUNCH((k_filter_rows_mfma<KS, kSobPP, 1, 0, 0, 1>), grid, dim3(64 * a.wpb), lds, st, a); return;
            case 4: RCV_LAUNCH((k_filter_rows_mfma<KS, kSobPP, 4, 0, 0, 1>), grid, dim3(64 * a.wpb), lds, st, a); return;
            case 5: RCV_LAUNCH((k_filter_rows_mfma<KS, kSobPP, 5, 0, 0, 1>), grid, dim3(64 * a.wpb), lds, st, a); return;
            case 2048: RCV_LAUNCH((k_filter_rows_mfma<KS, kSobPP, 2048, 0, 0, 1>), grid, dim3(64 * a.wpb), lds, st, a); return;
            case 2052: RCV_LAUNCH((k_filter_rows_mfma<KS, kSobPP, 2052, 0, 0, 1>), grid, dim3(64 * a.wpb), lds, st, a); return;
            case 4096: RCV_LAUNCH((k_filter_rows_mfma<KS, kSobPP, 4096, 0, 0, 1>), grid, dim3(64 * a.wpb), lds, st, a); return;
            case 1024: RCV_LAUNCH((k_filter_rows_mfma<KS, kSobPP, 1024, 0, 0, 1>), grid, dim3(64 * a.wpb), lds, st, a); return;   // wave timeline
            case 8192: RCV_LAUNCH((k_filter_rows_mfma<KS, 3, 0, 0, 0, 1>), grid, dim3(64 * a.wpb), lds, st, a); return;   // three row pairs in flight, two waves per SIMD (round 6's first form)
            default: break;
            }
        }
#endif
        RCV_LAUNCH((k_filter_rows_mfma<KS, kSobPP, 0, 0, 0, 1>), grid, dim3(64 * a.wpb), lds, st, a);
        return;
    }
    if (src_yuyv == 1) {   // (one weight table only: the caller checked)
        RCV_LAUNCH((k_filter_rows_mfma<KS, 3, 0, 0, 1>), grid, dim3(64 * a.wpb), lds, st, a);
        return;
    }
    if (src_yuyv == 2) {   // one-channel images
        RCV_LAUNCH((k_filter_rows_mfma<KS, 3, 0, 0, 2>), grid, dim3(64 * a.wpb), lds, st, a);
        return;
    }
    constexpr int kAll = (1 << (2 * ((KS + 1) / 2))) - 1;
    if (src_yuyv == 3) {   // BGR, any width / alignment
        if (dmask == 0) RCV_LAUNCH((k_filter_rows_mfma<KS, 3, 0, 0, 3>), grid, dim3(64 * a.wpb), lds, st, a);
        else if (KS == 7 && (dmask & ~kCentre7) == 0) RCV_LAUNCH((k_filter_rows_mfma<KS, 3, 0, KS == 7 ? kCentre7 : kAll, 3>), grid, dim3(64 * a.wpb), lds, st, a);
        else RCV_LAUNCH((k_filter_rows_mfma<KS, 3, 0, kAll, 3>), grid, dim3(64 * a.wpb), lds, st, a);
        return;
    }
    if (dmask != 0) {   // two weight tables
        if (KS == 7 && (dmask & ~kCentre7) == 0) RCV_LAUNCH((k_filter_rows_mfma<KS, 3, 0, KS == 7 ? kCentre7 : kAll>), grid, dim3(64 * a.wpb), lds, st, a);
        else RCV_LAUNCH((k_filter_rows_mfma<KS, 3, 0, kAll>), grid, dim3(64 * a.wpb), lds, st, a);
        return;
    }
#ifdef RCV_ROWS_BENCH   // measurement builds: prefetch depth selectable at run time
    if (KS == 7 && pp == 2) return launch_rows_dbg<7, 2>(a, grid, lds, st, dbg);
    if (KS == 7 && pp == 4) return launch_rows_dbg<7, 4>(a, grid, lds, st, dbg);
    if (KS == 7 && pp == 5) return launch_rows_dbg<7, 5>(a, grid, lds, st, dbg);
    if (KS == 7 && pp == 6) return launch_rows_dbg<7, 6>(a, grid, lds, st, dbg);
    if (KS == 7 && pp == 8) return launch_rows_dbg<7, 8>(a, grid, lds, st, dbg);
#endif
    (void)pp;
    launch_rows_dbg<KS, 3>(a, grid, lds, st, dbg);
}

} // namespace

// Does this launch belong on the row-streaming kernel?  |weights| <= 511; BGR: rows 4-byte aligned, width a multiple of 4 pixels;
// YUYV / gray sources: rows 16-byte aligned, width a multiple of 16; knob RCV_F7_ROWS = 1 takes every eligible shape (tests), 0 none, unset those with enough strip-rows to fill the GPU --
// small launches keep the strip kernel's latency variant.
// gx, gy (both or neither): the i16 gradient planes of the fused filter2D -> BGR2GRAY -> Sobel launch; `d` is not written then
// (BGR source with 4-byte aligned rows and a width that is a multiple of 4, |weights| <= 127, 8-byte aligned gradient rows).
static int rows_launch(rcv_ctx* ctx, const View& s, const View& d, const int16_t* k, int ksize, int shift, int src_yuyv, bool any_size, const View* gx,
                       const View* gy, const RowsTune& kn);

#ifndef RCV_ROWS_BENCH
// Completion check of the chained kernel, host side (rcv_internal.h has the scheme).
int rcv_chain_flush(rcv_ctx* ctx)
{
    // (both lanes' checks run on the context's stream: callers come through rcv_bind, which has made it wait for the half stream)
    for (int l = 0; l < 2; ++l) {
        rcv_ctx::ChainLane& ln = ctx->fr_lane[l];
        if (!ln.unchecked) continue;
        ln.unchecked = false;
        const unsigned long long* set = (const unsigned long long*)(ctx->kconst + (l ? RCV_KC_FR_TICKETS2 : RCV_KC_FR_TICKETS) + 2048 * ((ln.seq - 1u) & 3u));
        hipLaunchKernelGGL(k_chain_check, dim3(1), dim3(64), 0, ctx->stream, set, ln.prev[0], ln.prev[1], ln.prev[2], ln.prev[3], ln.prev[4], ctx->fr_fault);
        RCV_TRY(rcv_launch_check(ctx));
    }
    return RCV_OK;
}
int rcv_chain_poll(rcv_ctx* ctx)
{
    // (after a fault no chained launch is enqueued any more; the ones that were already queued behind the faulty one raise the word again --
    //  old news: the fault has been reported and those launches belong to the same failed stretch of the stream)
    if (!ctx->fr_fault || ctx->fr_chain_off || __atomic_load_n(ctx->fr_fault, __ATOMIC_ACQUIRE) == 0u) return RCV_OK;
    // some queue of a chained launch was left short: what that launch (and the chained ones after it) wrote is incomplete.  Reported once;
    // the counters are zeroed again before they are used and this context stays on the one-band-per-wave kernel from here on.
    __atomic_store_n(ctx->fr_fault, 0u, __ATOMIC_RELEASE);
    for (rcv_ctx::ChainLane& l : ctx->fr_lane) {
        l.tickets_ready = false;
        l.unchecked = false;
    }
    ctx->fr_chain_off = true;
    return RCV_ERR_DEVICE;
}

int rcv_filter_i16_rows(rcv_ctx* ctx, const View& s, const View& d, const int16_t* k, int ksize, int shift, int src_yuyv, bool any_size,
                        const View* gx, const View* gy)
{
    // (Round 3 experiments, removed: the batch as consecutive launches of 2 ... 32 frames -- 0.59 ... 0.82 ms against 0.55 for the one
    //  launch -- and its two halves as concurrent launches on two streams, forked and joined per call: 0.615 against 0.582 ms;
    //  profiles/r03_ablate_chunked_launches.txt.)
    const RcvKnobs& g = rcv_knobs();
    RowsTune t;
    t.f7_rows = g.f7_rows;
    t.dual_full = g.f7_dual_full;
    t.chain = g.fr_chain;
    t.chain_rows = g.fr_chain_rows;
    return rows_launch(ctx, s, d, k, ksize, shift, src_yuyv, any_size, gx, gy, t);
}

// ---- one call, two halves, two streams (rcv_internal.h: rcv_ctx::half has the contract, rcv_split_run the mechanism) ---------------------
// The chained filter2D of a batch of 16+ BGR frames as two launches: frames [0, n / 2) on the context's stream, the rest on its half stream, each
// with its own lane of ticket counters.  RCV_ERR_UNSUPPORTED = not taken, NOTHING was enqueued (the caller goes through rcv_bind and the ordinary
// path).  Called INSTEAD of rcv_bind.
int rcv_filter_i8_split(rcv_ctx* ctx, const View& s, const View& d, const int8_t* k, int ksize, int shift)
{
    const RcvKnobs& g = rcv_knobs();
    if (!ctx || g.fr_chain == 0 || g.f7_rows == 0 || g.fr_chain_drop_xcd >= 0 || ctx->fr_chain_off) return RCV_ERR_UNSUPPORTED;
    if (s.ch != 3 || d.ch != 3 || s.rows < 64 || ctx->cu_count != 256 || (ksize != 3 && ksize != 5 && ksize != 7) || s.n < 16) return RCV_ERR_UNSUPPORTED;
    const int h = s.n / 2;
    RcvRanges ra, rb;
    ra.read(s, 0, h); ra.write(d, 0, h);
    rb.read(s, h, s.n); rb.write(d, h, s.n);
    int16_t k16[49];
    for (int i = 0; i < ksize * ksize; ++i) k16[i] = k[i];
    return rcv_split_run(ctx, s.n, ra, rb, [&](int half) {
        RowsTune t;
        t.f7_rows = g.f7_rows;
        t.chain = g.fr_chain;
        t.chain_rows = g.fr_chain_rows;
        t.must_chain = true;   // (its halves are chained launches or the call is not split)
        t.lane = half;
        const int f0 = half ? h : 0, f1 = half ? s.n : h;
        return rows_launch(ctx, rcv_view_frames(s, f0, f1), rcv_view_frames(d, f0, f1), k16, ksize, shift, 0, true, nullptr, nullptr, t);
    });
}
#else
// Measurement entry (librustcv_hip_bench.so): the BGR -> BGR filter of a device-resident batch with every plan parameter explicit.
// tune: 15 ints {f7_rows, dual_full, chain, chain_rows, dbg, wpc, rounds, pp, order, bpf, band_rows, taper, wpb, edge_pct, var}; trace: the
// per-wave timeline buffer (dbg 24) or NULL.  With dbg != 0 the output is NOT a filtered image.
extern "C" int rcv__filter_rows_bench(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst, const int8_t* k, int ksize, int shift, const int* tune, void* trace)
{
    RCV_TRY(rcv_bind(ctx));
    if (!src || !dst || !k || !tune || (ksize != 3 && ksize != 5 && ksize != 7) || shift < 0 || shift > 24) return RCV_ERR_ARG;
    View s, d;
    RCV_TRY(rcv_view_batch(src, RCV_8U, &s));
    RCV_TRY(rcv_view_batch(dst, RCV_8U, &d));
    if (s.rows != d.rows || s.cols != d.cols || s.n != d.n || s.ch != 3 || d.ch != 3) return RCV_ERR_ARG;
    int16_t k16[49];
    for (int i = 0; i < ksize * ksize; ++i) k16[i] = k[i];
    RowsTune t;
    t.f7_rows = tune[0]; t.dual_full = tune[1]; t.chain = tune[2]; t.chain_rows = tune[3]; t.dbg = tune[4]; t.wpc = tune[5]; t.rounds = tune[6];
    t.pp = tune[7]; t.order = tune[8]; t.bpf = tune[9]; t.band_rows = tune[10]; t.taper = tune[11]; t.wpb = tune[12]; t.edge_pct = tune[13];
    t.var = tune[14];
    t.trace = trace;
    return rows_launch(ctx, s, d, k16, ksize, shift, 0, true, nullptr, nullptr, t);
}
// ... and the fused filter2D -> gray -> Sobel launch of a device-resident batch (dx, dy: i16 planes), same tune array
extern "C" int rcv__filter_rows_sobel_bench(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dx, rcv_batch* dy, const int8_t* k, int ksize, int shift, const int* tune,
                                            void* trace)
{
    RCV_TRY(rcv_bind(ctx));
    if (!src || !dx || !dy || !k || !tune || (ksize != 3 && ksize != 5 && ksize != 7) || shift < 0 || shift > 24) return RCV_ERR_ARG;
    View s, vx, vy;
    RCV_TRY(rcv_view_batch(src, RCV_8U, &s));
    RCV_TRY(rcv_view_batch(dx, RCV_16S, &vx));
    RCV_TRY(rcv_view_batch(dy, RCV_16S, &vy));
    int16_t k16[49];
    for (int i = 0; i < ksize * ksize; ++i) k16[i] = k[i];
    RowsTune t;
    t.f7_rows = tune[0]; t.dual_full = tune[1]; t.chain = tune[2]; t.chain_rows = tune[3]; t.dbg = tune[4]; t.wpc = tune[5]; t.rounds = tune[6];
    t.pp = tune[7]; t.order = tune[8]; t.bpf = tune[9]; t.band_rows = tune[10]; t.taper = tune[11]; t.wpb = tune[12]; t.edge_pct = tune[13];
    t.var = tune[14];
    t.trace = trace;   // (dbg 1024: two 64-bit words per wave, start / end of the 100 MHz counter)
    return rows_launch(ctx, s, s, k16, ksize, shift, 0, true, &vx, &vy, t);
}
#endif

static int rows_launch(rcv_ctx* ctx, const View& s, const View& d, const int16_t* k, int ksize, int shift, int src_yuyv, bool any_size, const View* gx,
                       const View* gy, const RowsTune& kn)
{
    const bool sob = gx != nullptr;
    if (sob != (gy != nullptr)) return RCV_ERR_ARG;
    if (kn.f7_rows == 0) return RCV_ERR_UNSUPPORTED;
    if (ksize != 3 && ksize != 5 && ksize != 7) return RCV_ERR_UNSUPPORTED;
    // src_yuyv: 0 BGR, 1 packed YUYV source, 2 one-channel (gray) source and destination
    const bool gray = src_yuyv == 2;
    if (s.ch != (gray ? 1 : (src_yuyv ? 2 : 3)) || d.ch != (gray ? 1 : 3)) return RCV_ERR_UNSUPPORTED;
    // widths: a multiple of 16 pixels; BGR -> BGR also any multiple of 4 (the right-border repair knows the four residues) with
    // rows that are only 4-byte aligned (a packed 1080-pixel-wide portrait frame: step = 3240): its loads and stores are dword-
    // aligned dwordx4 / dwordx3, which cost the same as 16-byte aligned ones
    // BGR -> BGR on any other width / alignment (an odd width of a packed image: byte-aligned rows): the SRC = 3 instantiation
    // (aligned dwords + v_alignbyte on the load side, byte-granular border repair, unaligned stores)
    const int wq = src_yuyv == 0 ? 4 : 16, al = src_yuyv == 0 ? 4 : 16;
    if (s.cols < 16 || s.rows < 4) return RCV_ERR_UNSUPPORTED;
    const bool fits = s.cols % wq == 0 && !((uintptr_t)s.p % al || s.step % al || (s.n > 1 && s.fstride % al)) &&
                      !((uintptr_t)d.p % al || d.step % al || (d.n > 1 && d.fstride % al));
    if (!fits) {
        if (src_yuyv != 0 || sob) return RCV_ERR_UNSUPPORTED;
        src_yuyv = 3;
    }
    if (sob) {
        if (src_yuyv != 0) return RCV_ERR_UNSUPPORTED;
        for (const View* g : {gx, gy})
            if (g->ch != 1 || g->rows != s.rows || g->cols != s.cols || g->n != s.n || (uintptr_t)g->p % 8 || g->step % 8 || (g->n > 1 && g->fstride % 8))
                return RCV_ERR_UNSUPPORTED;
        if (gx->step != gy->step || gx->fstride != gy->fstride) return RCV_ERR_UNSUPPORTED;
    }
    const long long rb = (long long)s.cols * (gray ? 1 : 3);
    // in-frame source offsets are 32-bit
    if (rb >= (1 << 30) || (unsigned long long)s.rows * s.step >= (1ull << 32)) return RCV_ERR_UNSUPPORTED;
    // (sob, rounds 3-5: single waves 240 pixels apart with plain stores; line-aligned 192-pixel strips with non-temporal stores measured 7 %
    //  slower in round 3 -- 33 % instead of 7 % redundant matrix work.  Round 6: groups of four waves, see fr_segment)
    const int nstrips = sob ? 4 * ((s.cols + 959) / 960) : (int)((rb + 767) / 768);   // (sob: four waves per group of 960 output pixels)
    const long long G = (long long)s.n * s.rows;
    // (any_size: shapes the strip kernel does not take -- widths that are not a multiple of 16 -- where the alternative is the
    //  streaming VALU kernel: 4-7x slower even on one frame)
    // launches below ~16 000 strip-rows (two 1080p BGR frames) keep the strip kernel's latency variant; above, this kernel with its
    // per-SIMD band plan is faster (round 3, tools/small_filter_latency.py: one 4K frame 7x7 11.4 against 16.4 us)
    if (kn.f7_rows < 0 && !any_size && G * nstrips < 16000LL * ctx->cu_count / 256) return RCV_ERR_UNSUPPORTED;

    // weights beyond i8: K = M + 2 * S when every weight fits that split (|w| <= 381; the integer Gaussian), else K = 4Q + R
    const int nk = ksize * ksize, np = (ksize + 1) / 2;
    long long ksum = 0;
    bool dual = false, split2 = true;
    for (int i = 0; i < nk; ++i) {
        if (k[i] < -512 || k[i] > 511) return RCV_ERR_UNSUPPORTED;
        if (k[i] < -128 || k[i] > 127) dual = true;
        if (k[i] > 127 + 2 * 127 || k[i] < -128 - 2 * 128) split2 = false;
        ksum += k[i];
    }
    if (dual && kn.dual_full) split2 = false;   // (the knob keeps the 4Q + R split testable)
    if (dual && (src_yuyv == 1 || src_yuyv == 2 || sob)) return RCV_ERR_UNSUPPORTED;   // (two tables: BGR -> BGR only)

    // weight tables: a small cache keyed by the kernel (see rcv_ctx::fr_tab)
    if (!ctx->fr_tabs) RCV_HIP(hipMalloc((void**)&ctx->fr_tabs, 4 * 16384));
    int slot = -1;
    for (int e = 0; e < 4; ++e) {
        const rcv_ctx::FrTab& t = ctx->fr_tab[e];
        if (t.valid && t.ksize == ksize && t.split2 == split2 && memcmp(t.k, k, (size_t)nk * sizeof(int16_t)) == 0) slot = e;
    }
    if (slot < 0) {
        slot = 0;
        for (int e = 0; e < 4; ++e) {   // an empty entry, else the least recently used
            if (!ctx->fr_tab[e].valid) { slot = e; break; }
            if (ctx->fr_tab[e].stamp < ctx->fr_tab[slot].stamp) slot = e;
        }
        rcv_ctx::FrTab& t = ctx->fr_tab[slot];
        if (!t.uploaded) RCV_HIP(hipEventCreateWithFlags(&t.uploaded, hipEventDisableTiming));
        else RCV_HIP(hipEventSynchronize(t.uploaded));   // (its previous upload has long passed: three other kernels were used since)
        t.valid = false;
        int8_t m8[49], s8[49];
        int dmask = 0;
        for (int i = 0; i < nk; ++i) {
            const int w = k[i];
            int sv = 0;
            if (dual) sv = split2 ? (w > 127 ? (w - 127 + 1) / 2 : (w < -128 ? -((-w - 128 + 1) / 2) : 0)) : (w >> 2);
            m8[i] = (int8_t)(dual ? (split2 ? w - 2 * sv : w - 4 * sv) : w);
            s8[i] = (int8_t)sv;
        }
        static_assert(sizeof(t.host) == 2 * 2 * 4 * 64 * 16, "two weight tables of four row pairs");
        build_rows_wtab(m8, ksize, t.host);
        if (dual) {
            build_rows_wtab(s8, ksize, t.host + (size_t)2 * np * 1024);
            for (int q = 0; q < 2 * np; ++q)
                for (int i = 0; i < 1024; ++i)
                    if (t.host[(size_t)(2 * np + q) * 1024 + i]) { dmask |= 1 << q; break; }
        }
        // stream-ordered behind every kernel that still reads this entry's old table; the host copy lives in the context
        RCV_HIP(hipMemcpyAsync(ctx->fr_tabs + (size_t)slot * 16384, t.host, (size_t)(dual ? 4 : 2) * np * 1024, hipMemcpyHostToDevice, ctx->stream));
        RCV_HIP(hipEventRecord(t.uploaded, ctx->stream));
        ++ctx->fr_uploads;   // (on the context's stream: the half stream must wait for it before it launches with this table)
        memcpy(t.k, k, (size_t)nk * sizeof(int16_t));
        t.ksize = ksize;
        t.split2 = split2;
        t.dmask = dmask;
        t.valid = true;
    }
    ctx->fr_tab[slot].stamp = ++ctx->fr_clock;
    const int dmask = ctx->fr_tab[slot].dmask;

    FRArgs a = {};
    a.src = s.p;
    a.dst = d.p;
    a.wtab = (const uint4*)(ctx->fr_tabs + (size_t)slot * 16384);
    a.dump = ctx->kconst + RCV_KC_FR_DUMP;
    a.sstep = s.step;
    a.dstep = d.step;
    a.sfs = s.fstride;
    a.dfs = d.fstride;
    a.rows = s.rows;
    a.cols = s.cols;
    a.nstrips = nstrips;
    a.nframes = s.n;
    a.dual_shift = split2 ? 1 : 2;
    a.gdx = sob ? gx->p : nullptr;
    a.gdy = sob ? gy->p : nullptr;
    a.gstep = sob ? gx->step : 0;
    a.gfs = sob ? gx->fstride : 0;
    a.trace = (unsigned long long*)kn.trace;
    // occupancy: 174 VGPRs (3 row pairs in flight) = 2 waves per SIMD = 8 waves per CU.  `wpc` is the number of wave slots per CU
    // the BANDS are sized for (10 measured best: slightly more, slightly shorter bands than the resident waves need); the knob
    // RCV_FR_WPC also caps the real occupancy below 8 through a dynamic-LDS request that the kernel never touches (sweeps).
    const int wpc = kn.wpc > 0 ? (kn.wpc > 12 ? 12 : kn.wpc) : (sob ? 12 : 10);   // (sob: three waves per SIMD)
    const unsigned lds = kn.wpc > 0 && wpc < 12 ? (unsigned)((163840 / wpc) & ~511) : 0u;
    int small_plan = 0;
    // bands: the batch's frame-rows in equal parts, `rounds` x as many (band, strip) waves as the GPU holds (measured on 64 4K
    // frames: 4..16 rounds within 1-2 %, one round -- a static partition -- +20 %).  Each band boundary costs 2 * (ksize / 2)
    // halo rows of re-reads.
    {
        const long long slots = (long long)wpc * ctx->cu_count;
        const int rounds = kn.rounds > 0 ? kn.rounds : 8;
        const long long want = rounds * slots / a.nstrips;   // bands for `rounds` fills of the wave slots
        long long nb;
        // launches of a few rounds: every SIMD the same number of equally long waves (rcv_plan_seg_rows with this kernel's own
        // figures: two waves per SIMD at most, a lone wave leaves its SIMD half idle; a band streams ksize - 1 halo rows and fills
        // its pipeline with a few more).  Whole bands per frame.
        const int small = small_plan = kn.bpf > 0 ? 0 : rcv_plan_seg_rows(s.rows, (long long)a.nstrips * s.n, ctx->cu_count, ksize + 5 + (sob ? 2 : 0), 6, 2.3, 1.17, 2, 2);
        if (small > 0 || kn.band_rows > 0) {
            const int br = kn.band_rows > 0 ? kn.band_rows : small;   // (knob: latency sweeps)
            nb = (long long)((s.rows + br - 1) / br) * s.n;
        } else if (s.n >= 8) {
            // a whole number of bands per FRAME: no band straddles a frame (one segment, one pipeline fill per wave), and with
            // n % 8 == 0 every XCD works on whole frames.  Measured on 64 4K frames: 21 bands per frame (8 rounds) 0.569 ms, 13.1
            // (5 rounds) 0.646, 15.75 / 18.4 (6 / 7 rounds) 0.592
            long long bpf = (want + s.n / 2) / s.n;
            long long most = (s.rows + 31) / 32;             // at least 32 rows per band
            if (kn.bpf > 0) bpf = kn.bpf, most = (s.rows + 3) / 4;   // tuning knob (sweep-order ablation: bands down to 4 rows)
            bpf = bpf < 1 ? 1 : (bpf > most ? most : bpf);
            nb = bpf * s.n;
        } else {
            nb = want / 8 * 8;
            nb = nb < 8 ? 8 : nb;
            const long long most = (G + 31) / 32;
            if (nb > most) nb = most;
            if (nb < 1) nb = 1;
        }
        a.nbands = (int)nb;
        a.bands_per_xcd = (int)((nb + 7) / 8);
        a.order = kn.order == 1 ? 1 : 0;
        a.tn = 0;
        // Tapered bands (round 3, tools/wave_timeline.py): with equal bands the launch's LAST round of waves starts spread over
        // one wave duration (56 us at 103-row bands) and the slots then drain for as long -- occupancy below 90 % for the last 60 us
        // of a 605-us launch.  So the tail of every XCD's list -- about one round of its wave slots -- is cut finer: the first half
        // into half-height bands, the second half into quarter-height ones; the last waves to start are the shortest.
        if (small == 0 && kn.band_rows == 0 && a.order == 0 && s.n >= 8 && s.n % 8 == 0 && kn.taper != 0 && nb >= 64) {
            const long long Lx = G / 8;
            const double H = (double)G / (double)nb;                              // rows per (big) band
            const double cb = 8.0 * ctx->cu_count / 8.0 / a.nstrips;             // bands one XCD has in flight (8 waves per CU)
            const int pct = kn.taper > 1 ? kn.taper : 100;                // (knob: the tapered part in % of one round)
            long long T = (long long)(cb * H * pct / 100.0 + 0.5);
            if (T > Lx / 2) T = Lx / 2;
            if (T >= (long long)(4 * H)) {
                const long long head = Lx - T, ta = T / 2, tb = T - ta;
                a.tn = 3;
                a.tstart[0] = 0;          a.tlen[0] = head; a.tnb[0] = (int)(head / H + 0.5);
                a.tstart[1] = head;       a.tlen[1] = ta;   a.tnb[1] = (int)(ta / (H / 2) + 0.5);
                a.tstart[2] = head + ta;  a.tlen[2] = tb;   a.tnb[2] = (int)(tb / (H / 4) + 0.5);
                a.tstart[3] = 0;          a.tlen[3] = 0;    a.tnb[3] = 1;
                for (int i = 0; i < 3; ++i) a.tnb[i] = a.tnb[i] < 1 ? 1 : a.tnb[i];
                a.bands_per_xcd = a.tnb[0] + a.tnb[1] + a.tnb[2];
                a.nbands = a.bands_per_xcd * 8;
            }
        }
    }
    a.shift = shift;
    a.acc_init = (int)(128 * ksum + (shift > 0 ? (1 << (shift - 1)) : 0));
    // Chained bands (k_filter_rows_chain): launches that fill the GPU, whole frames per XCD, the plain BGR instantiation.  Bands of
    // ~chain_rows rows (at least 8: an item must hold more pairs than the ring), items = (band, strip) drawn from per-XCD ticket counters.
    // (round 5: any frame count -- the batch's bands are dealt to the XCDs in eight contiguous runs.  Kernels with two weight tables stay on
    //  the one-band-per-wave kernel: chained, the 7x7 integer Gaussian measured 0.623 against 0.615 ms on 64 4K frames -- 13/8 of the matrix
    //  work per row, the launch is not memory-bound enough for the shorter bands to pay; profiles/r05_batch_cliffs.txt.  The measurement
    //  build keeps the chained two-table instantiations behind chain = 2.)
#ifdef RCV_ROWS_BENCH
    const bool chain_dual_ok = kn.chain == 2;
#else
    const bool chain_dual_ok = false;
#endif
    RCV_TRY(rcv_chain_poll(ctx));   // (a fault raised by an earlier chained launch of this context: reported before anything else is enqueued)
    if (src_yuyv == 0 && !sob && (dmask == 0 || chain_dual_ok) && (small_plan == 0 || kn.chain >= 1) && kn.band_rows == 0 && kn.chain != 0 && s.rows >= 64 &&
        ctx->cu_count == 256 && !ctx->fr_chain_off) {
        const int want_rows = kn.chain_rows > 0 ? (kn.chain_rows > 2048 ? 2048 : kn.chain_rows) : 32;   // (bmul < 2^32)
        int bpf = (s.rows + want_rows / 2) / want_rows;
        bpf = bpf < 1 ? 1 : (bpf > s.rows / 8 ? s.rows / 8 : bpf);
        const unsigned long long nbands = (unsigned long long)s.n * bpf, nitems = 4 * ((nbands + 7) / 8) * a.nstrips;   // (bound on the items of the longest run, all of it quartered)
        const int cwpc = kn.wpc == 12 || kn.wpc == 4 || kn.wpc == 6 || kn.wpc == 10 ? kn.wpc : 8;   // (knob: waves per CU, sweeps)
        const unsigned waves = (unsigned)(ctx->cu_count / 8 * cwpc);   // per XCD: cu_count / 8 CUs x 8 waves
        // (the kernel's quotients by multiply-and-shift are exact for these ranges)
        if (nbands >= 8 && nbands * (unsigned long long)(a.nstrips > bpf ? a.nstrips : bpf) < (1ull << 32) && nitems + waves < (1ull << 31) &&
            (unsigned long long)s.rows * d.step < (1ull << 32)) {
            a.bpf = bpf;
            a.cbands = (unsigned)nbands;
            a.bmul = (unsigned)((((unsigned long long)s.rows << 20) + bpf - 1) / bpf);
            // edge strips: strip 0 and every strip whose last window reaches past the row (the last one; the last two when the row ends
            // within 36 bytes of a strip seam)
            int n_edge = 0;
            for (int st = 0; st < a.nstrips; ++st)
                if (st == 0 || (long long)st * 768 + 804 > rb) ++n_edge;
            const int n_int = a.nstrips - n_edge;
            a.n_edge = n_edge;
            a.inv_edge = ((1ull << 32) + n_edge - 1) / n_edge;
            a.inv_int = n_int > 0 ? ((1ull << 32) + n_int - 1) / n_int : 0;
            a.inv_bpf = ((1ull << 32) + bpf - 1) / bpf;
            // waves that start on the edge queue: its share of the strips, weighted 1.15 (border repair, masked stores)
            {
                const double wgt = kn.edge_pct > 0 ? kn.edge_pct / 100.0 : 1.15;   // (tools/ablate_chain_edge.py: 80 ... 200 %, best 115)
                const double share = wgt * n_edge / (wgt * n_edge + n_int);
                int ew = (int)(share * waves + 0.5);
                a.edge_waves = n_int == 0 ? (int)waves : (ew < 1 ? 1 : ew);
            }
            if (!ctx->fr_fault) {
                RCV_HIP(hipHostMalloc((void**)&ctx->fr_fault, 64, hipHostMallocDefault));
                *ctx->fr_fault = 0u;
            }
            rcv_ctx::ChainLane& ln = ctx->fr_lane[kn.lane ? 1 : 0];
            uint8_t* const tk0 = ctx->kconst + (kn.lane ? RCV_KC_FR_TICKETS2 : RCV_KC_FR_TICKETS);
            const hipStream_t st = ctx->stream;   // (the second half of a split call runs with the context's two streams swapped: rcv_split_run)
            if (!ln.tickets_ready) {
                ln.unchecked = false;   // (the counters the check would read are about to be zeroed: a fault before this point was reported by the wait that cleared the flag)
                RCV_HIP(hipMemsetAsync(tk0, 0, RCV_KC_FR_TICKETS_BYTES, st));
                ln.seq = 0;
                ln.tickets_ready = true;
            }
            // tapered tail (kn.taper: -1 the product's plan, 0 none, else halved + 256 * quartered bands per XCD)
            {
                // Product plan: 8 bands halved + 4 quartered per XCD and queue (tools/chain_timeline.py, four boxes, 64 x 4K: the waves of an XCD
                // leave over ~21 us = one 32-row item without the taper, ~12-16 us with it; idle wave slots 2.9-3.5 % -> 2.1-2.8 %; back to
                // back -0.4 %, 16 frames -2.5 %; 16 + 8, 24 + 12, 0 + 8 within 0.2 % of it: profiles/r06_chain_timeline.txt), less on short runs
                const unsigned per_xcd = (unsigned)(nbands / 8);
                unsigned t1 = per_xcd / 8 < 8 ? per_xcd / 8 : 8, t2 = per_xcd / 16 < 4 ? per_xcd / 16 : 4;
                if (kn.taper >= 0) t1 = (unsigned)kn.taper & 255u, t2 = (unsigned)kn.taper >> 8;
                if (t1 + t2 > per_xcd / 2) t1 = t2 = 0;   // short runs: no taper
                // an item needs >= 7 rows (more row pairs than the ring's prefetch depth): halves of bands of 14+, quarters of bands of 28+ rows
                const int br_min = s.rows / bpf;
                if (br_min < 28) t2 = 0;
                if (br_min < 14) t1 = 0;
                a.tp1 = t1;
                a.tp2 = t2;
            }
            a.prev_tickets = ln.unchecked ? (const unsigned long long*)(tk0 + 2048 * ((ln.seq - 1u) & 3u)) : nullptr;
            a.prev_cbands = ln.prev[0]; a.prev_kint = ln.prev[1]; a.prev_kedge = ln.prev[2]; a.prev_tp1 = ln.prev[3]; a.prev_tp2 = ln.prev[4];
            a.fault = ctx->fr_fault;
            a.drop_xcd = rcv_knobs().fr_chain_drop_xcd;
            static_assert(RCV_KC_FR_TICKETS_BYTES == 4 * 2048, "four sets of 16 counters, 128 bytes apart");
            a.tickets = (unsigned long long*)(tk0 + 2048 * (ln.seq & 3u));
            a.tickets_next = (unsigned long long*)(tk0 + 2048 * ((ln.seq + 2u) & 3u));
            const dim3 grid(8u * waves);
            // EXACTLY 8 waves per CU, all resident from the start: the 7x7 instantiation's registers would let the dispatcher stack 12
            // waves on some CUs and leave others short.  An untouched dynamic-LDS request of an eighth of the CU's 160 KB caps it.
            const unsigned cap = (163840u / (unsigned)cwpc) & ~511u;
            constexpr int kAll7 = 255;
            (void)kAll7;
#ifdef RCV_ROWS_BENCH
            const bool centre7 = ksize == 7 && dmask != 0 && (dmask & ~kCentre7) == 0;
#endif
            if (ksize == 7) {
#ifdef RCV_ROWS_BENCH
                if ((kn.dbg & 255) == 4) RCV_LAUNCH((k_filter_rows_chain<7, 3, 256>), grid, dim3(64), cap, st, a);
                else if ((kn.dbg & 255) == 132) RCV_LAUNCH((k_filter_rows_chain<7, 3, 384>), grid, dim3(64), cap, st, a);
                else if ((kn.dbg & 255) == 128) RCV_LAUNCH((k_filter_rows_chain<7, 3, 128>), grid, dim3(64), cap, st, a);
                else if (kn.var == 1000) RCV_LAUNCH((k_filter_rows_chain<7, 3, 0, 0, 0>), grid, dim3(64), cap, st, a);   // (the round-5 form)
                else if (kn.var == 33) RCV_LAUNCH((k_filter_rows_chain<7, 3, 0, 0, 33>), grid, dim3(64), cap, st, a);   // wave timeline
                else if (kn.pp == 4) RCV_LAUNCH((k_filter_rows_chain<7, 4, 0>), grid, dim3(64), cap, st, a);
                else if (kn.pp == 2) RCV_LAUNCH((k_filter_rows_chain<7, 2, 0>), grid, dim3(64), cap, st, a);
                else
#endif
#ifdef RCV_ROWS_BENCH
                if (dmask != 0 && centre7) RCV_LAUNCH((k_filter_rows_chain<7, 3, 0, kCentre7>), grid, dim3(64), cap, st, a);
                else if (dmask != 0) RCV_LAUNCH((k_filter_rows_chain<7, 3, 0, kAll7>), grid, dim3(64), cap, st, a);
                else
#endif
                    RCV_LAUNCH((k_filter_rows_chain<7, 3, 0, 0, kChainVar>), grid, dim3(64), cap, st, a);
            } else if (dmask != 0) return RCV_ERR_UNSUPPORTED;   // (measurement build: two tables chained for 7x7 only)
            else if (ksize == 5) RCV_LAUNCH((k_filter_rows_chain<5, 3, 0, 0, kChainVar>), grid, dim3(64), cap, st, a);
            else RCV_LAUNCH((k_filter_rows_chain<3, 3, 0, 0, kChainVar>), grid, dim3(64), cap, st, a);
            const int rc = rcv_launch_check(ctx);
            if (rc == RCV_OK) {   // (a launch that did not start touched no counter: the same set serves the next one)
                ++ln.seq;
                ln.prev[0] = a.cbands; ln.prev[1] = (unsigned)n_int; ln.prev[2] = (unsigned)n_edge; ln.prev[3] = a.tp1; ln.prev[4] = a.tp2;
                ln.unchecked = true;
            } else {
                ln.tickets_ready = false;
            }
            return rc;
        }
    }
    if (kn.must_chain) return RCV_ERR_UNSUPPORTED;   // (the split call: its halves are chained launches or the call is not split)
    if ((long long)a.bands_per_xcd * a.nstrips * 8 > 0x3fffffffLL) return RCV_ERR_UNSUPPORTED;
    a.wpb = sob ? 4 : (kn.wpb == 2 || kn.wpb == 4 || kn.wpb == 8 ? kn.wpb : 1);   // (sob: a group's four waves are one workgroup)
    const unsigned ldsw = lds * (unsigned)a.wpb > 163840u ? 163840u : lds * (unsigned)a.wpb;   // (the occupancy cap is per workgroup)
    const int pp = kn.pp > 0 ? kn.pp : 3;
    if (ksize == 7) launch_rows<7>(a, pp, ldsw, dmask, src_yuyv, ctx->stream, kn.dbg);
    else if (ksize == 5) launch_rows<5>(a, pp, ldsw, dmask, src_yuyv, ctx->stream, kn.dbg);
    else launch_rows<3>(a, pp, ldsw, dmask, src_yuyv, ctx->stream, kn.dbg);
    return rcv_launch_check(ctx);
}
