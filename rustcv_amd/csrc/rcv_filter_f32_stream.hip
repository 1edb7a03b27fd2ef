// rcv_filter_f32_stream.hip -- f32-weight filter2D and GaussianBlur(sigma > 0) as a streaming VALU kernel.
//
// These two ops are VALU-bound by construction: the parity contract fixes one fmaf per tap in a fixed order
// (SURVEY.md 8-A), 49 dependent fmaf per sample for a dense 7x7 -- 49 flop per 2 algorithmic bytes against a chip
// balance of ~20 flop/B.  (The exact f32 MFMA would reproduce the chain bit for bit but a banded
// 16x16x4 operand wastes 15/22 of its MACs at the VECTOR rate, i.e. it is slower than the plain FMA loop.)
//
// One thread owns 4 adjacent byte columns (one dword of the interleaved row) and walks down a row segment.  Each
// source row is read once per thread as NW aligned dwords covering the +-rad pixel window, bytes go to f32 with
// v_cvt_f32_ubyteN, and every in-flight output (KS of them per column, one per kernel row) receives its fmaf in
// (ky, kx) order -- rows arrive in increasing order, so each accumulator sees exactly the oracle's sequence:
//     dense   : acc(y) = fmaf(k[ky][kx], p(y+ky-rad, x+kx-rad), acc)           KS*KS fmaf per sample
//     Gaussian: h(r,x) = fmaf chain over kx from 0 ; acc(y) = fmaf(t[ky], h(y+ky-rad, x), acc)   2*KS fmaf per sample
// The loop is unrolled by KS so accumulator slots (y mod KS) are static registers.  BORDER_REFLECT_101: rows by index;
// threads whose window leaves the row gather their bytes one by one with reflected pixel indices (first/last threads
// of a row only) -- done by a second, tiny launch of the generic kernel restricted to those byte columns.
#include "rcv_internal.h"
#include "rcv_kernels.h"
#include "rcv_device_utils.h"
#include <string.h>
#include <type_traits>
#include <utility>

#ifndef RCV_FS_SCHED
#define RCV_FS_SCHED 1   // scheduling barriers that keep the tap-major order of the fma chains
#endif

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kBlock = 256;
// This file is compiled a second time, with -DRCV_FS_BENCH, into librustcv_hip_bench.so: rcv__gauss_f32_bench runs the separable f32 launch with
// an untouched dynamic-LDS request per workgroup (an occupancy cap: what the pass gains from the waves it holds, tools/occupancy_f32.py) and with
// the row-pair kernel forced on or off.  The product passes 0 / its own rule.
#ifdef RCV_FS_BENCH
static unsigned g_fs_lds = 0;
static int g_fs_pairs = -1;
#else
constexpr unsigned g_fs_lds = 0;
constexpr int g_fs_pairs = -1;
#endif

// the row-pair kernel of the separable pass can be switched off for A/B tests (RCV_GAUSS_ROWS=0, the knob that also keeps small integer
// Gaussians off the register-window kernel)
static inline bool rcv_f32_pairs_off() { return rcv_knobs().gauss_rows == 0; }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Weights travel as kernel arguments and stay in SGPRs, two per 64-bit pair: v_pk_fma_f32 broadcasts either half of the pair
// to both of its lanes (op_sel / op_sel_hi), so a weight costs one SGPR, not a {w, w} pair (49 pairs would not fit).
template <int KS, bool SEP>
struct FWeights {
    f2 w2[((SEP ? KS : KS * KS) + 1) / 2];
    float delta;
    float iscale;   // != 0: INTEGER filter on this kernel -- integer weights, exact integer sums (< 2^23), result = (sum + half) >> shift
                    // formed as floor(sum * 2^-shift + 0.5) (iscale = 2^-shift; every step exact in f32), then saturated
};

// d = fma({w, w}, x, acc) with w = half HALF of the uniform pair `wp`
// (`half` is a compile-time constant after unrolling: the branch folds)
__device__ __forceinline__ f2 pk_fma_w(int half, f2 wp, f2 x, f2 acc)
{
    f2 d;
    if (half == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "s"(wp), "v"(x), "v"(acc));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "s"(wp), "v"(x), "v"(acc));
    return d;
}

// The NW window dwords of a thread as 16-byte loads (+ an 8- / 4-byte rest): the rows are 4-byte aligned, which is all a global load
// needs.  As NW separate dword loads (round 1-4) the window cost 8 address-path slots of ~13 cycles per wave and row -- as much
// as the row's arithmetic (tools/ubench_gather.hip: a wave-level load costs the same 13-16 cycles whether it moves 4 or 16 bytes
// per lane).
template <int NW>
__device__ __forceinline__ void load_window(const uint8_t* row, uint32_t* w)
{
    typedef uint32_t u4a __attribute__((ext_vector_type(4), aligned(4)));
    typedef uint32_t u2a __attribute__((ext_vector_type(2), aligned(4)));
    constexpr int N4 = NW / 4;
#pragma unroll
    for (int i = 0; i < N4; ++i) {
        const u4a v = *(const u4a*)(row + 16 * i);
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    if constexpr (NW % 4 >= 2) {
        const u2a v = *(const u2a*)(row + 16 * N4);
        w[4 * N4] = v.x; w[4 * N4 + 1] = v.y;
    }
    if constexpr (NW % 2 == 1) w[NW - 1] = *(const uint32_t*)(row + 4 * (NW - 1));
}

// EDGE = true is the same computation for the few threads per row whose byte window leaves the row: they gather their
// 2*LEAD+4 window bytes one by one at BORDER_REFLECT_101 positions (offsets fixed per thread, computed once) and then
// run the identical accumulation code -- so the border columns are bit-identical by construction and cost microseconds
// (a second launch of the generic per-sample kernel over those columns used to take as long as the main kernel).
// BPT = owned bytes per thread (4 or 8): with 8 the 2*LEAD halo conversions are shared by twice as many samples.
// RAG = true (BPT = 4): rows of ANY alignment and length (the reference's Mat::new gives step = cols * channels, so an odd
// width means byte-aligned rows): a row's misalignment is the same for every thread, so the window is fetched as the NW + 1
// ALIGNED dwords that contain it and shifted into place with v_alignbyte (unaligned per-lane loads would serialise in the
// address path); results go out as unaligned dword stores, the row's last partial dword byte by byte (EDGE instantiation).
template <int KS, int CH, bool SEP, bool EDGE, int BPT, bool RAG = false>
__global__ __launch_bounds__(kBlock) void k_filter_f32_stream(View s, View d, FWeights<KS, SEP> W, int seg_rows, int edge_nl, int edge_nr, int gx, int gy,
                                                              int nblocks, int blocks_per_xcd)
{
    // Block order (speed only): hardware places block b on XCD b % 8; with blocks_per_xcd > 0 (one-dimensional grid) every XCD works
    // through its own contiguous eighth of the (frame, row segment, column block) list, so that what ONE XCD has in flight is a
    // compact address range (DESIGN_HISTORY.md 6)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (blocks_per_xcd > 0) {
        const int tb = (int)(blockIdx.x & 7) * blocks_per_xcd + (int)(blockIdx.x >> 3);
        if (tb >= nblocks) return;
        bz = tb / (gx * gy);
        const int rem = tb - bz * gx * gy;
        by = rem / gx;
        bx = rem - by * gx;
    }
    constexpr int RAD = KS / 2;
    constexpr int LEAD = RAD * CH;                 // bytes of halo on each side of the 4 owned bytes
    constexpr int LEADW = (LEAD + 3) / 4 * 4;      // window starts LEADW bytes before the owned dword
    constexpr int OFF = LEADW - LEAD;              // first needed byte inside the window
    constexpr int NW = (LEADW + BPT + LEAD + 3) / 4; // window dwords
    constexpr int NB = 2 * LEAD + BPT;               // window bytes
    constexpr int NP = BPT / 2;                      // packed sample pairs per thread
    constexpr int NR = EDGE ? NB : (RAG ? NW + 2 : NW);   // registers per staged row (RAG: NW + 1 aligned dwords and the byte shift)
    const int rowbytes = s.cols * CH;
    int t = bx * (int)blockDim.x + (int)threadIdx.x;
    if (EDGE) {
        if (t >= edge_nl + edge_nr) return;
        if (t >= edge_nl) t = (rowbytes + BPT - 1) / BPT - edge_nr + (t - edge_nl);
    }
    const int xb0 = BPT * t;
    if (xb0 >= rowbytes) return;
    if (RAG && !EDGE && xb0 + BPT > rowbytes) return;   // the row's last, partial dword belongs to the EDGE launch
    const int ys = by * seg_rows, ye = min(s.rows, ys + seg_rows);
    const uint8_t* sf = s.p + (size_t)bz * s.fstride;
    uint8_t* df = d.p + (size_t)bz * d.fstride + xb0;
    // The window is clamped into the row, so the first/last few threads of a row compute garbage: the host
    // re-does exactly those byte columns with the generic kernel right after this launch (no divergent slow path here).
    const int wstart = EDGE ? 0 : min(max(xb0 - LEADW, 0), rowbytes - 4 * NW);
    int goff[EDGE ? NB : 1];                       // EDGE: row byte offset of window byte b (pixel index reflected, channel kept)
    if (EDGE) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int gi = xb0 - LEAD + b;                              // interleaved sample index, may be < 0 or >= rowbytes
            const int px = gi >= 0 ? gi / CH : -((-gi + CH - 1) / CH);  // floor
            goff[b] = rcv_reflect101(px, s.cols) * CH + (gi - px * CH);
        }
    }

    auto load_row = [&](int ry, uint32_t (&w)[NR]) __attribute__((always_inline)) {
        ry = min(ry, ye - 1 + RAD);
        const int r = rcv_reflect101(ry, s.rows);
        const uint8_t* row = sf + (size_t)r * s.step + wstart;
        if constexpr (EDGE) {
#pragma unroll
            for (int b = 0; b < NB; ++b) w[b] = row[goff[b]];
        } else if constexpr (RAG) {
            const unsigned mis = (unsigned)((uintptr_t)row & 3);
            const uint32_t* base = (const uint32_t*)(row - mis);
#pragma unroll
            for (int i = 0; i < NW; ++i) w[i] = base[i];
            w[NW] = base[mis ? NW : NW - 1];   // (an aligned row needs no further dword: never read past the window then)
            w[NW + 1] = mis;
        } else {
            load_window<NW>(row, w);
        }
    };

    // The four samples of a thread ride in two packed-f32 pairs (samples 0,1 and 2,3): v_pk_fma_f32 performs two IEEE
    // fmaf per instruction with the weight broadcast to both halves -- the same per-sample chain as the scalar oracle, at
    // half the issue cost.  Sample j of tap kx reads p[j + kx*CH]; the pair {p[m], p[m+1]} is formed by the compiler
    // (two v_mov_b32 when m is odd.  Round 6: written as vector shuffles of even-aligned pairs every odd pair is ONE v_pk_mov_b32 -- 960 -> 911
    //  instructions in the 7-tap BGR instantiation -- and the launches were 1 % (7-tap Gaussian) to 6.5 % (dense 7x7 f32) SLOWER, same box, three
    //  rotations: the packed move does not issue at the rate of a plain one.  Not kept; tools/ab_fs_variants.sh.)
    f2 acc[KS][NP];
#pragma unroll
    for (int i = 0; i < KS; ++i)
#pragma unroll
        for (int j = 0; j < NP; ++j) acc[i][j] = SEP ? f2{0.0f, 0.0f} : f2{W.delta, W.delta};

    // feed row r (r mod KS == RHO, static): contributes kernel row ky to output y = r - ky + RAD, slot y mod KS
    auto feed = [&](const uint32_t (&w)[NR], int r, auto rho_tag) __attribute__((always_inline)) {
        constexpr int RHO = decltype(rho_tag)::value;
        float p[NB];
        uint32_t wa[EDGE ? 1 : NW];   // the window's dwords
        if constexpr (!EDGE) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                if constexpr (RAG) wa[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], w[NW + 1]);
                else wa[i] = w[i];
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) p[b] = EDGE ? (float)w[b] : (float)((wa[(OFF + b) >> 2] >> (((OFF + b) & 3) * 8)) & 0xff);
        f2 h[NP];
        if (SEP) {   // (tap-major: NP independent chains side by side)
#pragma unroll
            for (int jj = 0; jj < NP; ++jj) h[jj] = f2{0.0f, 0.0f};
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
                for (int jj = 0; jj < NP; ++jj) h[jj] = pk_fma_w(kx & 1, W.w2[kx >> 1], f2{p[2 * jj + kx * CH], p[2 * jj + 1 + kx * CH]}, h[jj]);
                if (RCV_FS_SCHED) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (SEP) {
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int slot = ((RHO - ky + RAD) % KS + KS) % KS;
#pragma unroll
                for (int jj = 0; jj < NP; ++jj) acc[slot][jj] = pk_fma_w(ky & 1, W.w2[ky >> 1], h[jj], acc[slot][jj]);
            }
        } else {
            // dense: tap-major as well -- for one kx the KS output slots x NP pairs are independent accumulators (each still sees its own
            // taps in (ky, kx) order: a slot takes one ky per source row), instead of KS dependent fmas in a row per accumulator
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) {
                    const int slot = ((RHO - ky + RAD) % KS + KS) % KS;
#pragma unroll
                    for (int jj = 0; jj < NP; ++jj)
                        acc[slot][jj] = pk_fma_w((ky * KS + kx) & 1, W.w2[(ky * KS + kx) >> 1], f2{p[2 * jj + kx * CH], p[2 * jj + 1 + kx * CH]}, acc[slot][jj]);
                }
                if (RCV_FS_SCHED) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the output whose last kernel row (ky = KS-1) was just applied: y = r - RAD, slot (RHO + RAD + 1) % KS
        constexpr int done = ((RHO - (KS - 1) + RAD) % KS + KS) % KS;
        const int y = r - RAD;
        // v_cvt_pk_u8_f32 itself rounds half to even (measured on gfx950: 0.5 -> 0, 1.5 -> 2, 2.5 -> 2, 254.5 -> 254),
        // saturates to [0, 255] (NaN -> 0) and packs: saturate(rint(v)) of the specification in one instruction per sample
        if (W.iscale != 0.0f) {   // (uniform) integer mode: floor((sum + half) / 2^shift), exact
#pragma unroll
            for (int j = 0; j < NP; ++j)
                acc[done][j] = __builtin_elementwise_floor(__builtin_elementwise_fma(acc[done][j], f2{W.iscale, W.iscale}, f2{0.5f, 0.5f}));
        }
        uint32_t o[BPT / 4];
#pragma unroll
        for (int q = 0; q < BPT / 4; ++q) {
            uint32_t v = __builtin_amdgcn_cvt_pk_u8_f32(acc[done][2 * q].x, 0, 0u);
            v = __builtin_amdgcn_cvt_pk_u8_f32(acc[done][2 * q].y, 1, v);
            v = __builtin_amdgcn_cvt_pk_u8_f32(acc[done][2 * q + 1].x, 2, v);
            o[q] = __builtin_amdgcn_cvt_pk_u8_f32(acc[done][2 * q + 1].y, 3, v);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) acc[done][j] = SEP ? f2{0.0f, 0.0f} : f2{W.delta, W.delta};
        if (y >= ys && y < ye) {
            if constexpr (BPT == 8) *(uint2*)(df + (size_t)y * d.step) = make_uint2(o[0], o[1]);
            else if constexpr (RAG) {
                uint8_t* q = df + (size_t)y * d.step;
                if (xb0 + 4 <= rowbytes) {
                    typedef uint32_t u1m __attribute__((aligned(1)));
                    *(u1m*)q = o[0];
                } else {   // the row's last 1..3 bytes (EDGE launch only)
                    for (int b = 0; b < rowbytes - xb0; ++b) q[b] = (uint8_t)(o[0] >> (8 * b));
                }
            } else *(uint32_t*)(df + (size_t)y * d.step) = o[0];
        }
    };

    // rows ys-RAD .. ye-1+RAD; the unrolled body handles KS consecutive rows whose (row mod KS) is static when the
    // stream starts at a multiple of KS: start at r0 = floor((ys - RAD) / KS) * KS (the extra leading rows only touch
    // outputs above the segment, which are never stored).
    int r0 = ys - RAD;
    r0 = r0 >= 0 ? r0 / KS * KS : -((-r0 + KS - 1) / KS) * KS;
    // AH rows are in flight while one is consumed: two for the separable passes (the 7-tap Gaussian 1.05 -> 0.95 ms on 64 4K
    // frames, 9 taps 1.33 -> 1.24, one-channel 7 taps 0.38 -> 0.33), one for the dense kernels (VALU-bound: a second row in
    // flight costs registers and time, 7x7 1.88 -> 2.1 ms).  The AH + 1 row buffers take their roles by static index inside the
    // unrolled block of KS rows, so rows are not copied from buffer to buffer; after the block the live ones move back to the
    // canonical places (once per KS rows, and not at all when KS is a multiple of AH + 1).
#ifndef RCV_FS_AHEAD
#define RCV_FS_AHEAD (SEP ? 2 : 1)
#endif
    constexpr int AH = RCV_FS_AHEAD, NBUF = AH + 1;
    uint32_t B[NBUF][NR];
#pragma unroll
    for (int i = 0; i < AH; ++i) load_row(r0 + i, B[i]);
    for (int rb = r0; rb <= ye - 1 + RAD; rb += KS) {
        static_for<0, KS>([&](auto I) __attribute__((always_inline)) {
            constexpr int c = decltype(I)::value % NBUF, l = (decltype(I)::value + AH) % NBUF;
            load_row(rb + I + AH, B[l]);
            feed(B[c], rb + I, I);
        });
        constexpr int sh = KS % NBUF;   // the live rows sit in B[(sh + j) % NBUF], j = 0 .. AH-1
        if constexpr (sh != 0) {
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                if constexpr (NBUF == 2) B[0][q] = B[1][q];
                else if constexpr (sh == 1) {
                    B[0][q] = B[1][q];
                    B[1][q] = B[2][q];
                } else {
                    B[1][q] = B[0][q];
                    B[0][q] = B[2][q];
                }
            }
        }
    }
}

// ---- the separable pass on ROW PAIRS (round 5) ---------------------------------------------------------------------------------
// k_filter_f32_stream pairs ADJACENT SAMPLES of one row in a packed-f32 register: tap kx of the pair starting at sample m reads
// {p[m + kx CH], p[m + 1 + kx CH]}, which for three channels is an even-aligned register pair only for every other kx -- the rest are
// rebuilt with moves, and the converted window is per row.  Here the two halves of a pair are the SAME sample of two consecutive
// rows: P2[b] = {row r byte b, row r + 1 byte b} (each half one v_cvt_f32_ubyteN), every tap index is an aligned pair, the
// horizontal pass gives he = {h(r), h(r + 1)} per column, and the vertical pass accumulates OUTPUT row pairs (y, y + 1):
//     acc(y, y + 1) += t[ky] * {h(y + ky - RAD), h(y + 1 + ky - RAD)}
// -- the pair that starts on the row pair's first row is he itself, the one that starts a row earlier is ho = {previous he.y, he.x}
// (one v_pk_mov per column and row pair).  Every output sees its taps in ky order (within a row pair: ho, one ky lower, before he),
// each chain starts with a multiply (= fma(w, x, 0) for everything a u8 store can tell apart), so the result is the oracle's, bit for
// bit.  RAD + 1 output pairs are in flight (static slots: the loop is unrolled by RAD + 1 row pairs).  Per sample: 3.25 conversions +
// 2 x KS / 2 packed fma + 0.5 move + the store conversion -- 12.3 instructions for 7 taps of BGR against ~15.4.
template <int KS, int CH>
__global__ __launch_bounds__(kBlock) void k_gauss_f32_pairs(View s, View d, FWeights<KS, true> W, int seg_rows, int gx, int gy, int nblocks, int blocks_per_xcd)
{
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (blocks_per_xcd > 0) {
        const int tb = (int)(blockIdx.x & 7) * blocks_per_xcd + (int)(blockIdx.x >> 3);
        if (tb >= nblocks) return;
        bz = tb / (gx * gy);
        const int rem = tb - bz * gx * gy;
        by = rem / gx;
        bx = rem - by * gx;
    }
    constexpr int BPT = 8, RAD = KS / 2, LEAD = RAD * CH, LEADW = (LEAD + 3) / 4 * 4, OFF = LEADW - LEAD, NW = (LEADW + BPT + LEAD + 3) / 4, NB = 2 * LEAD + BPT;
    constexpr int NS = RAD + 1;   // output row pairs in flight
    const int rowbytes = s.cols * CH;
    const int xb0 = BPT * (bx * (int)blockDim.x + (int)threadIdx.x);
    if (xb0 >= rowbytes) return;
    const int ys = by * seg_rows, ye = min(s.rows, ys + seg_rows);
    const uint8_t* sf = s.p + (size_t)bz * s.fstride;
    uint8_t* df = d.p + (size_t)bz * d.fstride + xb0;
    const int wstart = min(max(xb0 - LEADW, 0), rowbytes - 4 * NW);   // (threads whose window leaves the row: redone by the EDGE launch)
    auto load_row = [&](int ry, uint32_t (&w)[NW]) __attribute__((always_inline)) {
        ry = min(ry, ye - 1 + RAD);
        const uint8_t* row = sf + (size_t)rcv_reflect101(ry, s.rows) * s.step + wstart;
        load_window<NW>(row, w);
    };
    auto byte_f = [](const uint32_t (&w)[NW], int b) __attribute__((always_inline)) { return (float)((w[(OFF + b) >> 2] >> (((OFF + b) & 3) * 8)) & 0xff); };
    f2 acc[NS][BPT];
    float hp[BPT];   // h of the row before the current pair
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
        hp[j] = 0.0f;
#pragma unroll
        for (int q = 0; q < NS; ++q) acc[q][j] = f2{0.0f, 0.0f};
    }
    const int r0 = ys - RAD;   // the stream's first row; output pairs are (r0 + 2q, r0 + 2q + 1)
    auto feed = [&](const uint32_t (&wa)[NW], const uint32_t (&wb)[NW], int R, auto pi_tag) __attribute__((always_inline)) {
        constexpr int PI = decltype(pi_tag)::value;   // (row pair index) mod NS
        f2 P2[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) P2[b] = f2{byte_f(wa, b), byte_f(wb, b)};
        f2 he[BPT], ho[BPT];
        // (tap-major: consecutive instructions belong to different columns -- eight independent chains instead of one seven deep)
#pragma unroll
        for (int j = 0; j < BPT; ++j) he[j] = P2[j] * f2{W.w2[0][0], W.w2[0][0]};
#pragma unroll
        for (int kx = 1; kx < KS; ++kx) {
#pragma unroll
            for (int j = 0; j < BPT; ++j) he[j] = pk_fma_w(kx & 1, W.w2[kx >> 1], P2[j + kx * CH], he[j]);
            __builtin_amdgcn_sched_barrier(0);   // (keeps the tap-major order: without it the compiler re-serialises the chains, +3 % at 7 taps)
        }
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            ho[j] = f2{hp[j], he[j].x};
            hp[j] = he[j].y;
        }
        // vertical taps: output pair (row pair index + dq) gets ky = RAD - 1 - 2 dq from ho and ky = RAD - 2 dq from he
#pragma unroll
        for (int dq = RAD; dq >= -RAD - 1; --dq) {
            constexpr int dummy = 0;
            (void)dummy;
            const int kyo = RAD - 1 - 2 * dq, kye = RAD - 2 * dq;
            const int slot = ((PI + dq) % NS + NS) % NS;
            if (kyo >= 0 && kyo < KS) {
#pragma unroll
                for (int j = 0; j < BPT; ++j)
                    acc[slot][j] = kyo == 0 ? ho[j] * f2{W.w2[0][0], W.w2[0][0]} : pk_fma_w(kyo & 1, W.w2[kyo >> 1], ho[j], acc[slot][j]);
            }
            if (kye >= 0 && kye < KS) {
#pragma unroll
                for (int j = 0; j < BPT; ++j)
                    acc[slot][j] = kye == 0 ? he[j] * f2{W.w2[0][0], W.w2[0][0]} : pk_fma_w(kye & 1, W.w2[kye >> 1], he[j], acc[slot][j]);
            }
            if (kyo == KS - 1 || kye == KS - 1) {   // this output pair is complete: rows y, y + 1
                const int y = (kyo == KS - 1 ? R - 1 : R) - RAD;
                if (W.iscale != 0.0f) {   // (uniform) integer mode: floor((sum + half) / 2^shift), exact
#pragma unroll
                    for (int j = 0; j < BPT; ++j)
                        acc[slot][j] = __builtin_elementwise_floor(__builtin_elementwise_fma(acc[slot][j], f2{W.iscale, W.iscale}, f2{0.5f, 0.5f}));
                }
                uint32_t o0[2], o1[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    uint32_t v = __builtin_amdgcn_cvt_pk_u8_f32(acc[slot][4 * q].x, 0, 0u);
                    v = __builtin_amdgcn_cvt_pk_u8_f32(acc[slot][4 * q + 1].x, 1, v);
                    v = __builtin_amdgcn_cvt_pk_u8_f32(acc[slot][4 * q + 2].x, 2, v);
                    o0[q] = __builtin_amdgcn_cvt_pk_u8_f32(acc[slot][4 * q + 3].x, 3, v);
                    uint32_t u = __builtin_amdgcn_cvt_pk_u8_f32(acc[slot][4 * q].y, 0, 0u);
                    u = __builtin_amdgcn_cvt_pk_u8_f32(acc[slot][4 * q + 1].y, 1, u);
                    u = __builtin_amdgcn_cvt_pk_u8_f32(acc[slot][4 * q + 2].y, 2, u);
                    o1[q] = __builtin_amdgcn_cvt_pk_u8_f32(acc[slot][4 * q + 3].y, 3, u);
                }
                if (y >= ys && y < ye) *(uint2*)(df + (size_t)y * d.step) = make_uint2(o0[0], o0[1]);
                if (y + 1 >= ys && y + 1 < ye) *(uint2*)(df + (size_t)(y + 1) * d.step) = make_uint2(o1[0], o1[1]);
            }
        }
    };
    uint32_t A0[NW], A1[NW], B0[NW], B1[NW];
    load_row(r0, A0);
    load_row(r0 + 1, A1);
    // the last output row ye - 1 completes with the row pair that holds row ye - 1 + RAD (he) or ye + RAD (ho)
    for (int R = r0; R <= ye + RAD; R += 2 * NS) {
        static_for<0, NS>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if constexpr (i % 2 == 0) {
                load_row(R + 2 * i + 2, B0);
                load_row(R + 2 * i + 3, B1);
                feed(A0, A1, R + 2 * i, I);
            } else {
                load_row(R + 2 * i + 2, A0);
                load_row(R + 2 * i + 3, A1);
                feed(B0, B1, R + 2 * i, I);
            }
        });
        if constexpr (NS % 2 == 1) {   // an odd number of row pairs per block: the pair in flight sits in B
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                A0[q] = B0[q];
                A1[q] = B1[q];
            }
        }
    }
}

template <int KS, int CH, bool SEP, int BPT, bool RAG = false>
int launch(rcv_ctx* ctx, const View& s, const View& d, const float* w, float delta, float iscale)
{
    constexpr int RAD = KS / 2, LEAD = RAD * CH, LEADW = (LEAD + 3) / 4 * 4, NW = (LEADW + BPT + LEAD + 3) / 4;
    const int rowbytes = s.cols * CH;
    if (rowbytes < 4 * NW + 4 || (!RAG && rowbytes % BPT != 0)) return RCV_ERR_UNSUPPORTED;
    const int nthreads = (rowbytes + BPT - 1) / BPT;   // threads per row (RAG: the last one may own fewer than BPT bytes)
    FWeights<KS, SEP> W;
    memset(&W, 0, sizeof(W));
    for (int i = 0; i < (SEP ? KS : KS * KS); ++i) W.w2[i >> 1][i & 1] = w[i];
    W.delta = delta;
    W.iscale = iscale;
    const unsigned gx = (unsigned)((nthreads + kBlock - 1) / kBlock);
    int seg = s.rows;
    // row segments: enough workgroups to fill the GPU a few times, as few as that allows -- every segment re-reads KS - 1 halo rows and
    // starts its stream at a multiple of KS (up to KS - 1 more).  Measured on 16 / 64 4K and 64 1080p frames: 7 taps 0.307 / 0.940 /
    // 0.326 ms at 4096 workgroups, 0.287 / 0.932 / 0.306 at 2048-3072; 3 taps the other way (0.758 -> 0.736 at 8192)
    constexpr int kMinBlocks = KS <= 3 ? 8192 : 2560;
    while ((long long)gx * ((s.rows + seg - 1) / seg) * s.n < kMinBlocks && seg > 8 * KS) seg = (seg + 1) / 2;
    {   // small launches: every SIMD one wave, as short as the halo allows (rcv_plan_seg_rows; 2 * (KS - 1) halo / stream-start rows)
        const int small = rcv_plan_seg_rows(s.rows, 4LL * gx * s.n, ctx->cu_count, 2 * (KS - 1) + 4, 2 * KS);
        if (small > 0 && small < seg) seg = small;
    }
    const unsigned gy = (unsigned)((s.rows + seg - 1) / seg);
    {
        const long long nb = (long long)gx * gy * s.n;
        const bool xcd = nb < (1LL << 30);
        const int bpx = xcd ? (int)((nb + 7) / 8) : 0;
        const dim3 grid = xcd ? dim3((unsigned)bpx * 8u) : dim3(gx, gy, s.n);
        if constexpr (SEP && BPT == 8 && !RAG) {
            // where it measures faster than the one-row kernel in its tap-major form (tools/ab_fs_variants.sh, same call, 64 x 4K BGR:
            // 3 taps 0.640 against 0.665 ms; 5 / 7 taps equal; 11 taps 3.5 % slower; one channel never); RCV_GAUSS_ROWS=1 (tests)
            // sends every shape here
            const bool pairs = g_fs_pairs >= 0 ? g_fs_pairs != 0 : (rcv_knobs().gauss_rows == 1 || (CH == 3 && KS == 3));
            if (g_fs_lds > 65536u) {   // (measurement build: more than the default 64 KB of dynamic LDS needs the attribute)
                (void)hipFuncSetAttribute((const void*)k_gauss_f32_pairs<KS, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_fs_lds);
                (void)hipFuncSetAttribute((const void*)k_filter_f32_stream<KS, CH, SEP, false, BPT, RAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_fs_lds);
            }
            if (pairs && (g_fs_pairs >= 0 || !rcv_f32_pairs_off())) RCV_LAUNCH((k_gauss_f32_pairs<KS, CH>), grid, dim3(kBlock), g_fs_lds, ctx->stream, s, d, W, seg, (int)gx, (int)gy, (int)nb, bpx);
            else RCV_LAUNCH((k_filter_f32_stream<KS, CH, SEP, false, BPT, RAG>), grid, dim3(kBlock), g_fs_lds, ctx->stream, s, d, W, seg, 0, 0, (int)gx, (int)gy, (int)nb, bpx);
        } else
        RCV_LAUNCH((k_filter_f32_stream<KS, CH, SEP, false, BPT, RAG>), grid, dim3(kBlock), 0, ctx->stream, s, d, W, seg, 0, 0, (int)gx, (int)gy, (int)nb, bpx);
    }
    RCV_TRY(rcv_launch_check(ctx));
    // threads whose window [xb0 - LEADW, xb0 - LEADW + 4 NW) left the row computed garbage: the first nl and the last nr of a
    // row -- redone by the EDGE instantiation (one wave per row segment: rows are short work, so use many small segments)
    const int limit = rowbytes - 4 * NW + LEADW;                  // last xb0 whose window still fits
    const int hi_begin = (limit / BPT + 1) * BPT;
    const int nl = min((LEADW + BPT - 1) / BPT, nthreads), nr = max(0, min(nthreads - hi_begin / BPT, nthreads - nl));
    const int eseg = 4 * KS < 32 ? 32 : 4 * KS;
    RCV_LAUNCH((k_filter_f32_stream<KS, CH, SEP, true, BPT, RAG>), dim3((unsigned)((nl + nr + 63) / 64), (unsigned)((s.rows + eseg - 1) / eseg), s.n),
                       dim3(64), 0, ctx->stream, s, d, W, eseg, nl, nr, 0, 0, 0, 0);
    return rcv_launch_check(ctx);
}

template <bool SEP>
int dispatch(rcv_ctx* ctx, const View& s, const View& d, const float* w, int ksize, float delta, float iscale = 0.0f)
{
    // rows that are not 4-byte aligned or whose length is not a multiple of 4 (odd widths of packed images): RAG instantiation
    const bool rag = (s.cols * s.ch) % 4 != 0 || (uintptr_t)s.p % 4 || s.step % 4 || (s.n > 1 && s.fstride % 4) || (uintptr_t)d.p % 4 || d.step % 4 ||
                     (d.n > 1 && d.fstride % 4);
    if (rag) {
#define RCV_CASE_RAG(KS, CH) \
    if (ksize == KS && s.ch == CH) return launch<KS, CH, SEP, 4, true>(ctx, s, d, w, delta, iscale);
        RCV_CASE_RAG(3, 1) RCV_CASE_RAG(5, 1) RCV_CASE_RAG(7, 1) RCV_CASE_RAG(3, 3) RCV_CASE_RAG(5, 3) RCV_CASE_RAG(7, 3)
#undef RCV_CASE_RAG
        return RCV_ERR_UNSUPPORTED;
    }
    // 8 bytes per thread where rows and row ends are 8-byte aligned (the halo conversions are shared by twice the samples)
    const bool wide = (s.cols * s.ch) % 8 == 0 && (uintptr_t)s.p % 8 == 0 && s.step % 8 == 0 && (s.n <= 1 || s.fstride % 8 == 0) &&
                      (uintptr_t)d.p % 8 == 0 && d.step % 8 == 0 && (d.n <= 1 || d.fstride % 8 == 0) &&
                      // (small launches -- at most ~2 rows of 8-byte threads per SIMD lane -- are latency-bound: twice the threads, half the work per row)
                      ((long long)s.rows * s.n * ((s.cols * s.ch + 7) / 8) > 128LL * 64 * 4 * ctx->cu_count ||   // (one or two 4K frames, four 1080p: -15 % with 4-byte threads)
                       (SEP && rcv_knobs().gauss_rows == 1));   // (tests: the row-pair kernel on small images)
#define RCV_CASE(KS, CH)                                                                 \
    if (ksize == KS && s.ch == CH) {                                                     \
        if (wide) {                                                                      \
            int rc = launch<KS, CH, SEP, 8>(ctx, s, d, w, delta, iscale);                \
            if (rc != RCV_ERR_UNSUPPORTED) return rc;                                    \
        }                                                                                \
        return launch<KS, CH, SEP, 4>(ctx, s, d, w, delta, iscale);                      \
    }
    RCV_CASE(3, 1) RCV_CASE(5, 1) RCV_CASE(7, 1) RCV_CASE(3, 3) RCV_CASE(5, 3) RCV_CASE(7, 3)
    if constexpr (SEP) { RCV_CASE(9, 1) RCV_CASE(11, 1) RCV_CASE(9, 3) RCV_CASE(11, 3) }
#undef RCV_CASE
    return RCV_ERR_UNSUPPORTED;
}

} // namespace

#ifdef RCV_FS_BENCH
// Measurement entry: GaussianBlur sigma > 0 of a device-resident batch; lds_bytes = untouched dynamic LDS per workgroup of four waves (0 = none),
// pairs = -1 the product's kernel choice, 0 the one-row kernel, 1 the row-pair kernel
extern "C" int rcv__gauss_f32_bench(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst, int ksize, double sigma, unsigned lds_bytes, int pairs)
{
    RCV_TRY(rcv_bind(ctx));
    if (!src || !dst || !(sigma > 0.0) || !(ksize & 1) || ksize < 3 || ksize > 11) return RCV_ERR_ARG;
    View s, d;
    RCV_TRY(rcv_view_batch(src, RCV_8U, &s));
    RCV_TRY(rcv_view_batch(dst, RCV_8U, &d));
    if (s.rows != d.rows || s.cols != d.cols || s.n != d.n || s.ch != d.ch) return RCV_ERR_ARG;
    float taps[32];
    RCV_TRY(rcv_gaussian_taps_f32(ksize, sigma, taps));
    g_fs_lds = lds_bytes;
    g_fs_pairs = pairs;
    const int rc = dispatch<true>(ctx, s, d, taps, ksize, 0.0f);
    g_fs_lds = 0;
    g_fs_pairs = -1;
    return rc;
}
#else
int rcv_filter_f32_fast(rcv_ctx* ctx, const View& s, const View& d, const float* k, int ksize, float delta)
{
    return dispatch<false>(ctx, s, d, k, ksize, delta);
}

int rcv_gauss_f32_fast(rcv_ctx* ctx, const View& s, const View& d, const float* taps, int ksize)
{
    return dispatch<true>(ctx, s, d, taps, ksize, 0.0f);
}

// Integer filter2D / integer GaussianBlur on the streaming kernel, for shapes the MFMA strip kernel does not take (rows that
// are only 4-byte aligned, widths that are not a multiple of 16 -- a packed 1080-pixel-wide BGR image): integer weights and
// u8 samples give integer partial sums, exact in f32 while sum(|k|) * 255 < 2^23, and the rounding shift is exact as well.
int rcv_filter_i16_stream(rcv_ctx* ctx, const View& s, const View& d, const int16_t* k, int ksize, int shift)
{
    if (ksize != 3 && ksize != 5 && ksize != 7) return RCV_ERR_UNSUPPORTED;
    long long mag = 0;
    float w[49];
    for (int i = 0; i < ksize * ksize; ++i) {
        mag += k[i] < 0 ? -k[i] : k[i];
        w[i] = (float)k[i];
    }
    if (mag * 255 >= (1 << 23) || shift < 0 || shift > 24) return RCV_ERR_UNSUPPORTED;
    return dispatch<false>(ctx, s, d, w, ksize, 0.0f, ldexpf(1.0f, -shift));
}

// integer GaussianBlur: taps t (sum 2^(shift/2)) applied separably -- the exact integer sum of the 2-D kernel t x t
int rcv_gauss_int_stream(rcv_ctx* ctx, const View& s, const View& d, const int* t, int ksize, int shift)
{
    if (ksize != 3 && ksize != 5 && ksize != 7) return RCV_ERR_UNSUPPORTED;
    long long sum = 0;
    float w[7];
    for (int i = 0; i < ksize; ++i) {
        if (t[i] < 0) return RCV_ERR_UNSUPPORTED;
        sum += t[i];
        w[i] = (float)t[i];
    }
    if (sum * sum * 255 >= (1 << 23)) return RCV_ERR_UNSUPPORTED;
    return dispatch<true>(ctx, s, d, w, ksize, 0.0f, ldexpf(1.0f, -shift));
}
#endif   // RCV_FS_BENCH
