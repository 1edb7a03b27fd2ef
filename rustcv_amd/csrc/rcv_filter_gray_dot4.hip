// rcv_filter_gray_dot4.hip -- integer filter2D / GaussianBlur(sigma <= 0) on ONE-channel images as a streaming VALU kernel.
//
// A gray 7x7 needs 49 MAC per pixel against 2 algorithmic bytes: 16 times fewer per byte of traffic than the BGR case that
// needs the matrix cores (rcv_filter7_mfma.hip), and within reach of v_dot4c_i32_i8 -- two instructions per kernel row and
// pixel.  Structure of rcv_filter_f32_stream.hip: one thread owns 4 adjacent pixels and walks down a row segment; each
// source row is read once as the three aligned dwords that contain the +-3 pixel window, xor 0x80 turns u8 into the
// signed operand (the accumulators start at 128*sum(K) + round, as in the MFMA kernel), v_alignbyte forms the two 4-byte
// operands per pixel ONCE per source row, and every in-flight output (KS of them, one per kernel row) receives its dot
// products; the loop is unrolled by KS so accumulator slots are static registers.  Integer arithmetic: exact in any order.
// Weights beyond the i8 range (integer Gaussian 7x7) are split K = 4Q + R (two dot chains, acc + (accQ << 2)).
// BORDER_REFLECT_101: rows by index; the first and the last threads of a row, whose window leaves the row, are redone by
// the EDGE instantiation with gathered, reflected window bytes and the identical accumulation code.
#include "rcv_internal.h"
#include "rcv_kernels.h"
#include "rcv_device_utils.h"
#include <string.h>
#include <type_traits>
#include <utility>

namespace {

constexpr int kBlock = 256;

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int KS>
struct GrayW {
    int r[KS][2];   // packed i8 weights of kernel row ky: taps 0..3 | taps 4..7 (zero padded, centred by RAD)
    int q[KS][2];   // DUAL: the 4Q part
    int acc_init, shift;
};

// RAG: rows of any alignment and any width >= 12 (an odd width of a packed image: step = cols): the row's misalignment is the same
// for every thread, so the 12-byte window is fetched as the four ALIGNED dwords that contain it and shifted into place with
// v_alignbyte; unaligned dword stores; the row's last 1-3 pixels byte by byte (EDGE launch).
template <int KS, bool DUAL, bool EDGE, bool RAG = false>
__global__ __launch_bounds__(kBlock) void k_filter_gray_dot4(View s, View d, GrayW<KS> W, int seg_rows, int edge_nl, int edge_nr)
{
    constexpr int RAD = KS / 2;
    constexpr int ND = KS == 3 ? 1 : 2;            // operand dwords per pixel and kernel row
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (EDGE) {
        if (t >= edge_nl + edge_nr) return;
        if (t >= edge_nl) t = (s.cols + 3) / 4 - edge_nr + (t - edge_nl);
    }
    const int xb0 = 4 * t;
    if (xb0 >= s.cols) return;
    if (RAG && !EDGE && xb0 + 4 > s.cols) return;   // the row's last, partial group belongs to the EDGE launch
    const int ys = blockIdx.y * seg_rows, ye = min(s.rows, ys + seg_rows);
    const uint8_t* sf = s.p + (size_t)blockIdx.z * s.fstride;
    uint8_t* df = d.p + (size_t)blockIdx.z * d.fstride + xb0;
    // window = the 12 bytes [xb0 - 4, xb0 + 8); clamped into the row for the (few) threads the EDGE launch redoes
    const int wstart = min(max(xb0 - 4, 0), s.cols - 12);
    int goff[EDGE ? 12 : 1];
    if (EDGE) {
#pragma unroll
        for (int b = 0; b < 12; ++b) goff[b] = rcv_reflect101(xb0 - 4 + b, s.cols);
    }

    struct Row { uint32_t d[3]; };
    auto load_row = [&](int ry) -> Row {
        ry = min(ry, ye - 1 + RAD);
        const uint8_t* row = sf + (size_t)rcv_reflect101(ry, s.rows) * s.step;
        Row w;
        if constexpr (EDGE) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                w.d[i] = (uint32_t)row[goff[4 * i]] | ((uint32_t)row[goff[4 * i + 1]] << 8) | ((uint32_t)row[goff[4 * i + 2]] << 16) |
                         ((uint32_t)row[goff[4 * i + 3]] << 24);
        } else if constexpr (RAG) {
            const uint8_t* p = row + wstart;
            const unsigned mis = (unsigned)((uintptr_t)p & 3);
            const uint32_t* base = (const uint32_t*)(p - mis);
            const uint32_t e0 = base[0], e1 = base[1], e2 = base[2], e3 = base[mis ? 3 : 2];   // (aligned window: no fourth dword)
            w.d[0] = __builtin_amdgcn_alignbyte(e1, e0, mis);
            w.d[1] = __builtin_amdgcn_alignbyte(e2, e1, mis);
            w.d[2] = __builtin_amdgcn_alignbyte(e3, e2, mis);
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) w.d[i] = *(const uint32_t*)(row + wstart + 4 * i);
        }
        return w;
    };

    int acc[KS][4], accq[DUAL ? KS : 1][4];
#pragma unroll
    for (int i = 0; i < KS; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = W.acc_init;
            if constexpr (DUAL) accq[i][j] = 0;
        }

    // feed row r (r mod KS == RHO, static): contributes kernel row ky to output y = r - ky + RAD, slot y mod KS
    auto feed = [&](const Row& w, int r, auto rho_tag) __attribute__((always_inline)) {
        constexpr int RHO = decltype(rho_tag)::value;
        const uint32_t D[4] = {w.d[0] ^ 0x80808080u, w.d[1] ^ 0x80808080u, w.d[2] ^ 0x80808080u, 0u};
        // operands of pixel j: window bytes o .. o+3 and o+4 .. o+7 with o = 4 - RAD + j (bytes that only meet zero weights
        // may come from anywhere)
        uint32_t op[4][ND];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < ND; ++h) {
                const int o = 4 - RAD + j + 4 * h, q = o >> 2, sh = o & 3;
                op[j][h] = sh == 0 ? D[q] : __builtin_amdgcn_alignbyte(D[q + 1 < 4 ? q + 1 : 3], D[q], sh);
            }
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int slot = ((RHO - ky + RAD) % KS + KS) % KS;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h = 0; h < ND; ++h) {
                    acc[slot][j] = __builtin_amdgcn_sdot4((int)op[j][h], W.r[ky][h], acc[slot][j], false);
                    if constexpr (DUAL) accq[slot][j] = __builtin_amdgcn_sdot4((int)op[j][h], W.q[ky][h], accq[slot][j], false);
                }
        }
        constexpr int done = ((RHO - (KS - 1) + RAD) % KS + KS) % KS;
        const int y = r - RAD;
        int v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = acc[done][j];
            if constexpr (DUAL) v[j] += accq[done][j] << 2;
            acc[done][j] = W.acc_init;
            if constexpr (DUAL) accq[done][j] = 0;
        }
        const uint32_t o = rcv_ashr_sat_pk4(v[0], v[1], v[2], v[3], W.shift);
        if (y >= ys && y < ye) {
            if constexpr (RAG) {
                uint8_t* q = df + (size_t)y * d.step;
                if (xb0 + 4 <= s.cols) {
                    typedef uint32_t u1m __attribute__((aligned(1)));
                    *(u1m*)q = o;
                } else {
                    for (int b = 0; b < s.cols - xb0; ++b) q[b] = (uint8_t)(o >> (8 * b));
                }
            } else {
                *(uint32_t*)(df + (size_t)y * d.step) = o;
            }
        }
    };

    int r0 = ys - RAD;
    r0 = r0 >= 0 ? r0 / KS * KS : -((-r0 + KS - 1) / KS) * KS;
    Row cur = load_row(r0), nxt;
    for (int rb = r0; rb <= ye - 1 + RAD; rb += KS) {
        static_for<0, KS>([&](auto I) __attribute__((always_inline)) {
            nxt = load_row(rb + I + 1);   // next row in flight while this one is consumed
            feed(cur, rb + I, I);
            cur = nxt;
        });
    }
}

template <int KS, bool DUAL, bool RAG = false>
int launch(rcv_ctx* ctx, const View& s, const View& d, const int16_t* k, int shift)
{
    constexpr int RAD = KS / 2;
    GrayW<KS> W;
    memset(&W, 0, sizeof(W));
    long long ksum = 0;
    for (int ky = 0; ky < KS; ++ky)
        for (int kx = 0; kx < KS; ++kx) {
            const int w = k[ky * KS + kx];
            ksum += w;
            const int qv = DUAL ? (w >> 2) : 0, rv = DUAL ? w - 4 * qv : w;   // floor split: R in 0..3, Q in i8
            // tap kx multiplies window byte (4 - RAD + j) + kx of pixel j: byte kx of the operand pair
            W.r[ky][kx >> 2] |= (int)((uint32_t)(uint8_t)(int8_t)rv << (8 * (kx & 3)));
            W.q[ky][kx >> 2] |= (int)((uint32_t)(uint8_t)(int8_t)qv << (8 * (kx & 3)));
        }
    (void)RAD;
    W.shift = shift;
    W.acc_init = (int)(128 * ksum + (shift > 0 ? (1 << (shift - 1)) : 0));
    const int nthreads = (s.cols + 3) / 4;   // per row (RAG: the last one may own fewer than 4 pixels)
    const unsigned gx = (unsigned)((nthreads + kBlock - 1) / kBlock);
    int seg = s.rows;
    while ((long long)gx * ((s.rows + seg - 1) / seg) * s.n < 4096 && seg > 8 * KS) seg = (seg + 1) / 2;
    const unsigned gy = (unsigned)((s.rows + seg - 1) / seg);
    RCV_LAUNCH((k_filter_gray_dot4<KS, DUAL, false, RAG>), dim3(gx, gy, s.n), dim3(kBlock), 0, ctx->stream, s, d, W, seg, 0, 0);
    RCV_TRY(rcv_launch_check(ctx));
    // threads whose 12-byte window [xb0 - 4, xb0 + 8) leaves the row: the first one and the last one (xb0 = cols - 4)
    // (a width that is not a multiple of 4: the last two threads, the partial one and the one before it)
    const int nl = 1, nr = min(nthreads - 1, s.cols % 4 ? 2 : 1);
    const int eseg = 4 * KS < 32 ? 32 : 4 * KS;
    RCV_LAUNCH((k_filter_gray_dot4<KS, DUAL, true, RAG>), dim3(1, (unsigned)((s.rows + eseg - 1) / eseg), s.n), dim3(64), 0, ctx->stream, s, d, W,
                       eseg, nl, nr);
    return rcv_launch_check(ctx);
}

} // namespace

// 1-channel images, ksize 3 / 5 / 7, weights in [-512, 511]; anything else: RCV_ERR_UNSUPPORTED (the generic kernel takes it)
int rcv_filter_i16_gray(rcv_ctx* ctx, const View& s, const View& d, const int16_t* k, int ksize, int shift)
{
    if (s.ch != 1 || d.ch != 1 || (ksize != 3 && ksize != 5 && ksize != 7)) return RCV_ERR_UNSUPPORTED;
    if (s.cols < 12 || s.rows > 65535 * 8) return RCV_ERR_UNSUPPORTED;
    const bool rag = s.cols % 4 != 0 || (uintptr_t)s.p % 4 || s.step % 4 || (s.n > 1 && s.fstride % 4) || (uintptr_t)d.p % 4 || d.step % 4 ||
                     (d.n > 1 && d.fstride % 4);
    bool dual = false;
    for (int i = 0; i < ksize * ksize; ++i) {
        if (k[i] < -512 || k[i] > 511) return RCV_ERR_UNSUPPORTED;
        if (k[i] < -128 || k[i] > 127) dual = true;
    }
#define RCV_CASE(KS)                                                                                                          \
    if (ksize == KS)                                                                                                          \
        return rag ? (dual ? launch<KS, true, true>(ctx, s, d, k, shift) : launch<KS, false, true>(ctx, s, d, k, shift))     \
                   : (dual ? launch<KS, true>(ctx, s, d, k, shift) : launch<KS, false>(ctx, s, d, k, shift));
    RCV_CASE(3) RCV_CASE(5) RCV_CASE(7)
#undef RCV_CASE
    return RCV_ERR_UNSUPPORTED;
}
