// rcv_gauss_rows.hip -- integer GaussianBlur (sigma <= 0), ksize 3 / 5, u8, 1 or 3 channels, for SMALL launches: a register
// sliding window in packed 16-bit arithmetic, no LDS, no MFMA, no barrier (round 3; BASELINE config 2: ONE 1080p BGR frame, 5x5).
//
// Why a second kernel for an op the MFMA kernels already cover: one 1080p frame is 12 MB of traffic -- 2 us at the roofline, all of
// it cache-resident -- so the launch is a LATENCY problem: what counts is how soon after dispatch every SIMD has its rows, and how
// few dependent steps follow.  The MFMA strip kernel's latency variant needs a weight table, LDS staging and a barrier per step
// (6.3 us kernel time); here a wave issues ALL the row loads of its segment up front (one memory latency) and then only computes.
//
// The op is separable with binomial taps, all three channels use the same taps, and BORDER_REFLECT_101 acts per channel -- so a row
// is treated as RB = cols * CH independent BYTE columns for the vertical pass, and the horizontal pass reaches CH and 2 CH bytes
// left and right:
//     v(x, y) = sum_j t[j] * p(x, y + j - R)                       vertical, packed u16 (<= 16 * 255)
//     h(x, y) = sum_i t[i] * v(x + (i - R) * CH, y) + D / 2        horizontal, packed u16 (<= 256 * 255 + 128 < 65536: exact)
//     out     = h >> log2(D)                                       D = 16 (ksize 3), 256 (ksize 5): round half up, SURVEY.md 8-A
// A lane owns 16 consecutive bytes of a row (one aligned dwordx4), unpacked to eight u16 pairs.  The vertical pass is the
// transposed FIR: four (two) accumulator rows, `out = s4 + P; s4 = s3 + 4 P; s3 = s2 + 6 P; s2 = s1 + 4 P; s1 = P` -- every source
// row is loaded once per segment and touched once.  Packed pairs are handled with 32-bit integer instructions wherever no carry can
// cross the halves (v_lshl_add_u32 = x4 + c, v_add3_u32 folds the rounding constant in), v_pk_mad_u16 for x6.  Horizontal
// neighbours: even distances are other pairs, odd distances (CH = 3: 3 values) are v_alignbit of two neighbouring pairs; what lies
// in the neighbouring lanes comes by DPP wave shifts (3 + 3 per row).  Lanes 0 and 63 of a wave are halo lanes: a strip owns 62 x 16 =
// 992 bytes of a row.  The mirrored columns left of byte 0 / right of byte RB are the lane's OWN vertical sums (the vertical pass is
// per column, so mirrored columns have equal sums): one v_perm + two moves in the first / last strip, no extra load.
// Shapes: RB % 16 == 0, 16-byte aligned rows and frames, rows > R, cols > R; launches of at most a few waves per SIMD (the caller's
// plan: rcv_plan_seg_rows) -- everything else stays on the MFMA kernels (HBM-bound there, nothing to gain).
#include "rcv_internal.h"
#include "rcv_kernels.h"

namespace {

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

struct GRArgs {
    const uint8_t* src;
    uint8_t *dst, *dump;
    size_t sstep, dstep, sfs, dfs;
    int rows, rb;                          // rows, bytes per row (cols * channels)
    int nstrips, seg_rows, nsegs, total_waves;
    int plain_stores;                      // (knob RCV_GR_PLAIN: ablation)
};

constexpr int kOwnLanes = 62, kStripBytes = kOwnLanes * 16;
constexpr int kGroup = 8;   // source rows per prefetch group; two groups are in flight

__device__ __forceinline__ uint32_t shl_add(uint32_t p, uint32_t c, int sh)   // (p << sh) + c as ONE instruction (the compiler splits it)
{
    uint32_t d;
    if (sh == 2) asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(d) : "v"(p), "v"(c));
    else asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(d) : "v"(p), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t pk_mad6(uint32_t p, uint32_t c)   // p * 6 + c on both halves
{
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, 6, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(p), "v"(c));
    return d;
}
// (v[a], v[b]) from the lane's own pairs V[i] = (v[2i], v[2i + 1]); a, b compile-time
template <int A, int B>
__device__ __forceinline__ uint32_t own_pair(const uint32_t (&V)[8])
{
    if constexpr ((A & 1) == 0 && B == A + 1) return V[A >> 1];
    else {
        constexpr uint32_t sel = (uint32_t)(2 * (A & 1)) | ((uint32_t)(2 * (A & 1) + 1) << 8) | ((uint32_t)(4 + 2 * (B & 1)) << 16) | ((uint32_t)(5 + 2 * (B & 1)) << 24);
        return __builtin_amdgcn_perm(V[B >> 1], V[A >> 1], sel);
    }
}
// REFLECT_101 of the value index i (relative to the lane's first byte) about the image's left edge at index 0 / right edge at 16
template <int CH> constexpr int mir_left(int i) { const int px = -((-i + CH - 1) / CH), c = i - px * CH; return -px * CH + c; }           // i < 0
template <int CH> constexpr int mir_right(int i) { const int p = (i - 16) / CH, c = (i - 16) - p * CH; return 16 - (p + 2) * CH + c; }   // i >= 16

template <int KS, int CH>
__global__ __launch_bounds__(256) void k_gauss_rows(GRArgs a)
{
    constexpr int R = KS / 2;
    constexpr int NH = (R * CH + 1) / 2;            // pairs needed from each neighbouring lane (CH = 3, KS = 5: 3)
    constexpr uint32_t kBias = KS == 5 ? 0x00800080u : 0x00080008u;
    static_assert(KS == 3 || KS == 5, "binomial taps whose sums fit 16 bits");
    static_assert(R * CH + CH - 1 <= 15 && 16 - (R + 1) * CH >= 0, "the mirrored columns lie inside the edge lane's own 16 bytes");
    // a workgroup = four independent waves (one per SIMD of its CU); wave -> (strip, segment, frame), strips fastest
    const int lane = threadIdx.x & 63;
    int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (wid >= a.total_waves) return;
    const int strip = wid % a.nstrips;
    wid /= a.nstrips;
    const int seg = wid % a.nsegs, frame = wid / a.nsegs;
    // segments of equal height up to one row: rows * seg / nsegs (the plan picks nsegs so that the waves fill the SIMDs evenly)
    const int ys = (int)((long long)a.rows * seg / a.nsegs), ye = (int)((long long)a.rows * (seg + 1) / a.nsegs);
    const int X = strip * kStripBytes;                      // first owned byte of the strip
    const int xb = X - 16 + 16 * lane;                      // the lane's first (logical) byte
    const unsigned xo = (unsigned)min(max(xb, 0), a.rb - 16);   // clamped load position: halo / dead lanes re-read valid bytes
    const bool owned = lane >= 1 && lane <= kOwnLanes && xb < a.rb;
    const bool edge_l = strip == 0, edge_r = X + kStripBytes >= a.rb;   // (uniform) this strip holds the image's first / last bytes
    const bool is_l = xb == 0, is_r = xb == a.rb - 16;       // the lane whose neighbour lies outside the image
    const uint8_t* sf = a.src + (size_t)frame * a.sfs + xo;
    uint8_t* df = a.dst + (size_t)frame * a.dfs + (size_t)max(xb, 0);

    const unsigned sstep32 = (unsigned)a.sstep, dstep32 = (unsigned)a.dstep;
    auto load_row = [&](int j) -> v4u {   // fed row j of the segment = image row ys - R + j, reflected; past the segment's last row: re-read it
        int y = min(ys - R + j, ye - 1 + R);
        y = y < 0 ? -y : (y >= a.rows ? 2 * a.rows - 2 - y : y);
        return *(const v4u*)(sf + (unsigned)y * sstep32);   // (in-frame offsets below 2^32: host check)
    };

    uint32_t s1[8], s2[8], s3[8], s4[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = s3[i] = s4[i] = 0u;

    auto feed = [&](const v4u& w, int j) {   // j = fed row index (scalar); completes output row ys + j - 2 R
        uint32_t V[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t P0 = __builtin_amdgcn_perm(0u, w[q], 0x0c010c00u), P1 = __builtin_amdgcn_perm(0u, w[q], 0x0c030c02u);
            const int i0 = 2 * q, i1 = 2 * q + 1;
            if constexpr (KS == 5) {
                V[i0] = s4[i0] + P0;            V[i1] = s4[i1] + P1;
                s4[i0] = shl_add(P0, s3[i0], 2); s4[i1] = shl_add(P1, s3[i1], 2);
                s3[i0] = pk_mad6(P0, s2[i0]);   s3[i1] = pk_mad6(P1, s2[i1]);
                s2[i0] = shl_add(P0, s1[i0], 2); s2[i1] = shl_add(P1, s1[i1], 2);
            } else {
                V[i0] = s2[i0] + P0;            V[i1] = s2[i1] + P1;
                s2[i0] = shl_add(P0, s1[i0], 1); s2[i1] = shl_add(P1, s1[i1], 1);
            }
            s1[i0] = P0;
            s1[i1] = P1;
        }
        const int y = ys + j - 2 * R;
        if (j < 2 * R || y >= ye) return;   // (uniform) the window is not full yet / rows past the segment
        // ---- horizontal pass: VX[i + NH] = pair i of the row for i = -NH .. 7 + NH ----
        uint32_t VX[8 + 2 * NH];
#pragma unroll
        for (int i = 0; i < 8; ++i) VX[NH + i] = V[i];
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            VX[i] = __builtin_amdgcn_update_dpp(0u, V[8 - NH + i], 0x138, 0xf, 0xf, true);       // wave_shr:1: lane - 1's last pairs
            VX[NH + 8 + i] = __builtin_amdgcn_update_dpp(0u, V[i], 0x130, 0xf, 0xf, true);     // wave_shl:1: lane + 1's first pairs
        }
        if (edge_l) {   // pairs -NH .. -1 of the lane at byte 0: the mirror images of its own columns
            uint32_t m[NH];
            if constexpr (NH == 3) { m[0] = own_pair<mir_left<CH>(-6), mir_left<CH>(-5)>(V); m[1] = own_pair<mir_left<CH>(-4), mir_left<CH>(-3)>(V); m[2] = own_pair<mir_left<CH>(-2), mir_left<CH>(-1)>(V); }
            else if constexpr (NH == 2) { m[0] = own_pair<mir_left<CH>(-4), mir_left<CH>(-3)>(V); m[1] = own_pair<mir_left<CH>(-2), mir_left<CH>(-1)>(V); }
            else m[0] = own_pair<mir_left<CH>(-2), mir_left<CH>(-1)>(V);
#pragma unroll
            for (int i = 0; i < NH; ++i) VX[i] = is_l ? m[i] : VX[i];
        }
        if (edge_r) {
            uint32_t m[NH];
            m[0] = own_pair<mir_right<CH>(16), mir_right<CH>(17)>(V);
            if constexpr (NH >= 2) m[1] = own_pair<mir_right<CH>(18), mir_right<CH>(19)>(V);
            if constexpr (NH >= 3) m[2] = own_pair<mir_right<CH>(20), mir_right<CH>(21)>(V);
#pragma unroll
            for (int i = 0; i < NH; ++i) VX[NH + 8 + i] = is_r ? m[i] : VX[NH + 8 + i];
        }
        // pair k shifted by an EVEN number of values d: pair k + d / 2
        auto at = [&](int k, int d) -> uint32_t { return VX[NH + k + d / 2]; };
        uint32_t H[8];
        // odd distances: (v[2k + d], v[2k + d + 1]) = O[k + floor((d - 1) / 2)] with the shifted pairs O[m] = (v[2m + 1], v[2m + 2]) are shared between the outputs that use them: built once)
        uint32_t O[8 + 2 * NH];
        if constexpr (CH & 1) {
#pragma unroll
            for (int m = 0; m < 8 + 2 * NH - 1; ++m) O[m] = __builtin_amdgcn_alignbit(VX[m + 1], VX[m], 16);   // O index m <-> pair m - NH
        }
        auto odd = [&](int k, int d) -> uint32_t {   // d odd
            const int m = k + (d >= 1 ? (d - 1) / 2 : -((2 - d) / 2));
            return O[NH + m];
        };
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if constexpr (KS == 5) {
                const uint32_t t = at(k, -2 * CH) + at(k, 2 * CH) + kBias;   // (v_add3_u32; 2 CH is even: plain pairs)
                const uint32_t u = (CH & 1) ? odd(k, -CH) + odd(k, CH) : at(k, -CH) + at(k, CH);
                H[k] = pk_mad6(V[k], shl_add(u, t, 2));
            } else {
                const uint32_t t = ((CH & 1) ? odd(k, -CH) + odd(k, CH) : at(k, -CH) + at(k, CH)) + kBias;
                H[k] = shl_add(V[k], t, 1) >> 4;   // (a 32-bit shift: only the low byte of each half is kept below)
            }
        }
        constexpr uint32_t kPack = KS == 5 ? 0x07050301u : 0x06040200u;   // byte 1 / byte 0 of the four halves
        const v4u o = {__builtin_amdgcn_perm(H[1], H[0], kPack), __builtin_amdgcn_perm(H[3], H[2], kPack), __builtin_amdgcn_perm(H[5], H[4], kPack),
                       __builtin_amdgcn_perm(H[7], H[6], kPack)};
        // Non-temporal: the output streams out while the kernel runs instead of being written back from L2 at the kernel's end (a
        // launch-bound op pays for that).  Exec-masked, not redirected to a dump line: every wave has two halo lanes, and
        // non-temporal stores of all waves to one shared line serialise (measured: 5.6 -> 12 us).
        if (owned) {
            if (a.plain_stores) *(v4u*)(df + (unsigned)y * dstep32) = o;
            else __builtin_nontemporal_store(o, (v4u*)(df + (unsigned)y * dstep32));
        }
    };

    const int nfed = ye - ys + 2 * R;
    v4u A[kGroup], B[kGroup];
#pragma unroll
    for (int i = 0; i < kGroup; ++i) A[i] = load_row(i);
    for (int g = 0; g < nfed; g += 2 * kGroup) {
#pragma unroll
        for (int i = 0; i < kGroup; ++i) B[i] = load_row(g + kGroup + i);
#pragma unroll
        for (int i = 0; i < kGroup; ++i) feed(A[i], g + i);
        if (g + kGroup >= nfed) break;
#pragma unroll
        for (int i = 0; i < kGroup; ++i) A[i] = load_row(g + 2 * kGroup + i);
#pragma unroll
        for (int i = 0; i < kGroup; ++i) feed(B[i], g + kGroup + i);
    }
}

} // namespace

// sigma <= 0 GaussianBlur of ksize 3 / 5 on the register-window kernel: small launches of aligned shapes (see the file header);
// RCV_ERR_UNSUPPORTED for everything else.  Knob RCV_GAUSS_ROWS: 1 = every eligible shape whatever the size, 0 = never.
int rcv_gauss_int_rows(rcv_ctx* ctx, const View& s, const View& d, int ksize)
{
    const RcvKnobs& kn = rcv_knobs();
    if (kn.gauss_rows == 0) return RCV_ERR_UNSUPPORTED;
    if ((ksize != 3 && ksize != 5) || (s.ch != 1 && s.ch != 3) || d.ch != s.ch) return RCV_ERR_UNSUPPORTED;
    const long long rb = (long long)s.cols * s.ch;
    const int R = ksize / 2;
    if (rb % 16 || rb < 32 || s.rows <= R || s.cols <= R + 1) return RCV_ERR_UNSUPPORTED;
    if ((uintptr_t)s.p % 16 || s.step % 16 || (s.n > 1 && s.fstride % 16) || (uintptr_t)d.p % 16 || d.step % 16 || (d.n > 1 && d.fstride % 16)) return RCV_ERR_UNSUPPORTED;
    if (rb >= (1 << 30) || s.rows >= (1 << 24) || (unsigned long long)s.rows * s.step >= (1ull << 32) || (unsigned long long)s.rows * d.step >= (1ull << 32)) return RCV_ERR_UNSUPPORTED;
    GRArgs a;
    a.nstrips = (int)((rb + kStripBytes - 1) / kStripBytes);
    // small launches only: every wave resident at once, at most ~3 per SIMD (rcv_plan_seg_rows returns 0 beyond that)
    int seg = rcv_plan_seg_rows(s.rows, (long long)a.nstrips * s.n, ctx->cu_count, 2 * R + 3, 4);
    if (seg == 0 && kn.gauss_rows != 1) return RCV_ERR_UNSUPPORTED;
    // beyond about one 1080p frame the row-streaming MFMA kernel is faster (one 4K frame: 10.8 against 12.6 us)
    if (kn.gauss_rows < 0 && (long long)a.nstrips * s.rows * s.n > 10000) return RCV_ERR_UNSUPPORTED;
    if (seg == 0) seg = 32;
    if (kn.gr_seg > 0) seg = kn.gr_seg;
    int nsegs = (s.rows + seg - 1) / seg;
    if (kn.gr_seg <= 0) {
        // a whole number of waves per SIMD: k x (4 x CUs) waves, k = 1, 2, 3 ... -- the launch lasts as long as its busiest SIMD
        const long long simds = 4LL * ctx->cu_count, per = (long long)a.nstrips * s.n;
        const long long k = ((long long)nsegs * per + simds / 2) / simds;
        if (k >= 1) {
            const long long ns = k * simds / per;
            if (ns >= 1 && ns <= s.rows / 2) nsegs = (int)ns;
        }
    }
    a.src = s.p;
    a.dst = d.p;
    a.dump = ctx->kconst + RCV_KC_SOBEL_DUMP;
    a.sstep = s.step;
    a.dstep = d.step;
    a.sfs = s.fstride;
    a.dfs = d.fstride;
    a.rows = s.rows;
    a.rb = (int)rb;
    a.seg_rows = seg;
    a.plain_stores = 0;
    a.nsegs = nsegs;
    const long long waves = (long long)a.nstrips * a.nsegs * s.n;
    if (waves > 0x3fffffff) return RCV_ERR_UNSUPPORTED;
    a.total_waves = (int)waves;
    const dim3 grid((unsigned)((waves + 3) / 4));
    if (ksize == 5 && s.ch == 3) RCV_LAUNCH((k_gauss_rows<5, 3>), grid, dim3(256), 0, ctx->stream, a);
    else if (ksize == 5) RCV_LAUNCH((k_gauss_rows<5, 1>), grid, dim3(256), 0, ctx->stream, a);
    else if (s.ch == 3) RCV_LAUNCH((k_gauss_rows<3, 3>), grid, dim3(256), 0, ctx->stream, a);
    else RCV_LAUNCH((k_gauss_rows<3, 1>), grid, dim3(256), 0, ctx->stream, a);
    return rcv_launch_check(ctx);
}
