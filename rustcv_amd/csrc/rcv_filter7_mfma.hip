// rcv_filter7_mfma.hip -- filter2D with integer (i8) weights, ksize 3/5/7, u8 BGR -> u8 BGR, as a
// sliding-window stencil whose 49 MACs per sample run on the i8 matrix cores of gfx950.
//
// Why MFMA here: the op moves 6 algorithmic bytes per pixel but needs
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
// Data movement (all global accesses are aligned 16-byte vectors):
//   * a workgroup (5 waves) owns a STRIP 240 px wide (15 tiles) x a row SEGMENT, and walks down it in
//     16-row steps; each source row is fetched from HBM once per strip (halo: 6 px per 240, 6 rows per
//     segment);
//   * staging: lane (row r, chunk q) loads the 64 contiguous bytes that contain its 16 pixels shifted
//     by the 3-pixel halo, realigns them with v_alignbyte, de-interleaves BGR -> three planar dwords x4
//     with v_perm, xors 0x80 and writes one ds_write_b128 per plane.  Planar rows live in a 48-slot
//     ring (3 blocks of 16 rows) with a 272-byte pitch (17 x 16 B: rows land on distinct bank groups);
//     block k+2 is loaded into registers while step k is computed, one barrier per step;
//   * BORDER_REFLECT_101 is resolved at staging: rows by picking the mirrored source row, the three
//     left/right halo pixels of the first/last strip by byte patches;
//   * epilogue: v_ashr_pk_u8_i32 does shift + saturate + pack (2 ops per 4 bytes), each lane stores
//     12 contiguous bytes (4 BGR pixels) with one dwordx3.
#include "rcv_internal.h"
#include "rcv_kernels.h"
#include "rcv_device_utils.h"
#include <string.h>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kWaves = 5;
constexpr int kThreads = kWaves * 64;
constexpr int kTiles = 15;            // 16-px tiles per strip
constexpr int kPitch = 272;           // bytes per planar row in LDS (xx = 0..255 used, +16 pad)
constexpr int kSlots = 48;            // ring of 3 blocks x 16 rows
constexpr int kPlane = kSlots * kPitch;

struct F7Args {
    const uint8_t* src;
    uint8_t* dst;
    const uint4* wtab;   // 4 MFMAs x 64 lanes x 16 B (A operands), device memory
    size_t sstep, dstep, sfs, dfs;
    int rows, cols;
    int ntiles_total, nstrips, seg_rows, nsegs;
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

__global__ __launch_bounds__(kThreads) void k_filter7_mfma(F7Args a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[3 * kPlane];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int strip = bid % a.nstrips;
    bid /= a.nstrips;
    const int seg = bid % a.nsegs;
    const int frame = bid / a.nsegs;

    const int tile0 = strip * kTiles;
    const int ntiles = min(kTiles, a.ntiles_total - tile0);
    const int x0 = tile0 * 16;
    const int ys = seg * a.seg_rows;
    const int ye = min(a.rows, ys + a.seg_rows);
    const int nsteps = (ye - ys + 15) >> 4;
    const int rowbytes = a.cols * 3;

    const uint8_t* sframe = a.src + (size_t)frame * a.sfs;
    uint8_t* dframe = a.dst + (size_t)frame * a.dfs;

    // A operands (banded weights), constant for the whole launch
    v4i A[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        uint4 w = a.wtab[p * 64 + lane];
        A[p] = v4i{(int)w.x, (int)w.y, (int)w.z, (int)w.w};
    }

    // ---- staging task of this thread: (row r of a block, 16-pixel chunk q of the strip) ----
    const int sr = tid >> 4, sq = tid & 15;
    const bool stager = tid < 256 && sq <= ntiles;
    const int soff0 = 3 * x0 + 48 * sq - 16;  // byte offset in the source row of the first of 4 vectors
    uint4 L[4];

    auto load_block = [&](int b) {
        const int ry = ys - 3 + 16 * b + sr;
        const bool act = stager && ry <= ye + 2;
        const int srow = ry < 0 ? -ry : (ry >= a.rows ? 2 * a.rows - 2 - ry : ry);
        const uint8_t* base = sframe + (size_t)srow * a.sstep;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = soff0 + 16 * j;
            const bool ok = act && o >= 0 && o + 16 <= rowbytes;
            L[j] = ok ? *(const uint4*)(base + o) : make_uint4(0, 0, 0, 0);
        }
    };

    auto store_block = [&](int b) {
        const int rr = 16 * b + sr;
        const int ry = ys - 3 + rr;
        if (!(stager && ry <= ye + 2)) return;
        const uint32_t w[16] = {L[0].x, L[0].y, L[0].z, L[0].w, L[1].x, L[1].y, L[1].z, L[1].w,
                                L[2].x, L[2].y, L[2].z, L[2].w, L[3].x, L[3].y, L[3].z, L[3].w};
        uint32_t s[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) s[i] = __builtin_amdgcn_alignbyte(w[i + 2], w[i + 1], 3);  // bytes [7+4i, 11+4i)
        uint32_t pb[4], pg[4], pr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) deint4(s[3 * i], s[3 * i + 1], s[3 * i + 2], pb[i], pg[i], pr[i]);
        const int slot = rr % kSlots;
        uint8_t* dstp = lds + slot * kPitch + 16 * sq;
        *(uint4*)(dstp) = make_uint4(pb[0] ^ 0x80808080u, pb[1] ^ 0x80808080u, pb[2] ^ 0x80808080u, pb[3] ^ 0x80808080u);
        *(uint4*)(dstp + kPlane) = make_uint4(pg[0] ^ 0x80808080u, pg[1] ^ 0x80808080u, pg[2] ^ 0x80808080u, pg[3] ^ 0x80808080u);
        *(uint4*)(dstp + 2 * kPlane) = make_uint4(pr[0] ^ 0x80808080u, pr[1] ^ 0x80808080u, pr[2] ^ 0x80808080u, pr[3] ^ 0x80808080u);
        // BORDER_REFLECT_101 in x: the chunk's pixel e is image x = x0 - 3 + 16*sq + e
        const int xa = x0 - 3 + 16 * sq;
        if (xa < 0 || xa + 5 >= a.cols) {
            const int srow = ry < 0 ? -ry : (ry >= a.rows ? 2 * a.rows - 2 - ry : ry);
            const uint8_t* base = sframe + (size_t)srow * a.sstep;
            for (int e = 0; e < 6; ++e) {
                const int x = xa + e;
                if (x >= 0 && x < a.cols) continue;
                const int xs = x < 0 ? -x : 2 * a.cols - 2 - x;
                if (xs < 0 || xs >= a.cols) continue;  // beyond the 3-px halo: multiplied by zero weights
                dstp[e] = base[3 * xs] ^ 0x80;
                dstp[kPlane + e] = base[3 * xs + 1] ^ 0x80;
                dstp[2 * kPlane + e] = base[3 * xs + 2] ^ 0x80;
            }
        }
    };

    // ---- prologue: blocks 0 and 1 ----
    load_block(0);
    store_block(0);
    load_block(1);
    store_block(1);
    __syncthreads();

    const int n = lane & 15, kb = lane >> 4;
    const int kyl = kb >> 1, xh = (kb & 1) * 16;

    for (int k = 0; k < nsteps; ++k) {
        const bool more = k + 1 < nsteps;
        if (more) load_block(k + 2);

        int off[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) off[p] = ((16 * k + n + 2 * p + kyl) % kSlots) * kPitch + xh;
        const int y = ys + 16 * k + n;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int t = 3 * wave + i;
            if (t < ntiles) {
                v4i acc[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    acc[c] = v4i{a.acc_init, a.acc_init, a.acc_init, a.acc_init};
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const v4i b = *(const v4i*)(lds + c * kPlane + off[p] + 16 * t);
                        acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[p], b, acc[c], 0, 0, 0);
                    }
                }
                if (y < ye) {
                    // lane holds x = x0 + 16t + 4*kb + {0,1,2,3} for row y: 12 interleaved bytes
                    const uint32_t w0 = rcv_ashr_sat_pk4(acc[0][0], acc[1][0], acc[2][0], acc[0][1], a.shift);
                    const uint32_t w1 = rcv_ashr_sat_pk4(acc[1][1], acc[2][1], acc[0][2], acc[1][2], a.shift);
                    const uint32_t w2 = rcv_ashr_sat_pk4(acc[2][2], acc[0][3], acc[1][3], acc[2][3], a.shift);
                    uint32_t* o = (uint32_t*)(dframe + (size_t)y * a.dstep + 3 * (x0 + 16 * t + 4 * kb));
                    struct U3 { uint32_t a, b, c; };
                    *(U3*)o = U3{w0, w1, w2};
                }
            }
        }

        if (more) store_block(k + 2);
        __syncthreads();
    }
}

// host: banded A operands.  K7 is the kernel embedded (centred) in 7x7.
void build_wtab(const int8_t* k, int ksize, int8_t* tab /*4*64*16*/)
{
    int8_t K7[7][7];
    memset(K7, 0, sizeof(K7));
    int o = (7 - ksize) / 2;
    for (int y = 0; y < ksize; ++y)
        for (int x = 0; x < ksize; ++x) K7[y + o][x + o] = k[y * ksize + x];
    for (int p = 0; p < 4; ++p)
        for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 16; ++i) {
                int m = lane & 15, kb = lane >> 4, kyl = kb >> 1, j = (kb & 1) * 16 + i;
                int ky = 2 * p + kyl, tap = j - m;
                tab[(p * 64 + lane) * 16 + i] = (ky < 7 && tap >= 0 && tap <= 6) ? K7[ky][tap] : 0;
            }
}

} // namespace

int rcv_filter_i8_fast(rcv_ctx* ctx, const View& s, const View& d, const int8_t* k, int ksize, int shift)
{
    if (ksize != 3 && ksize != 5 && ksize != 7) return RCV_ERR_UNSUPPORTED;
    if (s.ch != 3) return RCV_ERR_UNSUPPORTED;
    if (s.cols % 16 != 0 || s.cols < 16 || s.rows < 4) return RCV_ERR_UNSUPPORTED;
    if ((uintptr_t)s.p % 16 || s.step % 16 || (s.n > 1 && s.fstride % 16)) return RCV_ERR_UNSUPPORTED;
    if ((uintptr_t)d.p % 4 || d.step % 4 || (d.n > 1 && d.fstride % 4)) return RCV_ERR_UNSUPPORTED;

    // weight table: rebuilt/uploaded only when the kernel changes (the upload is stream-ordered)
    if (!ctx->f7_valid || ctx->f7_ksize != ksize || memcmp(ctx->f7_k, k, (size_t)ksize * ksize) != 0) {
        int8_t tab[4 * 64 * 16];
        build_wtab(k, ksize, tab);
        ctx->f7_valid = false;
        RCV_TRY(rcv_upload_const(ctx, tab, sizeof(tab), 0));
        RCV_HIP(hipStreamSynchronize(ctx->stream)); // `tab` is on this stack frame
        memcpy(ctx->f7_k, k, (size_t)ksize * ksize);
        ctx->f7_ksize = ksize;
        ctx->f7_valid = true;
    }
    int ksum = 0;
    for (int i = 0; i < ksize * ksize; ++i) ksum += k[i];

    F7Args a;
    a.src = s.p;
    a.dst = d.p;
    a.wtab = (const uint4*)ctx->kconst;
    a.sstep = s.step;
    a.dstep = d.step;
    a.sfs = s.fstride;
    a.dfs = d.fstride;
    a.rows = s.rows;
    a.cols = s.cols;
    a.ntiles_total = s.cols / 16;
    a.nstrips = (a.ntiles_total + kTiles - 1) / kTiles;
    // segments: enough workgroups to fill 256 CUs x 4 several times over, >= 8 steps each
    int seg_rows = 720;
    long long wgs = (long long)a.nstrips * ((s.rows + seg_rows - 1) / seg_rows) * s.n;
    while (wgs < 2048 && seg_rows > 128) {
        seg_rows /= 2;
        seg_rows = (seg_rows + 15) & ~15;
        wgs = (long long)a.nstrips * ((s.rows + seg_rows - 1) / seg_rows) * s.n;
    }
    a.seg_rows = seg_rows;
    a.nsegs = (s.rows + seg_rows - 1) / seg_rows;
    a.shift = shift;
    a.acc_init = 128 * ksum + (shift > 0 ? (1 << (shift - 1)) : 0);
    long long grid = (long long)a.nstrips * a.nsegs * s.n;
    if (grid > 0x7fffffffLL) return RCV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_filter7_mfma, dim3((unsigned)grid), dim3(kThreads), 0, ctx->stream, a);
    return rcv_launch_check(ctx);
}
