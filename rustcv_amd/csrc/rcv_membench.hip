// rcv_membench.hip -- device-memory calibration kernels: what a plain copy / read / write of N bytes costs on THIS GPU in THIS
// run.  bench.py times them next to the north-star kernel (roofline.copy_ceiling_gbs): the filter moves the same bytes as a
// copy of the batch, so the best copy rate measured in the same process is the ceiling it can be held against.
#include "rcv_internal.h"

namespace {

constexpr int kT = 256;
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

template <bool NTL, bool NTS>
__device__ __forceinline__ void cp4(const uint4* __restrict__ s, uint4* __restrict__ d, size_t i, size_t stride, size_t n)
{
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t j = i + u * stride;
        if (j < n) {
            if (NTL) {
                const u4v t = __builtin_nontemporal_load((const u4v*)(s + j));
                v[u] = make_uint4(t.x, t.y, t.z, t.w);
            } else {
                v[u] = s[j];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t j = i + u * stride;
        if (j < n) {
            if (NTS) __builtin_nontemporal_store(u4v{v[u].x, v[u].y, v[u].z, v[u].w}, (u4v*)(d + j));
            else d[j] = v[u];
        }
    }
}

// grid-stride: the whole grid sweeps the buffer front to back, 16 bytes per thread and access, four accesses in flight
template <bool NTL, bool NTS>
__global__ __launch_bounds__(kT) void k_copy_sweep(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n)
{
    const size_t stride = (size_t)gridDim.x * kT;
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += 4 * stride) cp4<NTL, NTS>(s, d, i, stride, n);
}

// the same sweep with U accesses in flight per thread (10 <= variant < 22: what the in-flight depth and the nt flags are worth)
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(kT) void k_copy_sweep_u(const u4v* __restrict__ s, u4v* __restrict__ d, size_t n)
{
    const size_t stride = (size_t)gridDim.x * kT;
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += U * stride) {
        u4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * stride;
            if (j < n) v[u] = NTL ? __builtin_nontemporal_load(s + j) : s[j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * stride;
            if (j < n) {
                if (NTS) __builtin_nontemporal_store(v[u], d + j);
                else d[j] = v[u];
            }
        }
    }
}

// Stream-count experiment (round 3): the nt sweep with U = 2 accesses in flight, but the buffer cut into `nstreams` equal contiguous
// regions, each swept by its own share of the workgroups (region r: workgroups r, r + nstreams, ...) -- what a band-streaming stencil
// does to the memory system (its resident waves walk ~136 separate row streams) without anything else of the stencil.  nstreams = 1
// is the plain sweep.  XCDL: a region's workgroups are consecutive in the list of ONE XCD (block b runs on XCD b % 8) instead of
// being dealt over all eight.
template <bool XCDL>
__global__ __launch_bounds__(kT) void k_copy_streams(const u4v* __restrict__ s, u4v* __restrict__ d, size_t n, int nstreams)
{
    int b = blockIdx.x;
    const int G = gridDim.x;
    if (XCDL) b = (b & 7) * (G / 8) + (b >> 3);   // position in the XCD-major list
    const int per = G / nstreams;                  // workgroups per region (host: G % nstreams == 0)
    const int r = XCDL ? b / per : b % nstreams, w = XCDL ? b % per : b / nstreams;
    const size_t len = n / nstreams, base = (size_t)r * len;
    const size_t stride = (size_t)per * kT;
    for (size_t i = (size_t)w * kT + threadIdx.x; i < len; i += 2 * stride) {
        u4v v0 = __builtin_nontemporal_load(s + base + i), v1 = v0;
        const size_t j = i + stride;
        if (j < len) v1 = __builtin_nontemporal_load(s + base + j);
        __builtin_nontemporal_store(v0, d + base + i);
        if (j < len) __builtin_nontemporal_store(v1, d + base + j);
    }
}

// Phase experiments (round 3): does it matter WHEN the GPU's reads and writes reach the memory system?  MODE 0: the plain nt
// sweep; 1: every wave de-synchronised by pseudo-random sleeps; 2: clock-gated -- loads are only issued while bit `hbit` of the
// chip-wide 100 MHz counter is 0 and stores while it is 1, so the whole GPU alternates between read and write bursts.
template <int U, int MODE>
__global__ __launch_bounds__(kT) void k_copy_phase(const u4v* __restrict__ s, u4v* __restrict__ d, size_t n, unsigned half_ticks)
{
    const size_t stride = (size_t)gridDim.x * kT;
    unsigned rng = (blockIdx.x * 2654435761u) ^ ((threadIdx.x >> 6) * 40503u);
    auto gate = [&](unsigned want) {
        if (MODE != 2) return;
        while (((unsigned)(__builtin_amdgcn_s_memrealtime() / half_ticks) & 1u) != want) __builtin_amdgcn_s_sleep(1);
    };
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += U * stride) {
        u4v v[U];
        if (MODE == 1) {
            rng = rng * 1664525u + 1013904223u;
            const unsigned k = __builtin_amdgcn_readfirstlane(rng >> 26);   // 0..63 x 64 clocks
            for (unsigned t = 0; t < k; ++t) __builtin_amdgcn_s_sleep(1);
        }
        gate(0u);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * stride;
            if (j < n) v[u] = __builtin_nontemporal_load(s + j);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        gate(1u);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * stride;
            if (j < n) __builtin_nontemporal_store(v[u], d + j);
        }
    }
}

// block-contiguous: every workgroup owns one contiguous slice of the buffer
template <bool NTL, bool NTS>
__global__ __launch_bounds__(kT) void k_copy_block(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n)
{
    const size_t per = (n + gridDim.x - 1) / gridDim.x, b0 = per * blockIdx.x, b1 = b0 + per < n ? b0 + per : n;
    for (size_t i = b0 + threadIdx.x; i < b1; i += 4 * kT) cp4<NTL, NTS>(s, d, i, kT, b1);
}

// XCD-local sweep: hardware places block b on XCD b % 8; XCD x sweeps its own contiguous eighth of the buffer (what one XCD has
// in flight stays a compact address range -- the layout the filter kernel uses, so the ceiling it is held against gets it too)
template <bool NTL, bool NTS>
__global__ __launch_bounds__(kT) void k_copy_xcd(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n)
{
    const size_t per = (n + 7) / 8, b0 = per * (blockIdx.x & 7), b1 = b0 + per < n ? b0 + per : n;
    const size_t stride = (size_t)(gridDim.x >> 3) * kT;
    for (size_t i = b0 + (size_t)(blockIdx.x >> 3) * kT + threadIdx.x; i < b1; i += 4 * stride) cp4<NTL, NTS>(s, d, i, stride, b1);
}

__global__ __launch_bounds__(kT) void k_read_sweep(const uint4* __restrict__ s, uint4* __restrict__ dump, size_t n)
{
    const size_t stride = (size_t)gridDim.x * kT;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += 4 * stride) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i + u * stride < n ? s[i + u * stride] : acc;
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) dump[threadIdx.x] = acc;   // (never: keeps the loads alive)
}

__global__ __launch_bounds__(kT) void k_write_sweep(uint4* __restrict__ d, size_t n, uint32_t seed)
{
    const size_t stride = (size_t)gridDim.x * kT;
    const uint4 v = make_uint4(seed, seed ^ threadIdx.x, seed + blockIdx.x, ~seed);
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += stride) d[i] = v;
}

// Store-pattern study (round 3, the Sobel kernel's store rate): waves own strips of `chunks` x 1 KB of a row in each of `planes`
// planes and walk down `seg_rows` rows, as the register-window kernels do; NT: non-temporal stores.  Block order as in those
// kernels (each XCD a contiguous run of the wave list) when blocks_per_xcd > 0.
struct StoreArgs {
    uint8_t* p[2];
    size_t step, fstride;
    int rows, nstrips, nsegs, seg_rows, total_waves, chunks, planes, blocks_per_xcd, pair_rows;
};
template <bool NT>
__global__ __launch_bounds__(256) void k_store_strips(StoreArgs a)
{
    const int lane = threadIdx.x & 63;
    const int blk = a.blocks_per_xcd > 0 ? (int)(blockIdx.x & 7) * a.blocks_per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    int wid = __builtin_amdgcn_readfirstlane(blk * 4 + (int)(threadIdx.x >> 6));
    if (wid >= a.total_waves) return;
    const int strip = wid % a.nstrips;
    wid /= a.nstrips;
    const int seg = wid % a.nsegs, frame = wid / a.nsegs;
    const int ys = seg * a.seg_rows, ye = min(a.rows, ys + a.seg_rows);
    const u4v v = {(unsigned)wid, (unsigned)lane, 0x5EEDu, (unsigned)strip};
    // pair_rows = r > 1: r rows of plane 0, then the same r rows of plane 1 (longer runs per plane before switching streams)
    const int pr = a.pair_rows > 1 ? a.pair_rows : 1;
    for (int y0 = ys; y0 < ye; y0 += pr)
        for (int pl = 0; pl < a.planes; ++pl)
            for (int y = y0; y < min(y0 + pr, ye); ++y) {
                uint8_t* row = a.p[pl] + (size_t)frame * a.fstride + (size_t)y * a.step + (size_t)strip * 1024 * a.chunks + 16 * lane;
                for (int c = 0; c < a.chunks; ++c) {
                    if (NT) __builtin_nontemporal_store(v, (u4v*)(row + 1024 * c));
                    else *(u4v*)(row + 1024 * c) = v;
                }
            }
}

// Strip-walker copy (round 4): the access pattern of a row-streaming stencil without the stencil -- which STRIP WIDTH / rows per
// request / depth does this memory system like, at a fixed number of resident waves?  A wave (= a 64-thread workgroup) owns a
// strip of W bytes of every row of a band and walks down it; one request = three 16-byte accesses per lane = 3072 bytes = R = 3072 / W
// whole strip-rows; DEPTH requests are in flight; every loaded byte is stored (non-temporal) to the same place in dst.  Bands,
// band order (each XCD a contiguous run of bands, the strips of a band neighbours) and occupancy (wpc waves per CU through an
// untouched dynamic-LDS request) as in k_filter_rows_mfma, whose plain instantiation is W = 768, R = 2, four requests in flight.
// halo > 0: a band also reads (never stores) `halo` rows above itself, like a 7-row stencil's band.
struct WalkArgs {
    const uint8_t* src;
    uint8_t* dst;
    size_t step, fstride;
    int rows, nframes, nstrips, W, R, nbands, bands_per_xcd, halo, ntl;
    int persist;   // ng > 0: ng band groups per XCD, wave (xcd, strip, g) walks bands g, g + ng, ... of its XCD itself (static, all waves resident)
    int dup;       // 1: every byte is requested by two lanes (lane pairs share their 16 bytes: the MFMA kernel's overlapping windows), half the rows per request
};
template <int DEPTH>
__global__ __launch_bounds__(64) void k_strip_walk(WalkArgs a)
{
    const int lane = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot = (int)(blockIdx.x >> 3);
    const int strip = slot % a.nstrips;
    int bi = slot / a.nstrips;
    if (a.persist ? bi >= a.persist : bi >= a.bands_per_xcd) return;
    const long long G = (long long)a.nframes * a.rows;
  for (; bi < a.bands_per_xcd; bi += (a.persist ? a.persist : a.bands_per_xcd)) {
    const int band = xcd * a.bands_per_xcd + bi;
    if (band >= a.nbands) return;
    long long g0 = G * band / a.nbands;
    const long long g1 = G * (band + 1) / a.nbands;
    const int spr = a.W >> 4;   // 16-byte slots per strip-row
    // the lane's three slots of a request: row and byte offset inside the R x W block
    size_t off[3];
    int rowj[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int sidx = a.dup ? (lane >> 1) + 32 * j : lane + 64 * j;
        rowj[j] = sidx / spr;
        off[j] = (size_t)rowj[j] * a.step + (size_t)(sidx % spr) * 16 + (size_t)strip * a.W;
    }
    while (g0 < g1) {
        const int frame = (int)(g0 / a.rows), ys = (int)(g0 - (long long)frame * a.rows);
        const int ye = (int)min((long long)a.rows, ys + (g1 - g0));
        const uint8_t* sf = a.src + (size_t)frame * a.fstride;
        uint8_t* df = a.dst + (size_t)frame * a.fstride;
        const int y0 = max(ys - a.halo, 0);
        const int nreq = (ye - y0 + a.R - 1) / a.R;
        u4v v[DEPTH][3];
        auto request = [&](int r, u4v(&d)[3]) {
            const int yb = y0 + r * a.R;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int y = min(yb + rowj[j], ye - 1) - (yb + rowj[j]);   // rows past the band re-read its last row
                const u4v* p = (const u4v*)(sf + (size_t)yb * a.step + off[j] + (long long)y * (long long)a.step);
                d[j] = a.ntl ? __builtin_nontemporal_load(p) : *p;
            }
        };
#pragma unroll
        for (int i = 0; i < DEPTH - 1; ++i) request(i < nreq ? i : nreq - 1, v[i]);
        for (int r0 = 0; r0 < nreq; r0 += DEPTH) {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                const int r = r0 + s;
                if (r >= nreq) break;
                request(min(r + DEPTH - 1, nreq - 1), v[(s + DEPTH - 1) % DEPTH]);
                const int yb = y0 + r * a.R;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int y = yb + rowj[j];
                    if (y >= ys && y < ye && !(a.dup && (lane & 1))) __builtin_nontemporal_store(v[s][j], (u4v*)(df + (size_t)yb * a.step + off[j]));
                }
            }
        }
        g0 += ye - ys;
    }
  }
}

__global__ __launch_bounds__(64) void k_nop(int* p)
{
    if (p && threadIdx.x == 1234567) *p = 0;   // (never)
}

// one wave per XCD-ish (8 workgroups): core-clock cycles (s_memtime) per tick of the constant 100 MHz counter (s_memrealtime)
// over `ticks` ticks -- the shader clock while whatever else runs on the GPU keeps running
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long* out, unsigned ticks)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0, c1 = c0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
        c1 = __builtin_amdgcn_s_memtime();
    }
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = c1 - c0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
}

} // namespace

// Shader clock in MHz, sampled for `us` microseconds on the context's SIDE stream -- i.e. concurrently with whatever the caller
// has enqueued on the context's stream (bench.py: the filter launches).  Blocks until the probe has finished.
extern "C" int rcv__clock_probe(rcv_ctx* ctx, int us, float* mhz)
{
    RCV_TRY(rcv_bind(ctx));
    if (!mhz || us < 1 || us > 1000000) return RCV_ERR_ARG;
    unsigned long long* d = (unsigned long long*)(ctx->kconst + RCV_KC_PROBE);
    // (the side stream exists only for this probe and is created on first use: ROCm multiplexes a process's streams onto four hardware queues, and a
    //  context's two launch streams must not end up sharing one)
    if (!ctx->side) RCV_HIP(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
    hipLaunchKernelGGL(k_clock_probe, dim3(8), dim3(64), 0, ctx->side, d, (unsigned)us * 100u);
    unsigned long long h[16];
    RCV_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->side));
    RCV_HIP(hipStreamSynchronize(ctx->side));
    double best = 0.0;
    for (int i = 0; i < 8; ++i)
        if (h[2 * i + 1]) {
            const double f = (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;
            if (f > best) best = f;
        }
    *mhz = (float)best;
    return RCV_OK;
}

// rows x row_bytes per plane per frame, n frames (frame stride = rows * step), written by strip-walking waves; see k_store_strips
extern "C" int rcv__storebench(rcv_ctx* ctx, void* p0, void* p1, int n, int rows, int row_bytes, size_t step, int chunks, int planes, int seg_rows, int nt,
                               int xcd_order, int wgs_per_cu, int pair_rows)
{
    RCV_TRY(rcv_bind(ctx));
    if (!p0 || (planes == 2 && !p1) || chunks < 1 || chunks > 4 || row_bytes % (1024 * chunks) || planes < 1 || planes > 2 || seg_rows < 1) return RCV_ERR_ARG;
    StoreArgs a;
    a.p[0] = (uint8_t*)p0;
    a.p[1] = (uint8_t*)p1;
    a.step = step;
    a.fstride = (size_t)rows * step;
    a.rows = rows;
    a.nstrips = row_bytes / (1024 * chunks);
    a.seg_rows = seg_rows;
    a.nsegs = (rows + seg_rows - 1) / seg_rows;
    a.total_waves = a.nstrips * a.nsegs * n;
    a.chunks = chunks;
    a.planes = planes;
    a.pair_rows = pair_rows;
    const long long nblocks = ((long long)a.total_waves + 3) / 4;
    a.blocks_per_xcd = xcd_order ? (int)((nblocks + 7) / 8) : 0;
    const dim3 grid((unsigned)(a.blocks_per_xcd > 0 ? a.blocks_per_xcd * 8 : nblocks));
    const unsigned lds = wgs_per_cu > 0 ? (unsigned)((163840 / wgs_per_cu) & ~511) : 0u;
    if (nt) hipLaunchKernelGGL((k_store_strips<true>), grid, dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL((k_store_strips<false>), grid, dim3(256), lds, ctx->stream, a);
    return rcv_launch_check(ctx);
}

// strip-walker copy of n frames of rows x row_bytes (frame stride = rows * step): strip width W (3072 % W == 0, row_bytes % W == 0,
// W % 16 == 0), `depth` requests of 3072 bytes in flight per wave (1, 2, 4), `rounds` bands per wave slot, wpc waves per CU (<= 8 ... 32),
// halo rows re-read per band, nt loads or plain.  Asynchronous on the context's stream.
extern "C" int rcv__stripwalk(rcv_ctx* ctx, void* dst, const void* src, int n, int rows, int row_bytes, size_t step, int W, int depth, int rounds, int wpc,
                              int halo, int flags)
{
    const int nt_loads = flags & 1, persist = (flags >> 1) & 1, dup = (flags >> 2) & 1;
    RCV_TRY(rcv_bind(ctx));
    if (!dst || !src || n < 1 || rows < 1 || W < 16 || W % 16 || 1536 % W || row_bytes % W || step % 16 || rounds < 1 || wpc < 1 || wpc > 32) return RCV_ERR_ARG;
    WalkArgs a;
    a.src = (const uint8_t*)src;
    a.dst = (uint8_t*)dst;
    a.step = step;
    a.fstride = (size_t)rows * step;
    a.rows = rows;
    a.nframes = n;
    a.W = W;
    a.R = (dup ? 1536 : 3072) / W;
    a.dup = dup;
    a.nstrips = row_bytes / W;
    a.halo = halo;
    a.ntl = nt_loads;
    const long long slots = (long long)wpc * ctx->cu_count;
    long long nb = (long long)rounds * slots / a.nstrips;
    if (n >= 8) {   // whole bands per frame, as the filter plans them
        long long bpf = (nb + n / 2) / n;
        bpf = bpf < 1 ? 1 : bpf;
        nb = bpf * n;
    }
    nb = nb < 8 ? 8 : nb;
    a.nbands = (int)nb;
    a.bands_per_xcd = (int)((nb + 7) / 8);
    a.persist = persist ? (int)(slots / 8 / a.nstrips) : 0;
    if (persist && a.persist < 1) return RCV_ERR_ARG;
    const dim3 grid((unsigned)((long long)(persist ? a.persist : a.bands_per_xcd) * a.nstrips * 8));
    const unsigned lds = wpc < 32 ? (unsigned)((163840 / wpc) & ~511) : 0u;
    if (lds > 65536u) {
        static bool once = false;
        if (!once) {
            once = true;
            (void)hipFuncSetAttribute((const void*)k_strip_walk<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
            (void)hipFuncSetAttribute((const void*)k_strip_walk<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
            (void)hipFuncSetAttribute((const void*)k_strip_walk<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        }
    }
    if (depth == 1) hipLaunchKernelGGL((k_strip_walk<1>), grid, dim3(64), lds, ctx->stream, a);
    else if (depth == 2) hipLaunchKernelGGL((k_strip_walk<2>), grid, dim3(64), lds, ctx->stream, a);
    else if (depth == 4) hipLaunchKernelGGL((k_strip_walk<4>), grid, dim3(64), lds, ctx->stream, a);
    else return RCV_ERR_ARG;
    return rcv_launch_check(ctx);
}

// variant: 0 hipMemcpyAsync D2D | 1 sweep | 2 block-contiguous | 3 sweep, nt loads + nt stores | 4 sweep, nt stores |
//          5 block-contiguous, nt loads + nt stores | 6 read only | 7 write only | 8 XCD-local sweep | 9 XCD-local sweep, nt / nt
//          (grid a multiple of 8).   Asynchronous on the context's stream.
extern "C" int rcv__membench(rcv_ctx* ctx, void* dst, const void* src, size_t bytes, int variant, int grid)
{
    RCV_TRY(rcv_bind(ctx));
    if (!dst || !src || bytes % 16 || ((uintptr_t)dst | (uintptr_t)src) % 16 || grid < 1) return RCV_ERR_ARG;
    const size_t n = bytes / 16;
    const dim3 g((unsigned)grid), b(kT);
    const uint4* s = (const uint4*)src;
    uint4* d = (uint4*)dst;
    switch (variant) {
    case 0: RCV_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream)); return RCV_OK;
    case 1: hipLaunchKernelGGL((k_copy_sweep<false, false>), g, b, 0, ctx->stream, s, d, n); break;
    case 2: hipLaunchKernelGGL((k_copy_block<false, false>), g, b, 0, ctx->stream, s, d, n); break;
    case 3: hipLaunchKernelGGL((k_copy_sweep<true, true>), g, b, 0, ctx->stream, s, d, n); break;
    case 4: hipLaunchKernelGGL((k_copy_sweep<false, true>), g, b, 0, ctx->stream, s, d, n); break;
    case 5: hipLaunchKernelGGL((k_copy_block<true, true>), g, b, 0, ctx->stream, s, d, n); break;
    case 6: hipLaunchKernelGGL(k_read_sweep, g, b, 0, ctx->stream, s, (uint4*)(ctx->kconst + RCV_KC_BENCH), n); break;
    case 7: hipLaunchKernelGGL(k_write_sweep, g, b, 0, ctx->stream, d, n, 0x5EEDu); break;
    case 30: hipLaunchKernelGGL(k_nop, g, dim3(64), 0, ctx->stream, (int*)nullptr); break;   // launch-to-launch floor: `grid` empty workgroups
    case 8:
        if (grid % 8) return RCV_ERR_ARG;
        hipLaunchKernelGGL((k_copy_xcd<false, false>), g, b, 0, ctx->stream, s, d, n);
        break;
    case 9:
        if (grid % 8) return RCV_ERR_ARG;
        hipLaunchKernelGGL((k_copy_xcd<true, true>), g, b, 0, ctx->stream, s, d, n);
        break;
    case 40: case 41: {   // N-stream sweep: `grid` bits 0..15 workgroups, bits 16..31 streams (41: a stream's workgroups on one XCD)
        const unsigned wgs = (unsigned)grid & 0xffffu, ns = (unsigned)grid >> 16;
        if (ns < 1 || wgs % ns || (variant == 41 && (wgs % 8 || (wgs / 8) % (wgs / ns)))) return RCV_ERR_ARG;
        if (n % ns) return RCV_ERR_ARG;
        if (variant == 40) hipLaunchKernelGGL((k_copy_streams<false>), dim3(wgs), b, 0, ctx->stream, (const u4v*)src, (u4v*)dst, n, (int)ns);
        else hipLaunchKernelGGL((k_copy_streams<true>), dim3(wgs), b, 0, ctx->stream, (const u4v*)src, (u4v*)dst, n, (int)ns);
        break;
    }
    default:
        if (variant >= 100 && variant < 400) {
            // 100 + 100 * mode + 10 * ui + 0: U = 2 / 4 / 8 / 16 (ui 0..3); the gate's half period comes in `grid` bits 16..31 (ticks of 10 ns)
            const int mode = (variant - 100) / 100, ui = ((variant - 100) % 100) / 10;
            const unsigned half = (unsigned)grid >> 16;
            const dim3 gg((unsigned)grid & 0xffffu);
            const u4v* sv = (const u4v*)src;
            u4v* dv = (u4v*)dst;
            if (mode == 2 && half == 0) return RCV_ERR_ARG;
#define RCV_PH(UI, U, M) if (ui == UI && mode == M) { hipLaunchKernelGGL((k_copy_phase<U, M>), gg, b, 0, ctx->stream, sv, dv, n, half ? half : 1u); return rcv_launch_check(ctx); }
            RCV_PH(0, 2, 0) RCV_PH(1, 4, 0) RCV_PH(2, 8, 0) RCV_PH(3, 16, 0)
            RCV_PH(0, 2, 1) RCV_PH(1, 4, 1) RCV_PH(2, 8, 1) RCV_PH(3, 16, 1)
            RCV_PH(0, 2, 2) RCV_PH(1, 4, 2) RCV_PH(2, 8, 2) RCV_PH(3, 16, 2)
#undef RCV_PH
            return RCV_ERR_ARG;
        }
        if (variant >= 10 && variant < 22) {
            const int nt = (variant - 10) & 3, ui = (variant - 10) >> 2;   // nt bit 0: loads, bit 1: stores; ui 0 / 1 / 2: 4 / 8 / 2 accesses in flight
            const u4v* sv = (const u4v*)src;
            u4v* dv = (u4v*)dst;
#define RCV_MB_CASE(U, N, NL, NS) if (ui == (U == 4 ? 0 : (U == 8 ? 1 : 2)) && nt == N) { hipLaunchKernelGGL((k_copy_sweep_u<U, NL, NS>), g, b, 0, ctx->stream, sv, dv, n); break; }
            RCV_MB_CASE(4, 0, false, false) RCV_MB_CASE(4, 1, true, false) RCV_MB_CASE(4, 2, false, true) RCV_MB_CASE(4, 3, true, true)
            RCV_MB_CASE(8, 0, false, false) RCV_MB_CASE(8, 1, true, false) RCV_MB_CASE(8, 2, false, true) RCV_MB_CASE(8, 3, true, true)
            RCV_MB_CASE(2, 0, false, false) RCV_MB_CASE(2, 1, true, false) RCV_MB_CASE(2, 2, false, true) RCV_MB_CASE(2, 3, true, true)
#undef RCV_MB_CASE
        }
        return RCV_ERR_ARG;
    }
    return rcv_launch_check(ctx);
}
