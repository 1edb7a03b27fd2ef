// rcv_pointwise.hip -- colour conversion (a1, a2, a4, a5), rectangle (a3), bgr2gray and the
// synthetic frame generator.  gfx950 only.
//
// Reference semantics followed (file:line under /root/reference):
//   yuyv_to_bgr   rustcv/src/videoio/mod.rs:344-382 ; twin rustcv-camera/src/decode.rs:160-191
//   bgra_to_bgr   rustcv/src/videoio/mod.rs:385-399 ; twin rustcv-camera/src/decode.rs:200-207
//   rgb_to_bgr    rustcv-camera/src/decode.rs:213-219
//   rectangle     rustcv/src/imgproc/drawing.rs:67-106
// The three conversions treat the frame as a FLAT byte array (row stride ignored), exactly as
// the reference does; the kernels are therefore 1-D streaming kernels: every lane moves whole
// 16-byte vectors (HBM-bound, 5 / 7 / 6 algorithmic bytes per pixel).
#include "rcv_internal.h"
#include "rcv_device_utils.h"

namespace {

constexpr int kBlock = 256;

// one macropixel [Y0 U Y1 V] (little-endian dword) -> the six pre-shift sums b0 g0 r0 b1 g1 r1
__device__ __forceinline__ void yuyv_pair(uint32_t m, int* o)
{
    int y0 = (int)(m & 0xff), u = (int)((m >> 8) & 0xff) - 128;
    int y1 = (int)((m >> 16) & 0xff), v = (int)(m >> 24) - 128;
    int c0 = 298 * (y0 - 16) + 128, c1 = 298 * (y1 - 16) + 128;
    int db = 516 * u, dg = -100 * u - 208 * v, dr = 409 * v;
    o[0] = c0 + db;
    o[1] = c0 + dg;
    o[2] = c0 + dr;
    o[3] = c1 + db;
    o[4] = c1 + dg;
    o[5] = c1 + dr;
}

// pairs macropixels per frame; fast path: 8 macropixels (32 B in, 48 B out) per thread.
__global__ __launch_bounds__(kBlock) void k_yuyv2bgr_vec(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                         size_t sfs, size_t dfs, size_t groups)
{
    const uint8_t* s = src + (size_t)blockIdx.y * sfs;
    uint8_t* d = dst + (size_t)blockIdx.y * dfs;
    for (size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += (size_t)gridDim.x * kBlock) {
        const uint4* sp = (const uint4*)(s + g * 32);
        uint4 a = sp[0], b = sp[1];
        uint32_t m[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        int by[48];
#pragma unroll
        for (int i = 0; i < 8; ++i) yuyv_pair(m[i], by + 6 * i);
        uint32_t w[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) w[i] = rcv_ashr_sat_pk4(by[4 * i], by[4 * i + 1], by[4 * i + 2], by[4 * i + 3], 8);
        uint4* dp = (uint4*)(d + g * 48);
        dp[0] = make_uint4(w[0], w[1], w[2], w[3]);
        dp[1] = make_uint4(w[4], w[5], w[6], w[7]);
        dp[2] = make_uint4(w[8], w[9], w[10], w[11]);
    }
}

// scalar path: one macropixel per thread, starting at macropixel `first`
__global__ __launch_bounds__(kBlock) void k_yuyv2bgr_scalar(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                            size_t sfs, size_t dfs, size_t first, size_t pairs)
{
    const uint8_t* s = src + (size_t)blockIdx.y * sfs;
    uint8_t* d = dst + (size_t)blockIdx.y * dfs;
    for (size_t i = first + (size_t)blockIdx.x * kBlock + threadIdx.x; i < pairs; i += (size_t)gridDim.x * kBlock) {
        const uint8_t* sp = s + i * 4;
        uint32_t m = (uint32_t)sp[0] | ((uint32_t)sp[1] << 8) | ((uint32_t)sp[2] << 16) | ((uint32_t)sp[3] << 24);
        int o[6];
        yuyv_pair(m, o);
        uint8_t* dp = d + i * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) dp[k] = (uint8_t)rcv_ashr_sat1(o[k], 8);
    }
}

// BGRA -> BGR: 16 pixels (64 B in, 48 B out) per thread.  v_perm_b32 packs 4 px -> 3 dwords.
__device__ __forceinline__ void pack4_drop_alpha(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t* o)
{
    // __builtin_amdgcn_perm(hi, lo, sel): byte i of result = byte sel[i] of {hi:lo} (lo = bytes 0-3, hi = 4-7)
    o[0] = __builtin_amdgcn_perm(p1, p0, 0x04020100u); // b0 g0 r0 b1
    o[1] = __builtin_amdgcn_perm(p2, p1, 0x05040201u); // g1 r1 b2 g2
    o[2] = __builtin_amdgcn_perm(p3, p2, 0x06050402u); // r2 b3 g3 r3
}

__global__ __launch_bounds__(kBlock) void k_bgra2bgr_vec(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                         size_t sfs, size_t dfs, size_t groups)
{
    const uint8_t* s = src + (size_t)blockIdx.y * sfs;
    uint8_t* d = dst + (size_t)blockIdx.y * dfs;
    for (size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += (size_t)gridDim.x * kBlock) {
        const uint4* sp = (const uint4*)(s + g * 64);
        uint4 a = sp[0], b = sp[1], c = sp[2], e = sp[3];
        uint32_t w[12];
        pack4_drop_alpha(a.x, a.y, a.z, a.w, w);
        pack4_drop_alpha(b.x, b.y, b.z, b.w, w + 3);
        pack4_drop_alpha(c.x, c.y, c.z, c.w, w + 6);
        pack4_drop_alpha(e.x, e.y, e.z, e.w, w + 9);
        uint4* dp = (uint4*)(d + g * 48);
        dp[0] = make_uint4(w[0], w[1], w[2], w[3]);
        dp[1] = make_uint4(w[4], w[5], w[6], w[7]);
        dp[2] = make_uint4(w[8], w[9], w[10], w[11]);
    }
}

__global__ __launch_bounds__(kBlock) void k_bgra2bgr_scalar(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                            size_t sfs, size_t dfs, size_t first, size_t npx)
{
    const uint8_t* s = src + (size_t)blockIdx.y * sfs;
    uint8_t* d = dst + (size_t)blockIdx.y * dfs;
    for (size_t i = first + (size_t)blockIdx.x * kBlock + threadIdx.x; i < npx; i += (size_t)gridDim.x * kBlock) {
        d[3 * i + 0] = s[4 * i + 0];
        d[3 * i + 1] = s[4 * i + 1];
        d[3 * i + 2] = s[4 * i + 2];
    }
}

// RGB -> BGR: 4 px = 3 dwords in, 3 dwords out
__device__ __forceinline__ void swap_rb4(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t* o)
{
    // in bytes: r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3 ; out: b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
    o[0] = __builtin_amdgcn_perm(d1, d0, 0x05000102u); // b0 g0 r0 b1  (bytes 2,1,0 of d0 ; byte1 of d1)
    // middle dword needs bytes from three dwords: g1(d1.0) r1(d0.3) b2(d2.0) g2(d1.3)
    uint32_t t = __builtin_amdgcn_perm(d0, d1, 0x03000700u); // {hi=d0, lo=d1}: g1(lo0) r1(hi3) x g2(lo3)
    o[1] = (t & 0xff00ffffu) | ((d2 & 0xffu) << 16);
    // last dword: r2(d1.2) b3(d2.3) g3(d2.2) r3(d2.1)
    o[2] = __builtin_amdgcn_perm(d2, d1, 0x05060702u);
}

__global__ __launch_bounds__(kBlock) void k_rgb2bgr_vec(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                        size_t sfs, size_t dfs, size_t groups)
{
    const uint8_t* s = src + (size_t)blockIdx.y * sfs;
    uint8_t* d = dst + (size_t)blockIdx.y * dfs;
    for (size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += (size_t)gridDim.x * kBlock) {
        const uint4* sp = (const uint4*)(s + g * 48);
        uint4 a = sp[0], b = sp[1], c = sp[2];
        uint32_t in[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
        uint32_t w[12];
#pragma unroll
        for (int i = 0; i < 4; ++i) swap_rb4(in[3 * i], in[3 * i + 1], in[3 * i + 2], w + 3 * i);
        uint4* dp = (uint4*)(d + g * 48);
        dp[0] = make_uint4(w[0], w[1], w[2], w[3]);
        dp[1] = make_uint4(w[4], w[5], w[6], w[7]);
        dp[2] = make_uint4(w[8], w[9], w[10], w[11]);
    }
}

__global__ __launch_bounds__(kBlock) void k_rgb2bgr_scalar(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                           size_t sfs, size_t dfs, size_t first, size_t npx)
{
    const uint8_t* s = src + (size_t)blockIdx.y * sfs;
    uint8_t* d = dst + (size_t)blockIdx.y * dfs;
    for (size_t i = first + (size_t)blockIdx.x * kBlock + threadIdx.x; i < npx; i += (size_t)gridDim.x * kBlock) {
        uint8_t r = s[3 * i], g = s[3 * i + 1], b = s[3 * i + 2];
        d[3 * i + 0] = b;
        d[3 * i + 1] = g;
        d[3 * i + 2] = r;
    }
}

// ---- "next" rows f2 / f4: display swizzle, codec swizzle, stride-aware YUV 4:2:2 / NV12 ---------------------------
// BGR (flat) -> u32 0x00RRGGBB == bytes B G R 0 (rustcv/src/highgui/mod.rs:125-141); pixels >= nconv are zero
__global__ __launch_bounds__(kBlock) void k_bgr2bgrx(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t sfs, size_t dfs,
                                                     size_t nconv, size_t npx, int vec)
{
    const uint8_t* s = src + (size_t)blockIdx.y * sfs;
    uint8_t* d = dst + (size_t)blockIdx.y * dfs;
    const size_t quads = vec ? nconv / 4 : 0;
    for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < quads; q += (size_t)gridDim.x * kBlock) {
        const uint32_t* sp = (const uint32_t*)(s + q * 12);
        const uint32_t d0 = sp[0], d1 = sp[1], d2 = sp[2];
        uint4 o;
        o.x = d0 & 0x00ffffffu;
        o.y = __builtin_amdgcn_perm(d1, d0, 0x0c050403u);  // b1(d0.3) g1(d1.0) r1(d1.1) 0
        o.z = __builtin_amdgcn_perm(d2, d1, 0x0c040302u);  // b2(d1.2) g2(d1.3) r2(d2.0) 0
        o.w = d2 >> 8;
        *(uint4*)(d + q * 16) = o;
    }
    for (size_t i = quads * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x; i < npx; i += (size_t)gridDim.x * kBlock) {
        uint32_t v = 0;
        if (i < nconv) v = (uint32_t)s[3 * i] | ((uint32_t)s[3 * i + 1] << 8) | ((uint32_t)s[3 * i + 2] << 16);
        d[4 * i] = (uint8_t)v;
        d[4 * i + 1] = (uint8_t)(v >> 8);
        d[4 * i + 2] = (uint8_t)(v >> 16);
        d[4 * i + 3] = 0;
    }
}

// BGR rows (strided) -> packed RGB (rustcv/src/imgcodecs/mod.rs:51-63)
__global__ __launch_bounds__(kBlock) void k_bgr2rgb_rows(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t sstep,
                                                         size_t sfs, size_t dfs, int cols, int vec)
{
    const int y = blockIdx.y;
    const uint8_t* s = src + (size_t)blockIdx.z * sfs + (size_t)y * sstep;
    uint8_t* d = dst + (size_t)blockIdx.z * dfs + (size_t)y * cols * 3;
    // vec = 0: rows of any alignment: the quad's 12 source bytes as the aligned dwords that contain them + v_alignbyte (the row's
    // misalignment is the same for all its quads), the result as one unaligned 12-byte store
    const int quads = cols / 4;
    const unsigned mis = vec ? 0u : (unsigned)((uintptr_t)s & 3);
    for (int q = blockIdx.x * kBlock + threadIdx.x; q < quads; q += gridDim.x * kBlock) {
        const uint32_t* sp = (const uint32_t*)(s + (size_t)q * 12 - mis);
        const uint32_t e0 = sp[0], e1 = sp[1], e2 = sp[2], e3 = sp[mis ? 3 : 2];
        uint32_t w[3];
        swap_rb4(__builtin_amdgcn_alignbyte(e1, e0, mis), __builtin_amdgcn_alignbyte(e2, e1, mis), __builtin_amdgcn_alignbyte(e3, e2, mis), w);
        typedef uint32_t u3m __attribute__((ext_vector_type(3), aligned(1)));
        *(u3m*)(d + (size_t)q * 12) = u3m{w[0], w[1], w[2]};
    }
    for (int x = quads * 4 + blockIdx.x * kBlock + threadIdx.x; x < cols; x += gridDim.x * kBlock) {
        d[3 * x] = s[3 * x + 2];
        d[3 * x + 1] = s[3 * x + 1];
        d[3 * x + 2] = s[3 * x];
    }
}

// strided BGRA rows -> strided BGR rows: what bgra_to_bgr (rustcv/src/videoio/mod.rs:385-399) computes per pixel, but
// honouring the row stride the capture backends report (rustcv-backend-avf/src/stream.rs:250-254 delivers BGRA with the
// CVPixelBuffer's bytes_per_row; the reference helper ignores it).  4 px (16 B in, 12 B out) per thread.
__global__ __launch_bounds__(kBlock) void k_bgra2bgr_rows(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t sstep, size_t dstep,
                                                          size_t sfs, size_t dfs, int cols, int vec)
{
    const int y = blockIdx.y;
    const uint8_t* s = src + (size_t)blockIdx.z * sfs + (size_t)y * sstep;
    uint8_t* d = dst + (size_t)blockIdx.z * dfs + (size_t)y * dstep;
    const int quads = vec ? cols / 4 : 0;
    for (int q = blockIdx.x * kBlock + threadIdx.x; q < quads; q += gridDim.x * kBlock) {
        const uint4 a = *(const uint4*)(s + (size_t)q * 16);
        uint32_t w[3];
        pack4_drop_alpha(a.x, a.y, a.z, a.w, w);
        struct U3 { uint32_t a, b, c; };
        *(U3*)(d + (size_t)q * 12) = U3{w[0], w[1], w[2]};
    }
    for (int x = quads * 4 + blockIdx.x * kBlock + threadIdx.x; x < cols; x += gridDim.x * kBlock) {
        d[3 * x] = s[4 * x];
        d[3 * x + 1] = s[4 * x + 1];
        d[3 * x + 2] = s[4 * x + 2];
    }
}

__device__ __forceinline__ void yuv3(int y, int u, int v, int* o)
{
    const int c = 298 * (y - 16) + 128;
    o[0] = c + 516 * u;
    o[1] = c - 100 * u - 208 * v;
    o[2] = c + 409 * v;
}

// YUYV / UYVY rows (strided) -> BGR rows (strided); 2 macropixels (8 B -> 12 B) per thread on the vector path
__global__ __launch_bounds__(kBlock) void k_yuv422_rows(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t sstep, size_t dstep,
                                                        size_t sfs, size_t dfs, int cols, int uyvy, int vec)
{
    const int y = blockIdx.y;
    const uint8_t* s = src + (size_t)blockIdx.z * sfs + (size_t)y * sstep;
    uint8_t* d = dst + (size_t)blockIdx.z * dfs + (size_t)y * dstep;
    const int pairs = cols / 2;
    const int duo = vec ? pairs / 2 : 0;
    for (int q = blockIdx.x * kBlock + threadIdx.x; q < duo; q += gridDim.x * kBlock) {
        const uint2 m = *(const uint2*)(s + (size_t)q * 8);
        int o[12];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t w = h ? m.y : m.x;
            const int b0 = w & 0xff, b1 = (w >> 8) & 0xff, b2 = (w >> 16) & 0xff, b3 = w >> 24;
            const int y0 = uyvy ? b1 : b0, u = (uyvy ? b0 : b1) - 128, y1 = uyvy ? b3 : b2, v = (uyvy ? b2 : b3) - 128;
            yuv3(y0, u, v, o + 6 * h);
            yuv3(y1, u, v, o + 6 * h + 3);
        }
        struct U3 { uint32_t a, b, c; };
        *(U3*)(d + (size_t)q * 12) = U3{rcv_ashr_sat_pk4(o[0], o[1], o[2], o[3], 8), rcv_ashr_sat_pk4(o[4], o[5], o[6], o[7], 8),
                                       rcv_ashr_sat_pk4(o[8], o[9], o[10], o[11], 8)};
    }
    for (int i = duo * 2 + blockIdx.x * kBlock + threadIdx.x; i < pairs; i += gridDim.x * kBlock) {
        const uint8_t* p = s + (size_t)i * 4;
        const int y0 = uyvy ? p[1] : p[0], u = (uyvy ? p[0] : p[1]) - 128, y1 = uyvy ? p[3] : p[2], v = (uyvy ? p[2] : p[3]) - 128;
        int o[6];
        yuv3(y0, u, v, o);
        yuv3(y1, u, v, o + 3);
#pragma unroll
        for (int k = 0; k < 6; ++k) d[(size_t)i * 6 + k] = (uint8_t)rcv_ashr_sat1(o[k], 8);
    }
}

// NV12 (luma rows then interleaved chroma rows, one step) -> BGR rows; 4 px (4 B luma + 4 B chroma -> 12 B) per thread
__global__ __launch_bounds__(kBlock) void k_nv12_rows(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t sstep, size_t dstep,
                                                      size_t sfs, size_t dfs, int rows, int cols, int vec)
{
    const int y = blockIdx.y;
    const uint8_t* yr = src + (size_t)blockIdx.z * sfs + (size_t)y * sstep;
    const uint8_t* uv = src + (size_t)blockIdx.z * sfs + (size_t)rows * sstep + (size_t)(y / 2) * sstep;
    uint8_t* d = dst + (size_t)blockIdx.z * dfs + (size_t)y * dstep;
    const int quads = vec ? cols / 4 : 0;
    for (int q = blockIdx.x * kBlock + threadIdx.x; q < quads; q += gridDim.x * kBlock) {
        const uint32_t yy = *(const uint32_t*)(yr + (size_t)q * 4), cc = *(const uint32_t*)(uv + (size_t)q * 4);
        const int u0 = (int)(cc & 0xff) - 128, v0 = (int)((cc >> 8) & 0xff) - 128, u1 = (int)((cc >> 16) & 0xff) - 128, v1 = (int)(cc >> 24) - 128;
        int o[12];
        yuv3(yy & 0xff, u0, v0, o);
        yuv3((yy >> 8) & 0xff, u0, v0, o + 3);
        yuv3((yy >> 16) & 0xff, u1, v1, o + 6);
        yuv3(yy >> 24, u1, v1, o + 9);
        struct U3 { uint32_t a, b, c; };
        *(U3*)(d + (size_t)q * 12) = U3{rcv_ashr_sat_pk4(o[0], o[1], o[2], o[3], 8), rcv_ashr_sat_pk4(o[4], o[5], o[6], o[7], 8),
                                       rcv_ashr_sat_pk4(o[8], o[9], o[10], o[11], 8)};
    }
    for (int x = quads * 4 + blockIdx.x * kBlock + threadIdx.x; x < cols; x += gridDim.x * kBlock) {
        const uint8_t* c2 = uv + (size_t)(x / 2) * 2;
        int o[3];
        yuv3(yr[x], (int)c2[0] - 128, (int)c2[1] - 128, o);
#pragma unroll
        for (int k = 0; k < 3; ++k) d[(size_t)x * 3 + k] = (uint8_t)rcv_ashr_sat1(o[k], 8);
    }
}

// BGR -> gray, stride-aware.  Fast: 4 px (3 dwords) -> 1 dword per thread.
__device__ __forceinline__ uint32_t gray1(uint32_t b, uint32_t g, uint32_t r)
{
    return (1868u * b + 9617u * g + 4899u * r + 8192u) >> 14;
}

__global__ __launch_bounds__(kBlock) void k_bgr2gray(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                     size_t sstep, size_t dstep, size_t sfs, size_t dfs,
                                                     int rows, int cols, int vec)
{
    int y = blockIdx.y;
    const uint8_t* s = src + (size_t)blockIdx.z * sfs + (size_t)y * sstep;
    uint8_t* d = dst + (size_t)blockIdx.z * dfs + (size_t)y * dstep;
    // vec = 0: rows of any alignment (an odd width of a packed image: step = cols * 3).  A row's misalignment is the same for all
    // of its quads, so the 12 bytes of a quad are fetched as the ALIGNED dwords that contain them and shifted into place
    // (unaligned per-lane loads would serialise in the address path); the four gray bytes go out as one unaligned dword store.
    const int quads = cols / 4;
    const unsigned mis = vec ? 0u : (unsigned)((uintptr_t)s & 3);
    for (int q = blockIdx.x * kBlock + threadIdx.x; q < quads; q += gridDim.x * kBlock) {
        const uint32_t* sp = (const uint32_t*)(s + (size_t)q * 12 - mis);
        const uint32_t e0 = sp[0], e1 = sp[1], e2 = sp[2], e3 = sp[mis ? 3 : 2];   // (an aligned quad needs no fourth dword: never read past it)
        const uint32_t d0 = __builtin_amdgcn_alignbyte(e1, e0, mis), d1 = __builtin_amdgcn_alignbyte(e2, e1, mis), d2 = __builtin_amdgcn_alignbyte(e3, e2, mis);
        uint32_t g0 = gray1(d0 & 0xff, (d0 >> 8) & 0xff, (d0 >> 16) & 0xff);
        uint32_t g1 = gray1(d0 >> 24, d1 & 0xff, (d1 >> 8) & 0xff);
        uint32_t g2 = gray1((d1 >> 16) & 0xff, d1 >> 24, d2 & 0xff);
        uint32_t g3 = gray1((d2 >> 8) & 0xff, (d2 >> 16) & 0xff, d2 >> 24);
        typedef uint32_t u1m __attribute__((aligned(1)));
        *(u1m*)(d + (size_t)q * 4) = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    }
    for (int x = quads * 4 + blockIdx.x * kBlock + threadIdx.x; x < cols; x += gridDim.x * kBlock)
        d[x] = (uint8_t)gray1(s[3 * x], s[3 * x + 1], s[3 * x + 2]);
}

// BGR -> gray, 16 pixels per thread: three aligned 16-byte loads, one 16-byte store, gray by two v_dot4_u32_u8 per pixel.
// weights 1868, 9617, 4899 = 256*{7,37,19} + {76,145,35}: hi = 7B+37G+19R, lo = 76B+145G+35R+8192, g = ((hi<<8)+lo)>>14
// (identical integer result to gray1()).  Rows must be 16-byte aligned and cols a multiple of 16.
__global__ __launch_bounds__(kBlock) void k_bgr2gray16(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t sstep, size_t dstep,
                                                       size_t sfs, size_t dfs, int cols)
{
    const int y = blockIdx.y;
    const uint8_t* s = src + (size_t)blockIdx.z * sfs + (size_t)y * sstep;
    uint8_t* d = dst + (size_t)blockIdx.z * dfs + (size_t)y * dstep;
    const int groups = cols / 16;
    for (int g = blockIdx.x * kBlock + threadIdx.x; g < groups; g += gridDim.x * kBlock) {
        const uint4* sp = (const uint4*)(s + (size_t)g * 48);
        const uint4 a = sp[0], b = sp[1], c = sp[2];
        const uint32_t w[13] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, 0u};
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k0 = 3 * (4 * q + j), w0 = k0 >> 2, sh = k0 & 3;   // pixel = bytes k0..k0+2 of the 48
                const uint32_t px = sh == 0 ? w[w0] : __builtin_amdgcn_alignbyte(w[w0 + 1], w[w0], sh);
                const uint32_t hi8 = __builtin_amdgcn_udot4(px, 0x00132507u, 0u, false);
                const uint32_t lo8 = __builtin_amdgcn_udot4(px, 0x0023914cu, 8192u, false);
                acc |= (((hi8 << 8) + lo8) >> 14) << (8 * j);
            }
            o[q] = acc;
        }
        *(uint4*)(d + (size_t)g * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// rectangle: one thread per (perimeter position, t).  Every write stores the same colour, so
// overdraw order is irrelevant.  Index math is the reference's: 64-bit wrapping idx, guard idx+2 < len.
__device__ __forceinline__ void set_pixel(uint8_t* data, size_t len, size_t step, int r, int c,
                                          uint8_t b, uint8_t g, uint8_t rr)
{
    size_t idx = (size_t)(long long)r * step + (size_t)(long long)c * 3u;
    if (idx + 2 < len && idx + 2 >= 2) {
        data[idx] = b;
        data[idx + 1] = g;
        data[idx + 2] = rr;
    }
}

__global__ __launch_bounds__(kBlock) void k_rectangle(uint8_t* __restrict__ base, size_t fs, size_t cap, size_t step,
                                                      int x_min, int y_min, int x_max, int y_max, long long thick,
                                                      uint8_t b, uint8_t g, uint8_t r)
{
    uint8_t* data = base + (size_t)blockIdx.y * fs;
    long long W = x_max - x_min, H = y_max - y_min;
    long long total = (W + H) * thick;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        long long pos = i / thick;
        int t = (int)(i - pos * thick);
        if (pos < W) {
            int c = x_min + (int)pos;
            set_pixel(data, cap, step, y_min + t, c, b, g, r);
            set_pixel(data, cap, step, y_max - 1 - t, c, b, g, r);
        } else {
            int rr = y_min + (int)(pos - W);
            set_pixel(data, cap, step, rr, x_min + t, b, g, r);
            set_pixel(data, cap, step, rr, x_max - 1 - t, b, g, r);
        }
    }
}

// Degenerate rectangles (thickness larger than the clipped rect, on a Mat whose step is not a
// multiple of 3): wrapped pixels land off the 3-byte grid, writes of different channels overlap
// and the reference's SEQUENTIAL order decides each byte.  One thread per frame replays that order.
__global__ void k_rectangle_serial(uint8_t* __restrict__ base, size_t fs, size_t cap, size_t step,
                                   int x_min, int y_min, int x_max, int y_max, long long thick,
                                   uint8_t b, uint8_t g, uint8_t r)
{
    if (threadIdx.x != 0) return;
    uint8_t* data = base + (size_t)blockIdx.x * fs;
    for (int c = x_min; c < x_max; ++c)
        for (long long t = 0; t < thick; ++t) {
            set_pixel(data, cap, step, y_min + (int)t, c, b, g, r);
            set_pixel(data, cap, step, y_max - 1 - (int)t, c, b, g, r);
        }
    for (int rr = y_min; rr < y_max; ++rr)
        for (long long t = 0; t < thick; ++t) {
            set_pixel(data, cap, step, rr, x_min + (int)t, b, g, r);
            set_pixel(data, cap, step, rr, x_max - 1 - (int)t, b, g, r);
        }
}

// ---- synthetic frames; same counter-based definition as oracle/rcv_oracle.c ----
__device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t synth_noise(uint64_t seed, uint64_t frame, int y, int x, int c)
{
    uint64_t ctr = (frame << 40) + ((uint64_t)y << 20) + ((uint64_t)x << 2) + (uint64_t)c;
    return (uint32_t)(splitmix64(seed ^ ctr) >> 56);
}

__global__ __launch_bounds__(kBlock) void k_synth(uint8_t* __restrict__ base, size_t fs, size_t step, int rows, int cols,
                                                  int ch, int family, uint64_t seed, uint64_t frame_base)
{
    int y = blockIdx.y;
    uint64_t frame = frame_base + blockIdx.z;
    uint8_t* row = base + (size_t)blockIdx.z * fs + (size_t)y * step;
    int mx = cols - 200 > 1 ? cols - 200 : 1, my = rows - 200 > 1 ? rows - 200 : 1;
    int sx0 = (int)((frame * 37u) % (uint64_t)mx), sy0 = (int)((frame * 23u) % (uint64_t)my);
    for (int x = blockIdx.x * kBlock + threadIdx.x; x < cols; x += gridDim.x * kBlock) {
        if (family == RCV_SYNTH_YUYV) {
            row[2 * x] = (uint8_t)synth_noise(seed, frame, y, x, 0);
            row[2 * x + 1] = (uint8_t)synth_noise(seed ^ 0x9E3779B97F4A7C15ull, frame, y, x >> 1, (x & 1) ? 3 : 1);
            continue;
        }
        for (int c = 0; c < ch; ++c) {
            uint32_t n = synth_noise(seed, frame, y, x, c);
            uint32_t v = n;
            if (family == RCV_SYNTH_SCENE) {
                uint32_t q = n >> 2;
                if (x >= sx0 && x < sx0 + 200 && y >= sy0 && y < sy0 + 200) v = 255 - q;
                else v = q + (uint32_t)(((long long)x * 96) / cols) + ((((x >> 6) + (y >> 6)) & 1) ? 64u : 0u);
            }
            row[(size_t)x * ch + c] = (uint8_t)v;
        }
    }
}

inline unsigned grid1d(size_t work_items)
{
    size_t b = (work_items + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > 16384) b = 16384; // grid-stride beyond 16k blocks
    return (unsigned)b;
}

bool aligned16(const void* p, size_t fs, int n) { return ((uintptr_t)p % 16 == 0) && (n <= 1 || fs % 16 == 0); }

} // namespace

// ------------------------------------------------------------------------------------------------
// entry points
// ------------------------------------------------------------------------------------------------

static int cvt_flat(rcv_ctx* ctx, int code, const rcv_batch* src, rcv_batch* dst)
{
    const rcv_mat* sm = &src->frame0;
    rcv_mat* dm = &dst->frame0;
    if (sm->device != RCV_DEVICE || dm->device != RCV_DEVICE) return RCV_ERR_ARG;
    if (src->n != dst->n || src->n < 0) return RCV_ERR_ARG;
    if (sm->depth != RCV_8U || dm->depth != RCV_8U) return RCV_ERR_UNSUPPORTED;
    if (dm->rows < 0 || dm->cols < 0) return RCV_ERR_ARG;
    int n = src->n;
    size_t w = (size_t)dm->cols, h = (size_t)dm->rows;
    size_t slen = sm->cap, dlen = dm->cap;
    if (n > 1 && (src->frame_stride < slen || dst->frame_stride < dlen)) return RCV_ERR_SIZE;
    size_t units = 0; // macropixels or pixels to convert
    switch (code) {
    case RCV_YUYV2BGR: // rustcv/src/videoio/mod.rs:345-348 : only src is checked
        if (slen < w * h * 2) return RCV_NOOP;
        units = w * h / 2;
        if (dlen < units * 6) return RCV_ERR_SIZE; // reference would panic on the out-of-range index
        break;
    case RCV_YUYV2BGR_TWIN: // rustcv-camera/src/decode.rs:160-167
        units = w * h / 2;
        if (slen < units * 4 || dlen < units * 6) return RCV_NOOP;
        break;
    case RCV_BGRA2BGR: // rustcv/src/videoio/mod.rs:386-390
        units = w * h;
        if (slen < units * 4 || dlen < units * 3) return RCV_NOOP;
        break;
    case RCV_BGRA2BGR_TWIN: { // rustcv-camera/src/decode.rs:200-207 : zip of whole chunks
        size_t a = slen / 4, b = dlen / 3;
        units = a < b ? a : b;
        break;
    }
    case RCV_RGB2BGR: { // rustcv-camera/src/decode.rs:213-219
        size_t a = slen / 3, b = dlen / 3;
        units = a < b ? a : b;
        break;
    }
    default: return RCV_ERR_ARG;
    }
    if (units == 0 || n == 0) return RCV_OK;
    if (!sm->data || !dm->data) return RCV_ERR_ARG;
    if (n > 65535) return RCV_ERR_UNSUPPORTED;   // frames ride on grid.y
    const uint8_t* s = (const uint8_t*)sm->data;
    uint8_t* d = (uint8_t*)dm->data;
    size_t sfs = src->frame_stride, dfs = dst->frame_stride;
    bool vec = aligned16(s, sfs, n) && aligned16(d, dfs, n);
    hipStream_t st = ctx->stream;
    if (code == RCV_YUYV2BGR || code == RCV_YUYV2BGR_TWIN) {
        size_t groups = vec ? units / 8 : 0;
        if (groups) RCV_LAUNCH(k_yuyv2bgr_vec, dim3(grid1d(groups), n), dim3(kBlock), 0, st, s, d, sfs, dfs, groups);
        if (groups * 8 < units)
            RCV_LAUNCH(k_yuyv2bgr_scalar, dim3(grid1d(units - groups * 8), n), dim3(kBlock), 0, st, s, d, sfs, dfs, groups * 8, units);
    } else if (code == RCV_BGRA2BGR || code == RCV_BGRA2BGR_TWIN) {
        size_t groups = vec ? units / 16 : 0;
        // (3 workgroups per CU -- an untouched dynamic-LDS request -- measured 0.625 against 0.661 ms on 64 4K frames)
        if (groups) RCV_LAUNCH(k_bgra2bgr_vec, dim3(grid1d(groups), n), dim3(kBlock), 54272, st, s, d, sfs, dfs, groups);
        if (groups * 16 < units)
            RCV_LAUNCH(k_bgra2bgr_scalar, dim3(grid1d(units - groups * 16), n), dim3(kBlock), 0, st, s, d, sfs, dfs, groups * 16, units);
    } else {
        size_t groups = vec ? units / 16 : 0;
        if (groups) RCV_LAUNCH(k_rgb2bgr_vec, dim3(grid1d(groups), n), dim3(kBlock), 0, st, s, d, sfs, dfs, groups);
        if (groups * 16 < units)
            RCV_LAUNCH(k_rgb2bgr_scalar, dim3(grid1d(units - groups * 16), n), dim3(kBlock), 0, st, s, d, sfs, dfs, groups * 16, units);
    }
    return rcv_launch_check(ctx);
}

static inline bool al(const void* p, size_t step, size_t fs, int n, size_t a) { return (uintptr_t)p % a == 0 && step % a == 0 && (n <= 1 || fs % a == 0); }

static int cvt_gray(rcv_ctx* ctx, const rcv_batch* src, rcv_batch* dst)
{
    View s, d;
    RCV_TRY(rcv_view_batch(src, RCV_8U, &s));
    RCV_TRY(rcv_view_batch(dst, RCV_8U, &d));
    if (s.ch != 3 || d.ch != 1) return RCV_ERR_UNSUPPORTED;
    if (s.rows != d.rows || s.cols != d.cols || s.n != d.n) return RCV_ERR_ARG;
    if (s.rows == 0 || s.cols == 0 || s.n == 0) return RCV_OK;
    if (s.rows > 65535 || s.n > 65535) return RCV_ERR_UNSUPPORTED;   // rows ride on grid.y, frames on grid.z
    int vec = ((uintptr_t)s.p % 4 == 0) && (s.step % 4 == 0) && (s.fstride % 4 == 0) &&
              ((uintptr_t)d.p % 4 == 0) && (d.step % 4 == 0) && (d.fstride % 4 == 0);
    if (s.cols % 16 == 0 && al(s.p, s.step, s.fstride, s.n, 16) && al(d.p, d.step, d.fstride, d.n, 16)) {
        // (4 workgroups per CU measured 0.348 against 0.367 ms on 64 4K frames)
        RCV_LAUNCH(k_bgr2gray16, dim3(grid1d((size_t)s.cols / 16), s.rows, s.n), dim3(kBlock), 40960, ctx->stream, s.p, d.p, s.step, d.step,
                           s.fstride, d.fstride, s.cols);
        return rcv_launch_check(ctx);
    }
    dim3 grid(grid1d((size_t)(s.cols + 3) / 4), s.rows, s.n);
    RCV_LAUNCH(k_bgr2gray, grid, dim3(kBlock), 0, ctx->stream, s.p, d.p, s.step, d.step, s.fstride, d.fstride,
                       s.rows, s.cols, vec);
    return rcv_launch_check(ctx);
}

// f4: BGR (flat) -> BGRX u32 ; f4: BGR rows -> packed RGB ; f2: strided YUYV/UYVY and NV12 -> BGR
static int cvt_next_rows(rcv_ctx* ctx, int code, const rcv_batch* src, rcv_batch* dst)
{
    const rcv_mat* sm = &src->frame0;
    rcv_mat* dm = &dst->frame0;
    if (sm->device != RCV_DEVICE || dm->device != RCV_DEVICE) return RCV_ERR_ARG;
    if (src->n != dst->n || src->n < 0) return RCV_ERR_ARG;
    if (sm->depth != RCV_8U || dm->depth != RCV_8U) return RCV_ERR_UNSUPPORTED;
    const int n = src->n;
    hipStream_t st = ctx->stream;
    if (code == RCV_BGR2BGRX) {
        if (dm->rows < 0 || dm->cols < 0 || dm->channels != 4) return RCV_ERR_ARG;
        const size_t npx = (size_t)dm->rows * dm->cols;
        if (dm->cap < npx * 4) return RCV_ERR_SIZE;
        if (n > 1 && (src->frame_stride < sm->cap || dst->frame_stride < npx * 4)) return RCV_ERR_SIZE;
        const size_t nconv = sm->cap / 3 < npx ? sm->cap / 3 : npx;
        if (npx == 0 || n == 0) return RCV_OK;
        if (!dm->data || (nconv && !sm->data)) return RCV_ERR_ARG;
        const int vec = al(sm->data, 4, src->frame_stride, n, 4) && al(dm->data, 16, dst->frame_stride, n, 16);
        RCV_LAUNCH(k_bgr2bgrx, dim3(grid1d((npx + 3) / 4), n), dim3(kBlock), 0, st, (const uint8_t*)sm->data, (uint8_t*)dm->data,
                           src->frame_stride, dst->frame_stride, nconv, npx, vec);
        return rcv_launch_check(ctx);
    }
    View s;
    RCV_TRY(rcv_view_batch(src, RCV_8U, &s));
    if (s.rows > 65535 || n > 65535) return RCV_ERR_UNSUPPORTED;
    if (code == RCV_BGR2RGB) {
        if (s.ch != 3) return RCV_ERR_UNSUPPORTED;
        const size_t need = (size_t)s.rows * s.cols * 3;
        if (dm->cap < need) return RCV_ERR_SIZE;
        if (n > 1 && dst->frame_stride < need) return RCV_ERR_SIZE;
        if (need == 0 || n == 0) return RCV_OK;
        if (!dm->data) return RCV_ERR_ARG;
        const int vec = al(s.p, s.step, s.fstride, n, 4) && al(dm->data, (size_t)s.cols * 3, dst->frame_stride, n, 4);
        RCV_LAUNCH(k_bgr2rgb_rows, dim3(grid1d((size_t)(s.cols + 3) / 4), s.rows, n), dim3(kBlock), 0, st, s.p, (uint8_t*)dm->data, s.step,
                           s.fstride, dst->frame_stride, s.cols, vec);
        return rcv_launch_check(ctx);
    }
    if (code == RCV_NV12_2BGR) {
        // src describes the luma plane (rows x cols, 1 channel); the chroma plane follows at rows*step
        if (sm->channels != 1 || sm->rows < 0 || sm->cols < 0) return RCV_ERR_ARG;
        const size_t need = sm->step * ((size_t)sm->rows + (size_t)(sm->rows + 1) / 2);
        if (sm->step < (size_t)sm->cols + (sm->cols & 1)) return RCV_ERR_SIZE;
        if (sm->cap < need) return RCV_NOOP;  // convert.rs:56-58: silent return
        View d;
        RCV_TRY(rcv_view_batch(dst, RCV_8U, &d));
        if (d.ch != 3 || d.rows != sm->rows || d.cols != sm->cols) return RCV_ERR_ARG;
        if (n > 1 && src->frame_stride < need) return RCV_ERR_SIZE;
        if (d.rows == 0 || d.cols == 0 || n == 0) return RCV_OK;
        if (!sm->data) return RCV_ERR_ARG;
        const int vec = al(sm->data, sm->step, src->frame_stride, n, 4) && al(d.p, d.step, d.fstride, n, 4);
        RCV_LAUNCH(k_nv12_rows, dim3(grid1d((size_t)(d.cols + 3) / 4), d.rows, n), dim3(kBlock), 0, st, (const uint8_t*)sm->data, d.p,
                           sm->step, d.step, src->frame_stride, d.fstride, d.rows, d.cols, vec);
        return rcv_launch_check(ctx);
    }
    View d;
    RCV_TRY(rcv_view_batch(dst, RCV_8U, &d));
    if (code == RCV_BGRA2BGR_STRIDED) {
        if (s.ch != 4 || d.ch != 3 || s.rows != d.rows || s.cols != d.cols) return RCV_ERR_ARG;
        if (s.rows == 0 || s.cols == 0 || n == 0) return RCV_OK;
        const int vec = al(s.p, s.step, s.fstride, n, 16) && al(d.p, d.step, d.fstride, n, 4);
        RCV_LAUNCH(k_bgra2bgr_rows, dim3(grid1d((size_t)(s.cols + 3) / 4), s.rows, n), dim3(kBlock), 0, st, s.p, d.p, s.step, d.step,
                           s.fstride, d.fstride, s.cols, vec);
        return rcv_launch_check(ctx);
    }
    // RCV_YUYV2BGR_STRIDED / RCV_UYVY2BGR_STRIDED
    if (s.ch != 2 || d.ch != 3 || s.rows != d.rows || s.cols != d.cols) return RCV_ERR_ARG;
    if (s.rows == 0 || s.cols < 2 || n == 0) return RCV_OK;
    const int vec = al(s.p, s.step, s.fstride, n, 8) && al(d.p, d.step, d.fstride, n, 4);
    RCV_LAUNCH(k_yuv422_rows, dim3(grid1d((size_t)(s.cols / 4 + 1)), s.rows, n), dim3(kBlock), 0, st, s.p, d.p, s.step, d.step, s.fstride,
                       d.fstride, s.cols, code == RCV_UYVY2BGR_STRIDED ? 1 : 0, vec);
    return rcv_launch_check(ctx);
}

extern "C" int rcv_cvt_color_batch(rcv_ctx* ctx, int code, const rcv_batch* src, rcv_batch* dst)
{
    if (!src || !dst) return RCV_ERR_ARG;
    RCV_TRY(rcv_bind(ctx));
    if (code == RCV_BGR2GRAY) return cvt_gray(ctx, src, dst);
    if (code >= RCV_BGR2BGRX && code <= RCV_BGRA2BGR_STRIDED) return cvt_next_rows(ctx, code, src, dst);
    return cvt_flat(ctx, code, src, dst);
}

extern "C" int rcv_cvt_color(rcv_ctx* ctx, int code, const rcv_mat* src, rcv_mat* dst)
{
    if (!src || !dst) return RCV_ERR_ARG;
    Stage st;
    RCV_TRY(stage_begin(&st, ctx));
    rcv_mat *ds, *dd;
    RCV_TRY(stage_in(&st, src, true, false, &ds));
    // dst is uploaded too: the conversions may legitimately leave part of it untouched
    RCV_TRY(stage_in(&st, dst, true, true, &dd));
    rcv_batch bs = rcv_single(ds), bd = rcv_single(dd);
    int rc = rcv_cvt_color_batch(ctx, code, &bs, &bd);
    return stage_finish(&st, rc);
}

extern "C" int rcv_rectangle_batch(rcv_ctx* ctx, rcv_batch* mats, int32_t x, int32_t y, int32_t w, int32_t h,
                                   uint8_t b, uint8_t g, uint8_t r, int32_t thickness)
{
    if (!mats) return RCV_ERR_ARG;
    RCV_TRY(rcv_bind(ctx));
    const rcv_mat* m = &mats->frame0;
    if (m->device != RCV_DEVICE || mats->n < 0) return RCV_ERR_ARG;
    if (m->depth != RCV_8U) return RCV_ERR_UNSUPPORTED;
    if (mats->n > 1 && mats->frame_stride < m->cap) return RCV_ERR_SIZE;
    // rustcv/src/imgproc/drawing.rs:68-75
    int32_t x_min = x > 0 ? x : 0, y_min = y > 0 ? y : 0;
    int32_t xe = (int32_t)((uint32_t)x + (uint32_t)w), ye = (int32_t)((uint32_t)y + (uint32_t)h);
    int32_t x_max = xe < m->cols ? xe : m->cols, y_max = ye < m->rows ? ye : m->rows;
    if (x_min >= x_max || y_min >= y_max) return RCV_OK;
    if (thickness <= 0 || mats->n == 0 || m->cap == 0) return RCV_OK;
    if (!m->data) return RCV_ERR_ARG;
    // beyond this many steps inward no index can pass the `idx+2 < len` guard any more
    long long tcap = (long long)(m->cap / 3) + (long long)m->cols + (long long)m->rows + 2;
    long long thick = thickness < tcap ? thickness : tcap;
    long long total = ((long long)(x_max - x_min) + (long long)(y_max - y_min)) * thick;
    // every generated (row, col) inside the Mat -> pixels are disjoint 3-byte cells, order is irrelevant;
    // likewise when step % 3 == 0 (all writes sit on one 3-byte grid and carry the same colour).
    bool in_range = (long long)y_min + thick <= m->rows && (long long)y_max - thick >= 0 &&
                    (long long)x_min + thick <= m->cols && (long long)x_max - thick >= 0;
    if (!in_range && m->step % 3 != 0) {
        RCV_LAUNCH(k_rectangle_serial, dim3(mats->n), dim3(64), 0, ctx->stream, (uint8_t*)m->data, mats->frame_stride,
                           m->cap, m->step, x_min, y_min, x_max, y_max, thick, b, g, r);
        return rcv_launch_check(ctx);
    }
    RCV_LAUNCH(k_rectangle, dim3(grid1d((size_t)total), mats->n), dim3(kBlock), 0, ctx->stream, (uint8_t*)m->data,
                       mats->frame_stride, m->cap, m->step, x_min, y_min, x_max, y_max, thick, b, g, r);
    return rcv_launch_check(ctx);
}

extern "C" int rcv_rectangle(rcv_ctx* ctx, rcv_mat* mat, int32_t x, int32_t y, int32_t w, int32_t h,
                             uint8_t b, uint8_t g, uint8_t r, int32_t thickness)
{
    if (!mat) return RCV_ERR_ARG;
    Stage st;
    RCV_TRY(stage_begin(&st, ctx));
    rcv_mat* dm;
    RCV_TRY(stage_in(&st, mat, true, true, &dm));
    rcv_batch bm = rcv_single(dm);
    int rc = rcv_rectangle_batch(ctx, &bm, x, y, w, h, b, g, r, thickness);
    return stage_finish(&st, rc);
}

extern "C" int rcv_synth_batch(rcv_ctx* ctx, rcv_batch* dst, int family, uint64_t seed, uint64_t frame_base)
{
    if (!dst) return RCV_ERR_ARG;
    RCV_TRY(rcv_bind(ctx));
    if (family != RCV_SYNTH_NOISE && family != RCV_SYNTH_SCENE && family != RCV_SYNTH_YUYV) return RCV_ERR_ARG;
    View d;
    RCV_TRY(rcv_view_batch(dst, RCV_8U, &d));
    if (family == RCV_SYNTH_YUYV && d.ch != 2) return RCV_ERR_ARG;
    if (family != RCV_SYNTH_YUYV && d.ch != 1 && d.ch != 3 && d.ch != 4) return RCV_ERR_UNSUPPORTED;
    if (d.rows == 0 || d.cols == 0 || d.n == 0) return RCV_OK;
    dim3 grid(grid1d((size_t)d.cols), d.rows, d.n);
    RCV_LAUNCH(k_synth, grid, dim3(kBlock), 0, ctx->stream, d.p, d.fstride, d.step, d.rows, d.cols, d.ch, family, seed,
                       frame_base);
    return rcv_launch_check(ctx);
}
