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

} // namespace

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
    case 6: hipLaunchKernelGGL(k_read_sweep, g, b, 0, ctx->stream, s, (uint4*)(ctx->kconst + 49152), n); break;
    case 7: hipLaunchKernelGGL(k_write_sweep, g, b, 0, ctx->stream, d, n, 0x5EEDu); break;
    case 8:
        if (grid % 8) return RCV_ERR_ARG;
        hipLaunchKernelGGL((k_copy_xcd<false, false>), g, b, 0, ctx->stream, s, d, n);
        break;
    case 9:
        if (grid % 8) return RCV_ERR_ARG;
        hipLaunchKernelGGL((k_copy_xcd<true, true>), g, b, 0, ctx->stream, s, d, n);
        break;
    default: return RCV_ERR_ARG;
    }
    return rcv_launch_check(ctx);
}
