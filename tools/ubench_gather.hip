// microbenchmark: cost of per-lane gathers from an L1/L2-resident buffer, by vector width / alignment / lane stride
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
struct u3 { uint32_t a, b, c; };
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
template <class T, int ALIGN>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ buf, uint32_t* out, int stride, int iters, uint32_t mask)
{
    const int lane = threadIdx.x;
    uint32_t acc = 0;
    uint32_t off = (uint32_t)(lane * stride + blockIdx.x * 64);
#pragma unroll 8
    for (int i = 0; i < iters; ++i) {
        const uint32_t o = (off & mask) & ~(uint32_t)(ALIGN - 1);
        T v;
        __builtin_memcpy(&v, __builtin_assume_aligned(buf + o, ALIGN), sizeof(T));
        if constexpr (sizeof(T) == 4) acc ^= *(uint32_t*)&v;
        else { const uint32_t* p = (const uint32_t*)&v; for (unsigned j = 0; j < sizeof(T) / 4; ++j) acc ^= p[j]; }
        off += 7919u * 4u + (acc & 0);   // new window each iteration (keeps loads independent)
    }
    out[blockIdx.x * 256 + lane] = acc;
}
template <class T, int ALIGN>
void run(const char* name, const uint8_t* buf, uint32_t* out, int stride)
{
    const int iters = 4096, blocks = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        k<T, ALIGN><<<blocks, 256>>>(buf, out, stride, iters, (1u << 16) - 64);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per CU: blocks*4 waves*iters / 256 CUs ; cycles at 2.3 GHz
    const double winstr_per_cu = (double)blocks * 4 * iters / 256.0;
    printf("%-28s stride %3d: %.3f ms  -> %.1f cycles per wave load instruction per CU (at 2.3 GHz)\n", name, stride, ms, ms * 1e-3 * 2.3e9 / winstr_per_cu);
}
int main()
{
    uint8_t* buf; uint32_t* out;
    hipMalloc(&buf, 1 << 20); hipMemset(buf, 1, 1 << 20); hipMalloc(&out, 1 << 22);
    for (int warm = 0; warm < 3; ++warm) run<uint32_t, 4>("warm", buf, out, 4);
    for (int stride : {3, 4, 12, 16, 64}) {
        run<uint32_t, 4>("dword   align4", buf, out, stride);
        run<u2, 4>("dwordx2 align4", buf, out, stride);
        run<u2, 8>("dwordx2 align8", buf, out, stride);
        run<u3, 4>("dwordx3 align4", buf, out, stride);
        run<u3, 16>("dwordx3 align16", buf, out, stride);
        run<u4, 4>("dwordx4 align4", buf, out, stride);
        run<u4, 8>("dwordx4 align8", buf, out, stride);
        run<u4, 16>("dwordx4 align16", buf, out, stride);
    }
    return 0;
}
