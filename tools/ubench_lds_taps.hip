// tools/ubench_lds_taps.hip -- (round 5) cost of the tap reads of an LDS-staged BGR patch: lanes 12 bytes apart (four source pixels), the
// read as one unaligned ds_read_b64, as a dword-aligned ds_read_b96 / ds_read2_b32 + ds_read_b32 / ds_read_b128, or aligned references.
// One workgroup of 256 threads per CU x 4, 8 reads per iteration; prints cycles per wave-level LDS instruction per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, int k0, int stride, int rowjump)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    for (int i = threadIdx.x; i < 6144; i += 256) ((uint32_t*)lds)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lanes `stride` bytes apart; every 8 lanes the row changes (a tilted row piece): + rowjump bytes
    unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)lds + k0 + stride * lane + (lane >> 3) * rowjump + wave * 1040;
    if (MODE != 0) a &= ~3u;
    if (MODE == 4) a &= ~7u;
    if (MODE == 5) a &= ~15u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            uint32_t v0, v1, v2, v3;
            if (MODE == 0 || MODE == 4) { uint64_t v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(r * 1536)); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); acc ^= (uint32_t)v ^ (uint32_t)(v >> 32); }
            if (MODE == 1) { typedef uint32_t u3 __attribute__((ext_vector_type(3))); u3 v; asm volatile("ds_read_b96 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(r * 1536)); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); acc ^= v.x ^ v.y ^ v.z; }
            if (MODE == 2) { uint64_t v; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(a), "n"(r * 8), "n"(r * 8 + 1)); asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v2) : "v"(a), "n"(r * 32 + 8)); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v), "+v"(v2)); acc ^= (uint32_t)v ^ (uint32_t)(v >> 32) ^ v2; }
            if (MODE == 3 || MODE == 5) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(r * 1536)); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
            (void)v0; (void)v1; (void)v3;
        }
        a ^= (it & 1) * 0;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main()
{
    uint32_t* o; hipMalloc(&o, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[6] = {"ds_read_b64 unaligned", "ds_read_b96 dword-aligned", "ds_read2_b32 + ds_read_b32 dword-aligned", "ds_read_b128 dword-aligned", "ds_read_b64 8-aligned", "ds_read_b128 16-aligned"};
    const int iters = 2000;
    for (int stride : {12, 16, 6, 3}) for (int rowjump : {0, 436}) for (int k0 = 0; k0 < 4; ++k0) for (int mode = 0; mode < 6; ++mode) {
        if (mode != 0 && k0 != 0) continue;
        auto go = [&] {
            switch (mode) { case 0: k<0><<<1024, 256, 32768>>>(o, iters, k0, stride, rowjump); break; case 1: k<1><<<1024, 256, 32768>>>(o, iters, k0, stride, rowjump); break;
              case 2: k<2><<<1024, 256, 32768>>>(o, iters, k0, stride, rowjump); break; case 3: k<3><<<1024, 256, 32768>>>(o, iters, k0, stride, rowjump); break;
              case 4: k<4><<<1024, 256, 32768>>>(o, iters, k0, stride, rowjump); break; default: k<5><<<1024, 256, 32768>>>(o, iters, k0, stride, rowjump); } };
        go(); hipDeviceSynchronize();
        hipEventRecord(e0); go(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per CU: 4 workgroups x 4 waves x iters x 8 read groups (mode 2: two instructions per group)
        const double cyc = ms * 1e-3 * 2.3e9 / (4.0 * 4 * iters * 8);
        printf("stride %2d rowjump %3d k0 %d  %-42s %7.3f ms  %6.1f cycles per wave read (per CU, at 2.3 GHz)\n", stride, rowjump, k0, names[mode], ms, cyc);
    }
    return 0;
}
