// microbenchmark (round 5): can the SCALAR cache path pull lines into L2 beside the vector L1?  Every wave walks its own chunk of a
// large buffer with one s_load_dwordx2 per `stride` bytes (a scalar-cache miss fetches the line through L2); the vector variant reads the
// same bytes with coalesced dwordx4 loads.  Rate = bytes covered / time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void k_scalar(const uint8_t* buf, size_t per_wave, int stride, int depth, uint32_t* out)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
    const uint8_t* p = buf + (size_t)wave * per_wave;
    asm volatile("" : "+s"(p));
    const uint8_t* end = p + per_wave;
    uint64_t acc = 0;
    while (p < end) {
        // `depth` loads in flight (lgkmcnt holds 15), all into the same register pair: only the fetch matters
        for (int i = 0; i < depth; ++i) {
            uint64_t v;
            asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(v) : "s"(p));
            p += stride;
            asm volatile("" : "+s"(p));
            acc ^= 0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (threadIdx.x == 1000) out[0] = (uint32_t)acc;
}
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_vector(const uint8_t* buf, size_t per_wave, uint32_t* out)
{
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const u4* p = (const u4*)(buf + (size_t)wave * per_wave) + lane;
    const size_t n = per_wave / 1024;
    u4 acc = {0, 0, 0, 0};
#pragma unroll 8
    for (size_t i = 0; i < n; ++i) acc ^= __builtin_nontemporal_load(p + i * 64);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[1] = 1;
}
int main()
{
    const size_t bytes = (size_t)3 << 30;
    uint8_t* buf; uint32_t* out;
    hipMalloc(&buf, bytes + (1 << 20)); hipMemset(buf, 1, bytes + (1 << 20)); hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int waves_per_cu : {4, 8, 16, 32}) {
        const int blocks = 256 * waves_per_cu / 4;
        const size_t per_wave = (bytes / ((size_t)blocks * 4)) & ~(size_t)4095;
        for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); k_vector<<<blocks, 256>>>(buf, per_wave, out); hipEventRecord(e1); hipEventSynchronize(e1); }
        hipEventElapsedTime(&ms, e0, e1);
        printf("vector dwordx4 stream     %2d waves/CU: %.3f ms  %.0f GB/s\n", waves_per_cu, ms, per_wave * blocks * 4.0 / ms / 1e6);
        for (int stride : {64, 128}) for (int depth : {4, 8, 15}) {
            for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); k_scalar<<<blocks, 256>>>(buf, per_wave, stride, depth, out); hipEventRecord(e1); hipEventSynchronize(e1); }
            hipEventElapsedTime(&ms, e0, e1);
            printf("scalar x2 stride %3d depth %2d, %2d waves/CU: %.3f ms  %.0f GB/s of lines covered\n", stride, depth, waves_per_cu, ms, per_wave * blocks * 4.0 / ms / 1e6);
        }
    }
    // both at once on two streams: does the scalar stream take bandwidth FROM the vector stream or add to it?
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    const int blocks = 256 * 8 / 4; const size_t half = bytes / 2, per_wave = (half / ((size_t)blocks * 4)) & ~(size_t)4095;
    for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize(); hipEventRecord(e0, 0);
        hipStreamWaitEvent(s1, e0, 0); hipStreamWaitEvent(s2, e0, 0);
        k_vector<<<blocks, 256, 0, s1>>>(buf, per_wave, out);
        k_scalar<<<blocks, 256, 0, s2>>>(buf + half, per_wave, 128, 15, out);
        hipDeviceSynchronize(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    }
    hipEventElapsedTime(&ms, e0, e1);
    printf("vector + scalar (128, 15) concurrently, 8 + 8 waves/CU, half the buffer each: %.3f ms  %.0f GB/s together\n", ms, 2.0 * per_wave * blocks * 4.0 / ms / 1e6);
    return 0;
}
