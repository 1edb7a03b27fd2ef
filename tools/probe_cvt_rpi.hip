// Does v_cvt_rpi_i32_f32 (documented as floor(x + 0.5)) equal floor(fl32(x + 0.5)) -- the reference's rounding -- for every
// float in [0, 256)?  Exhaustive over the bit patterns 0 .. 0x43800000.   hipcc --offload-arch=gfx950 -O2 -o probe_cvt_rpi probe_cvt_rpi.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void k(unsigned long long* nbad, uint32_t* first)
{
    const uint32_t lim = 0x43800000u;
    for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < lim; b += (uint64_t)gridDim.x * blockDim.x) {
        const float v = __uint_as_float((uint32_t)b);
        int r;
        asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
        const float w = v + 0.5f;
        const int ref = (int)floorf(w);
        if (r != ref) {
            const unsigned long long i = atomicAdd(nbad, 1ull);
            if (i < 16) first[i] = (uint32_t)b;
        }
    }
}
int main()
{
    unsigned long long* nb; uint32_t* fi;
    hipMalloc(&nb, 8); hipMalloc(&fi, 64); hipMemset(nb, 0, 8); hipMemset(fi, 0, 64);
    k<<<4096, 256>>>(nb, fi);
    unsigned long long h = 0; uint32_t f[16];
    hipMemcpy(&h, nb, 8, hipMemcpyDeviceToHost); hipMemcpy(f, fi, 64, hipMemcpyDeviceToHost);
    printf("mismatches: %llu\n", h);
    for (int i = 0; i < 16 && i < (int)h; ++i) { float v; memcpy(&v, &f[i], 4); printf("  0x%08x %.10g\n", f[i], v); }
    return 0;
}
