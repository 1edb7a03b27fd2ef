#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
// (1) direct global -> LDS loads of 16 B per lane; (2) unaligned ds_read_b64 / b96
__global__ void k(const uint8_t* src, uint32_t* out, int mode)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x;
    // each lane fetches 16 B from its own address (reverse order) into LDS at lane * 16
    const uint8_t* g = src + (63 - lane) * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    // unaligned read: 8 bytes at byte offset lane * 3 + 1
    const uint8_t* p = lds + lane * 3 + 1;
    uint64_t v;
    if (mode == 0) { asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)p)); }
    else { typedef uint32_t u3 __attribute__((ext_vector_type(3))); u3 w; asm volatile("ds_read_b96 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)p)); v = w.x | ((uint64_t)(w.y ^ w.z) << 32); }
    out[2 * lane] = (uint32_t)v;
    out[2 * lane + 1] = (uint32_t)(v >> 32);
}
int main()
{
    uint8_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t* d; uint32_t* o; hipMalloc(&d, 1024); hipMalloc(&o, 512); hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(o, 0, 512);
        k<<<1, 64, 4096>>>(d, o, mode);
        hipError_t e = hipDeviceSynchronize();
        uint32_t r[128]; hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
        // expected LDS image: chunk l = src chunk 63 - l
        uint8_t img[1040] = {0}; for (int l = 0; l < 64; ++l) for (int b = 0; b < 16; ++b) img[l * 16 + b] = h[(63 - l) * 16 + b];
        int bad = 0;
        for (int l = 0; l < 64; ++l) { uint64_t want = 0; for (int b = 0; b < 8; ++b) want |= (uint64_t)img[l * 3 + 1 + b] << (8 * b);
            if (mode == 1) { uint32_t w2 = 0; for (int b = 0; b < 4; ++b) w2 |= (uint32_t)img[l * 3 + 1 + 8 + b] << (8 * b); want = (uint32_t)want | ((uint64_t)((uint32_t)(want >> 32) ^ w2) << 32); }
            uint64_t got = r[2 * l] | ((uint64_t)r[2 * l + 1] << 32); if (got != want) ++bad; }
        printf("mode %d: err %d, %d of 64 lanes wrong\n", mode, (int)e, bad);
    }
    return 0;
}
