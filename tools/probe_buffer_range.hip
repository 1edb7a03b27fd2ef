// tools/probe_buffer_range.hip -- (round 5) how a raw buffer_load_dwordx4 (to VGPRs and direct to LDS) that STRADDLES num_records is range-checked on
// gfx950: per dword (the dwords below num_records arrive, the rest are 0) or per access (everything 0)?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(const uint32_t* src, uint32_t* out, unsigned nrec)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) ((uint32_t*)lds)[i] = 0xEEEEEEEEu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec, 0x00020000);
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, 16u * lane, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, 16u * lane, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int j = 0; j < 4; ++j) { out[8 * lane + j] = v[j]; out[8 * lane + 4 + j] = ((uint32_t*)lds)[4 * lane + j]; }
}
int main()
{
    uint32_t h[256]; for (int i = 0; i < 256; ++i) h[i] = 0x11110000u + i;
    uint32_t *d, *o; hipMalloc(&d, 1024); hipMalloc(&o, 64 * 32); hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    for (unsigned nrec : {40u, 44u, 37u}) {   // lane 2 reads bytes [32, 48): straddles
        k<<<1, 64, 4096>>>(d, o, nrec); hipDeviceSynchronize();
        uint32_t r[512]; hipMemcpy(r, o, 2048, hipMemcpyDeviceToHost);
        printf("num_records %u:\n", nrec);
        for (int l = 1; l < 4; ++l) { printf("  lane %d  vgpr:", l); for (int j = 0; j < 4; ++j) printf(" %08x", r[8 * l + j]); printf("   lds:"); for (int j = 0; j < 4; ++j) printf(" %08x", r[8 * l + 4 + j]); printf("\n"); }
    }
    return 0;
}
