// tools/ubench_dpp.hip -- (round 6) what a DPP move costs on gfx950 by kind: plain v_mov_b32, row_shr:1, wave_shr:1, wave_shl:1 (the Harris / Sobel windows use the
// wave-wide shifts: nine per row of the fused Harris kernel), and the packed-f32 ops beside them.  One wave per SIMD (1 024 waves), 8 independent chains per
// lane, 4 000 x 8 instructions per wave, 1 .. 4 waves per SIMD (1 024 .. 4 096 one-wave workgroups); prints cycles per instruction from s_memtime.
//   hipcc --offload-arch=gfx950 -O2 -w -o /tmp/ubench_dpp tools/ubench_dpp.hip && /tmp/ubench_dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int KIND>
__global__ __launch_bounds__(64) void k(uint32_t* out, unsigned long long* cyc, int iters)
{
    uint32_t r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 7 + i;
    uint32_t sr[4] = {1, 2, 3, 4};
    float f[16];
    for (int i = 0; i < 16; ++i) f[i] = (float)(threadIdx.x + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define MOVP(i) asm volatile("v_mov_b32 %0, %0" : "+v"(r[i]));
#define ROWS(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]));
#define WSHR(i) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]));
#define WSHL(i) asm volatile("v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]));
#define PKAD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&f[2 * i]) : "v"(*(double*)&f[(2 * i + 2) & 15]));
#define PKMV(i) asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]" : "+v"(*(double*)&f[2 * i]) : "v"(*(double*)&f[(2 * i + 2) & 15]));
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[i + 8]), "v"(f[(i + 1) & 7]));
#define DOT4(i) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
#define CVTB(i) asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(f[i]) : "v"(r[i]));
#define PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 7]), "s"(0x0c0c0b09u));
#define SADD(i) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sr[i & 3]));
#define SNOP(i) asm volatile("s_nop 0");
#define MIX1(i) asm volatile("v_pk_add_f32 %0, %0, %2\n\ts_add_u32 %1, %1, 3" : "+v"(*(double*)&f[2 * i]), "+s"(sr[i & 3]) : "v"(*(double*)&f[(2 * i + 2) & 15]));
#define MIX2(i) asm volatile("v_pk_add_f32 %0, %0, %1\n\ts_nop 0" : "+v"(*(double*)&f[2 * i]) : "v"(*(double*)&f[(2 * i + 2) & 15]));
#define MIX3(i) asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_mov_b32 %1, %1" : "+v"(*(double*)&f[2 * i]), "+v"(r[i]) : "v"(*(double*)&f[(2 * i + 2) & 15]));
        if (KIND == 10) { REP8(SADD) }
        if (KIND == 11) { REP8(SNOP) }
        if (KIND == 12) { REP8(MIX1) }
        if (KIND == 13) { REP8(MIX2) }
        if (KIND == 14) { REP8(MIX3) }
        if (KIND == 0) { REP8(MOVP) }
        if (KIND == 1) { REP8(ROWS) }
        if (KIND == 2) { REP8(WSHR) }
        if (KIND == 3) { REP8(WSHL) }
        if (KIND == 4) { REP8(PKAD) }
        if (KIND == 5) { REP8(PKMV) }
        if (KIND == 6) { REP8(MAX3) }
        if (KIND == 7) { REP8(DOT4) }
        if (KIND == 8) { REP8(CVTB) }
        if (KIND == 9) { REP8(PERM) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (int i = 0; i < 8; ++i) acc += r[i];
    for (int i = 0; i < 16; ++i) acc += (uint32_t)f[i];
    for (int i = 0; i < 4; ++i) acc += sr[i];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* name, uint32_t* out, unsigned long long* cyc, int nblocks)
{
    const int iters = 4000;
    hipLaunchKernelGGL(k<KIND>, dim3(nblocks), dim3(64), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<KIND>, dim3(nblocks), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblocks);
    hipMemcpy(h.data(), cyc, nblocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("  %-30s %2d waves per SIMD: a wave issues one every %6.2f cycles = %5.2f cycles of the SIMD per instruction (pairs: per pair)\n", name, nblocks / 1024, (double)h[nblocks / 2] / (iters * 8.0),
           (double)h[nblocks / 2] / (iters * 8.0) / (nblocks / 1024));
}

int main()
{
    uint32_t* out;
    unsigned long long* cyc;
    hipMalloc(&out, 8192 * 64 * 4);
    hipMalloc(&cyc, 8192 * 8);
    for (int nblocks : {1024, 3072, 4096, 6144, 8192}) {
    run<0>("v_mov_b32", out, cyc, nblocks);
    run<1>("v_mov_b32_dpp row_shr:1", out, cyc, nblocks);
    run<2>("v_mov_b32_dpp wave_shr:1", out, cyc, nblocks);
    run<3>("v_mov_b32_dpp wave_shl:1", out, cyc, nblocks);
    run<4>("v_pk_add_f32", out, cyc, nblocks);
    run<5>("v_pk_mov_b32 op_sel:[1,0]", out, cyc, nblocks);
    run<6>("v_max3_f32", out, cyc, nblocks);
    run<7>("v_dot4_u32_u8", out, cyc, nblocks);
    run<8>("v_cvt_f32_ubyte2", out, cyc, nblocks);
    run<9>("v_perm_b32", out, cyc, nblocks);
    run<11>("s_nop 0", out, cyc, nblocks);
    run<13>("pairs: v_pk_add_f32 + s_nop 0", out, cyc, nblocks);
    run<14>("pairs: v_pk_add_f32 + v_mov_b32", out, cyc, nblocks);
    }
    return 0;
}
