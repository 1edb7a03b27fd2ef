// tools/probe_ashr_pk_opsel.hip -- (round 6) does v_ashr_pk_u8_i32 honour op_sel[3] on gfx950?  Without it the instruction writes the two
// saturated bytes into D[15:0] and PRESERVES D[31:16] (tools notes, DESIGN_HISTORY.md 6); with op_sel:[0,0,0,1] the 16-bit result should
// land in D[31:16] and D[15:0] stay -- then four accumulators pack into one dword with TWO instructions (today: 2 + shift + or = 4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const int* in, uint32_t* out, int sh)
{
    const int a = in[threadIdx.x * 4], b = in[threadIdx.x * 4 + 1], c = in[threadIdx.x * 4 + 2], d = in[threadIdx.x * 4 + 3];
    uint32_t r = 0xdeadbeefu;
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "+v"(r) : "v"(a), "v"(b), "s"(sh));
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(r) : "v"(c), "v"(d), "s"(sh));
    out[threadIdx.x] = r;
}
static uint32_t sat(int x, int sh) { int t = x >> sh; return (uint32_t)(t < 0 ? 0 : (t > 255 ? 255 : t)); }
int main()
{
    int h[256]; srand(7);
    for (int i = 0; i < 256; ++i) h[i] = (rand() % 80000) - 20000;
    h[0] = -1; h[1] = 255 << 6; h[2] = (256 << 6); h[3] = 0x7fffffff; h[4] = (int)0x80000000; h[5] = 0; h[6] = 63; h[7] = 64;
    int* d; uint32_t* o; hipMalloc(&d, sizeof h); hipMalloc(&o, 256); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    int bad = 0;
    for (int sh : {0, 6, 8, 12}) {
        k<<<1, 64>>>(d, o, sh); hipDeviceSynchronize();
        uint32_t r[64]; hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) {
            const uint32_t want = sat(h[4 * l], sh) | (sat(h[4 * l + 1], sh) << 8) | (sat(h[4 * l + 2], sh) << 16) | (sat(h[4 * l + 3], sh) << 24);
            if (r[l] != want) { if (bad < 6) printf("shift %d lane %d: got %08x want %08x\n", sh, l, r[l], want); ++bad; }
        }
    }
    printf(bad ? "op_sel[3] NOT honoured as assumed: %d mismatches\n" : "op_sel[3] honoured: v_ashr_pk_u8_i32 writes D[31:16], keeps D[15:0] (%d mismatches)\n", bad);
    return bad != 0;
}
