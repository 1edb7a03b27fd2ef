// tools/probe_smfmac.hip -- operand layout probe for v_smfmac_i32_16x16x128_i8 on gfx950 (no public layout table at hand).
//   hipcc -O2 --offload-arch=gfx950 -o gpurun_out/probe_smfmac tools/probe_smfmac.hip && gpurun_out/probe_smfmac > gpurun_out/smfmac_layout.txt
// Phase 1: for every stored A element (lane La, byte ja) and 2-bit index code c (all 16 index fields of every lane set to c),
//   A one-hot, B element (lane Lb, byte jb) carries its own location number 1 + 32*Lb + jb bit by bit over 12 runs:
//   the D entries that light up give the output row m and, per column n, WHICH B element the A element multiplies.
// Phase 3: which 2-bit field of the index register belongs to which stored element (lanes 0 and 17).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));

// cfg: La (6 bits) | ja (4) | code (2) | mode (2) | field (4)
__global__ void probe(uint16_t* out)
{
    const int cfg = blockIdx.x, lane = threadIdx.x;
    const int La = cfg & 63, ja = (cfg >> 6) & 15, code = (cfg >> 10) & 3, mode = (cfg >> 12) & 3, field = (cfg >> 14) & 15;
    uint32_t areg[4] = {0, 0, 0, 0};
    if (lane == La) areg[ja >> 2] = 1u << (8 * (ja & 3));
    uint32_t idx;
    if (mode == 0) idx = 0x55555555u * (uint32_t)code;                 // every field = code
    else {                                                             // field `field` = code, every other field = 3 - code
        idx = 0x55555555u * (uint32_t)(3 - code);
        idx = (idx & ~(3u << (2 * field))) | ((uint32_t)code << (2 * field));
    }
    uint32_t acc[4] = {0, 0, 0, 0};
    for (int r = 0; r < 12; ++r) {
        uint32_t breg[8];
        for (int w = 0; w < 8; ++w) {
            uint32_t v = 0;
            for (int b = 0; b < 4; ++b) {
                const int loc = 1 + 32 * lane + 4 * w + b;
                v |= (uint32_t)((loc >> r) & 1) << (8 * b);
            }
            breg[w] = v;
        }
        v4i a = {(int)areg[0], (int)areg[1], (int)areg[2], (int)areg[3]};
        v8i b = {(int)breg[0], (int)breg[1], (int)breg[2], (int)breg[3], (int)breg[4], (int)breg[5], (int)breg[6], (int)breg[7]};
        v4i c = {0, 0, 0, 0};
        v4i d = __builtin_amdgcn_smfmac_i32_16x16x128_i8(a, b, c, (int)idx, 0, 0);
        for (int i = 0; i < 4; ++i) acc[i] |= (uint32_t)(d[i] & 1) << r;
    }
    for (int i = 0; i < 4; ++i) out[((size_t)cfg * 64 + lane) * 4 + i] = (uint16_t)acc[i];
}

int main()
{
    const int ncfg = 1 << 18;
    uint16_t* d;
    if (hipMalloc(&d, (size_t)ncfg * 256 * 2) != hipSuccess) return 1;
    hipMemset(d, 0, (size_t)ncfg * 256 * 2);
    hipLaunchKernelGGL(probe, dim3(ncfg), dim3(64), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<uint16_t> h((size_t)ncfg * 256);
    hipMemcpy(h.data(), d, h.size() * 2, hipMemcpyDeviceToHost);
    // phase 1
    printf("# phase1: La ja code -> m ; per n: Lb.jb  (D layout assumed: lane l, reg i -> m = 4*(l>>4)+i, n = l&15)\n");
    for (int code = 0; code < 4; ++code)
        for (int La = 0; La < 64; ++La)
            for (int ja = 0; ja < 16; ++ja) {
                const int cfg = La | (ja << 6) | (code << 10);
                const uint16_t* o = &h[(size_t)cfg * 256];
                int mset = 0;
                int loc[16][16];
                for (int l = 0; l < 64; ++l)
                    for (int i = 0; i < 4; ++i) {
                        const int m = 4 * (l >> 4) + i, n = l & 15;
                        loc[m][n] = o[l * 4 + i];
                        if (o[l * 4 + i]) mset |= 1 << m;
                    }
                printf("P1 %d %d %d rows=%04x :", La, ja, code, mset);
                for (int m = 0; m < 16; ++m)
                    if (mset >> m & 1) {
                        printf(" m=%d", m);
                        for (int n = 0; n < 16; ++n) {
                            const int v = loc[m][n] - 1;
                            printf(" %d.%d", v >> 5, v & 31);
                        }
                    }
                printf("\n");
            }
    // phase 3: field mapping
    printf("# phase3: La ja field -> does the element follow `code`(=1) [Y] or the others (=2) [n]\n");
    for (int La : {0, 17})
        for (int ja = 0; ja < 16; ++ja) {
            printf("P3 %d %d :", La, ja);
            for (int field = 0; field < 16; ++field) {
                const int cfg = La | (ja << 6) | (1 << 10) | (1 << 12) | (field << 14);
                const int ref1 = La | (ja << 6) | (1 << 10), ref2 = La | (ja << 6) | (2 << 10);
                bool e1 = true, e2 = true;
                for (int t = 0; t < 256; ++t) {
                    if (h[(size_t)cfg * 256 + t] != h[(size_t)ref1 * 256 + t]) e1 = false;
                    if (h[(size_t)cfg * 256 + t] != h[(size_t)ref2 * 256 + t]) e2 = false;
                }
                printf(" %c", e1 ? 'Y' : (e2 ? 'n' : '?'));
            }
            printf("\n");
        }
    return 0;
}
