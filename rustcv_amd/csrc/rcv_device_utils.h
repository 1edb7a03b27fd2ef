// rcv_device_utils.h -- small gfx950 device helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// v_ashr_pk_u8_i32 D, S0, S1, S2 (gfx950):  D[7:0] = sat_u8(S0 >> S2), D[15:8] = sat_u8(S1 >> S2), D[31:16] are PRESERVED
// (measured on MI355X: D preset to 0xDEADBEEF comes back 0xDEADxxxx; DESIGN_HISTORY.md 6).  The ROCm 7.2 compiler pattern-
// matches pairs of clamp(x >> s, 0, 255) into this instruction and then treats D[31:16] as zero, which silently
// corrupts the neighbouring bytes.  The BUILTIN, in contrast, is typed as a 16-bit result and is handled correctly, and
// being a real instruction to the compiler it also gets the MFMA -> VALU wait states that an inline-asm copy would miss
// (accumulators in VGPR form feed it directly).  So: (a) packed saturation goes through the builtin, one VALU op per two
// values; (b) no raw `clamp(x >> s)` pair is ever left for the matcher (rcv_ashr_sat1 hides the shift behind an empty
// asm); (c) `make check-isa` rebuilds every file with -DRCV_NO_PK_BUILTIN, where the helper avoids the instruction, and
// fails if the compiler still emitted one on its own.
__device__ __forceinline__ uint32_t rcv_ashr_sat_pk2(int a, int b, int sh)
{
#ifdef RCV_NO_PK_BUILTIN
    int x = a >> sh, y = b >> sh;
    asm("" : "+v"(x));
    asm("" : "+v"(y));
    return (uint32_t)min(max(x, 0), 255) | ((uint32_t)min(max(y, 0), 255) << 8);
#else
    return (uint32_t)(unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(a, b, sh);
#endif
}

// four i32 -> (x >> sh) saturated to u8, packed little-endian into one dword: 4 VALU ops
__device__ __forceinline__ uint32_t rcv_ashr_sat_pk4(int a, int b, int c, int d, int sh)
{
    return rcv_ashr_sat_pk2(a, b, sh) | (rcv_ashr_sat_pk2(c, d, sh) << 16);
}

// (x >> sh) saturated to [0,255]; opaque to the compiler's (broken) v_ashr_pk_u8_i32 matcher
__device__ __forceinline__ int rcv_ashr_sat1(int x, int sh)
{
    int t = x >> sh;
    asm("" : "+v"(t));
    return min(max(t, 0), 255);
}

__device__ __forceinline__ int rcv_reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}
