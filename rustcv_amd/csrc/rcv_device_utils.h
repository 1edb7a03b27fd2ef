// rcv_device_utils.h -- small gfx950 device helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// v_ashr_pk_u8_i32 D, S0, S1, S2 (gfx950):  D[7:0] = sat_u8(S0 >> S2), D[15:8] = sat_u8(S1 >> S2),
// D[31:16] are PRESERVED (measured on MI355X, scratch probe in DESIGN.md §5).  The ROCm 7.2 compiler
// pattern-matches clamp(x >> s, 0, 255) pairs into this instruction but then treats D[31:16] as
// zero, which silently corrupts the neighbouring bytes.  We therefore (a) emit it ourselves, with
// the upper half declared garbage and dropped by v_perm_b32, and (b) never leave a raw
// `clamp(x >> s)` pair for the compiler to find (csrc/Makefile `check-isa` enforces this: every
// v_ashr_pk_u8_i32 in the ISA must carry the "rcv" marker below).
__device__ __forceinline__ uint32_t rcv_ashr_sat_pk2(int a, int b, int sh)
{
    uint32_t d;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3 ; rcv" : "=v"(d) : "v"(a), "v"(b), "v"(sh));
    return d; // only bits [15:0] are meaningful
}

// four i32 -> (x >> sh) saturated to u8, packed little-endian into one dword: 3 VALU ops
__device__ __forceinline__ uint32_t rcv_ashr_sat_pk4(int a, int b, int c, int d, int sh)
{
    uint32_t lo = rcv_ashr_sat_pk2(a, b, sh), hi = rcv_ashr_sat_pk2(c, d, sh);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
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
