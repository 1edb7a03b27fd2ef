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

// Twelve i32 (three MFMA accumulator quads read crosswise) -> three dwords of saturated bytes with SIX VALU ops: op_sel[3] makes
// v_ashr_pk_u8_i32 write D[31:16] and keep D[15:0] (tools/probe_ashr_pk_opsel.hip, measured on MI355X), so a dword is two
// instructions instead of 2 + shift + or.  The compiler has no pattern for that form, hence inline asm -- and the hazard recognizer
// does not look into inline asm: a VALU read of an XDL result needs EIGHT wait states after the LAST matrix instruction that wrote an
// input (what the compiler itself puts between v_mfma_i32_16x16x64_i8 and a dependent VALU instruction on gfx950: `s_nop 7`; LLVM
// GCNHazardRecognizer::checkMAIVALUHazards).  The block therefore opens with its own `s_nop 7` (costs this wave 8 issue cycles, the SIMD's
// other waves run on), which makes it correct wherever the scheduler puts it -- directly behind the last MFMA included (235 of the 620 blocks of
// rcv_filter_rows_mfma.hip are; tests/test_isa_waits.py checks that every block opens with `s_nop 7`).  (The first builds of round 6 carried
// `s_nop 6`, one wait state less than the compiler's own figure; every parity test and soak passed with it, but the guarantee must be the
// compiler's, not the luck of a placement.)
// Outputs are early-clobber: they are written while later inputs are still to be read.
// Order of the inputs: the bytes of the three output dwords, low to high.
__device__ __forceinline__ void rcv_ashr_sat_pk12_mfma(const int (&v)[12], int sh, uint32_t& o0, uint32_t& o1, uint32_t& o2)
{
#ifdef RCV_NO_PK_BUILTIN
    o0 = rcv_ashr_sat_pk4(v[0], v[1], v[2], v[3], sh);
    o1 = rcv_ashr_sat_pk4(v[4], v[5], v[6], v[7], sh);
    o2 = rcv_ashr_sat_pk4(v[8], v[9], v[10], v[11], sh);
#else
    asm("s_nop 7\n\t"
        "v_ashr_pk_u8_i32 %0, %3, %4, %15\n\t"
        "v_ashr_pk_u8_i32 %1, %7, %8, %15\n\t"
        "v_ashr_pk_u8_i32 %2, %11, %12, %15\n\t"
        "v_ashr_pk_u8_i32 %0, %5, %6, %15 op_sel:[0,0,0,1]\n\t"
        "v_ashr_pk_u8_i32 %1, %9, %10, %15 op_sel:[0,0,0,1]\n\t"
        "v_ashr_pk_u8_i32 %2, %13, %14, %15 op_sel:[0,0,0,1]"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "s"(sh));
#endif
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

// The f32 row of a 496-pixel strip leaves its wave as WHOLE LINES (round 6, profiles/r06_harris_resp_stores.txt).  The register-window kernels give lane
// l (1 .. 62) the 8 pixels 8 (l - 1) .. + 7 of the strip: stored from there, each of the lane's two 16-byte stores writes every other 16 bytes of the wave's 2 KB
// -- two half-written visits to every 128-byte line.  Through 2 KB of wave-private LDS (`wl`, 16-byte aligned; no barrier: a wave's LDS instructions
// execute in order) lane i instead stores float4 number i and number 64 + i of the strip's row, non-temporal.  `row` = the strip's first pixel in the
// destination row (16-byte aligned), n4 = float4s of the row that lie inside the image (wave-uniform).
__device__ __forceinline__ void rcv_store_strip_row_f32(float* wl, const float (&r)[8], int lane, uint8_t* row, int n4)
{
    typedef float f4s __attribute__((ext_vector_type(4)));
    if (lane >= 1 && lane <= 62) {
        *(f4s*)(wl + 8 * (lane - 1)) = f4s{r[0], r[1], r[2], r[3]};
        *(f4s*)(wl + 8 * (lane - 1) + 4) = f4s{r[4], r[5], r[6], r[7]};
    }
    const f4s o0 = *(const f4s*)(wl + 4 * lane), o1 = *(const f4s*)(wl + 256 + 4 * lane);
    if (lane < n4) __builtin_nontemporal_store(o0, (f4s*)row + lane);
    if (64 + lane < n4) __builtin_nontemporal_store(o1, (f4s*)row + 64 + lane);
}
