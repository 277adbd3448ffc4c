// Broad per-opcode VALU issue-rate probe on gfx950 (round 2): 8 waves/SIMD, 4 independent chains per wave.
// Reports ns per wave-instruction per SIMD (no clock assumption) and the ratio to v_xor_b32.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, b = seed | 1, c = seed * 11 + 5;
    unsigned long long d0 = a0, d1 = a1, d2 = a2, d3 = a3, bb = b;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 1) { REP16(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 2) { REP16(asm volatile("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 3) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 4) { REP16(asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 5) { REP16(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 6) { REP16(asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 7) { REP16(asm volatile("v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %4\n v_max_u32 %2, %2, %4\n v_max_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 8) { REP16(asm volatile("v_min_i32 %0, %0, %4\n v_min_i32 %1, %1, %4\n v_min_i32 %2, %2, %4\n v_min_i32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 9) { REP16(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 10) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc", "s2");) }
        if (OP == 11) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 12) { REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 13) { REP16(asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 14) { REP16(asm volatile("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 15) { REP16(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 16) { REP16(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 17) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(bb));) }
        if (OP == 18) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(bb));) }
        if (OP == 19) { REP16(asm volatile("v_pk_fma_f32 %0, %4, %4, %0\n v_pk_fma_f32 %1, %4, %4, %1\n v_pk_fma_f32 %2, %4, %4, %2\n v_pk_fma_f32 %3, %4, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(bb));) }
        if (OP == 20) { REP16(asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 21) { REP16(asm volatile("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 22) { REP16(asm volatile("v_bfe_u32 %0, %0, 3, 8\n v_bfe_u32 %1, %1, 3, 8\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 23) { REP16(asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte0 %1, %1\n v_cvt_f32_ubyte0 %2, %2\n v_cvt_f32_ubyte0 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 24) { REP16(asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 25) { REP16(asm volatile("v_cvt_pk_u8_f32 %0, %4, 1, %0\n v_cvt_pk_u8_f32 %1, %4, 1, %1\n v_cvt_pk_u8_f32 %2, %4, 1, %2\n v_cvt_pk_u8_f32 %3, %4, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 26) { REP16(asm volatile("v_sad_u8 %0, %0, %4, %5\n v_sad_u8 %1, %1, %4, %5\n v_sad_u8 %2, %2, %4, %5\n v_sad_u8 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 27) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 28) { REP16(asm volatile("v_mad_u32_u24 %0, %4, %5, %0\n v_mad_u32_u24 %1, %4, %5, %1\n v_mad_u32_u24 %2, %4, %5, %2\n v_mad_u32_u24 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 29) { REP16(asm volatile("v_add_u16 %0, %0, %4\n v_add_u16 %1, %1, %4\n v_add_u16 %2, %2, %4\n v_add_u16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 30) { REP16(asm volatile("v_max_u16 %0, %0, %4\n v_max_u16 %1, %1, %4\n v_max_u16 %2, %2, %4\n v_max_u16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 31) { REP16(asm volatile("v_min_i16 %0, %0, %4\n v_min_i16 %1, %1, %4\n v_min_i16 %2, %2, %4\n v_min_i16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 32) { REP16(asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 33) { REP16(asm volatile("v_pk_sub_i16 %0, %0, %4\n v_pk_sub_i16 %1, %1, %4\n v_pk_sub_i16 %2, %2, %4\n v_pk_sub_i16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 34) { REP16(asm volatile("v_pk_max_i16 %0, %0, %4\n v_pk_max_i16 %1, %1, %4\n v_pk_max_i16 %2, %2, %4\n v_pk_max_i16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 35) { REP16(asm volatile("v_pk_min_u16 %0, %0, %4\n v_pk_min_u16 %1, %1, %4\n v_pk_min_u16 %2, %2, %4\n v_pk_min_u16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 36) { REP16(asm volatile("v_pk_add_f16 %0, %0, %4\n v_pk_add_f16 %1, %1, %4\n v_pk_add_f16 %2, %2, %4\n v_pk_add_f16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 37) { REP16(asm volatile("v_pk_min_f16 %0, %0, %4\n v_pk_min_f16 %1, %1, %4\n v_pk_min_f16 %2, %2, %4\n v_pk_min_f16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 38) { REP16(asm volatile("v_pk_max_f16 %0, %0, %4\n v_pk_max_f16 %1, %1, %4\n v_pk_max_f16 %2, %2, %4\n v_pk_max_f16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 39) { REP16(asm volatile("v_pk_minimum3_f16 %0, %0, %4, %5\n v_pk_minimum3_f16 %1, %1, %4, %5\n v_pk_minimum3_f16 %2, %2, %4, %5\n v_pk_minimum3_f16 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 40) { REP16(asm volatile("v_pk_maximum3_f16 %0, %0, %4, %5\n v_pk_maximum3_f16 %1, %1, %4, %5\n v_pk_maximum3_f16 %2, %2, %4, %5\n v_pk_maximum3_f16 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 41) { REP16(asm volatile("v_pk_mad_u16 %0, %4, %5, %0\n v_pk_mad_u16 %1, %4, %5, %1\n v_pk_mad_u16 %2, %4, %5, %2\n v_pk_mad_u16 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 42) { REP16(asm volatile("v_pk_fma_f16 %0, %4, %5, %0\n v_pk_fma_f16 %1, %4, %5, %1\n v_pk_fma_f16 %2, %4, %5, %2\n v_pk_fma_f16 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 43) { REP16(asm volatile("v_med3_i32 %0, %0, %4, %5\n v_med3_i32 %1, %1, %4, %5\n v_med3_i32 %2, %2, %4, %5\n v_med3_i32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 44) { REP16(asm volatile("v_min3_u32 %0, %0, %4, %5\n v_min3_u32 %1, %1, %4, %5\n v_min3_u32 %2, %2, %4, %5\n v_min3_u32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 45) { REP16(asm volatile("v_max3_i16 %0, %0, %4, %5\n v_max3_i16 %1, %1, %4, %5\n v_max3_i16 %2, %2, %4, %5\n v_max3_i16 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 46) { REP16(asm volatile("v_min_f16 %0, %0, %4\n v_min_f16 %1, %1, %4\n v_min_f16 %2, %2, %4\n v_min_f16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 47) { REP16(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 48) { REP16(asm volatile("v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %1, %1, %4, 31\n v_alignbit_b32 %2, %2, %4, 31\n v_alignbit_b32 %3, %3, %4, 31" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 49) { REP16(asm volatile("v_dot4_u32_u8 %0, %4, %5, %0\n v_dot4_u32_u8 %1, %4, %5, %1\n v_dot4_u32_u8 %2, %4, %5, %2\n v_dot4_u32_u8 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 50) { REP16(asm volatile("v_dot2_u32_u16 %0, %4, %5, %0\n v_dot2_u32_u16 %1, %4, %5, %1\n v_dot2_u32_u16 %2, %4, %5, %2\n v_dot2_u32_u16 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 51) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 16, %4\n v_lshl_or_b32 %1, %1, 16, %4\n v_lshl_or_b32 %2, %2, 16, %4\n v_lshl_or_b32 %3, %3, 16, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 52) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_lshl_add_u32 %2, %2, 2, %4\n v_lshl_add_u32 %3, %3, 2, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 53) { REP16(asm volatile("v_xor_b32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_xor_b32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_xor_b32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_xor_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 54) { REP16(asm volatile("v_sub_u16_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_sub_u16_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_sub_u16_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_sub_u16_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 55) { REP16(asm volatile("v_add_u32_dpp %0, %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 56) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc", "s2");) }
        if (OP == 57) { REP16(asm volatile("v_bitop3_b32 %0, %0, %4, %5 bitop3:0x96\n v_bitop3_b32 %1, %1, %4, %5 bitop3:0x96\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0x96" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 58) { REP16(asm volatile("v_xor_b32 %0, 0x80808080, %0\n v_xor_b32 %1, 0x80808080, %1\n v_xor_b32 %2, 0x80808080, %2\n v_xor_b32 %3, 0x80808080, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 59) { REP16(asm volatile("v_xor_b32 %0, s2, %0\n v_xor_b32 %1, s2, %1\n v_xor_b32 %2, s2, %2\n v_xor_b32 %3, s2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc", "s2");) }
        if (OP == 60) { REP16(asm volatile("v_xor_b32 %0, 17, %0\n v_xor_b32 %1, 17, %1\n v_xor_b32 %2, 17, %2\n v_xor_b32 %3, 17, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (unsigned)(d0 ^ d1 ^ d2 ^ d3);
}
static double base = 0;
template <int OP> void run(const char* name, unsigned* d) {
    const int iters = 1000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)blocks * 4 / 1024.0 * iters * 64.0;
    const double ns = ms * 1e6 / inst_per_simd;
    if (base == 0) base = ns;
    printf("%-20s %.3f ns per wave-instruction per SIMD   x%.2f\n", name, ns, ns / base);
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_xor_b32", d);
    run<1>("v_and_b32", d);
    run<2>("v_or_b32", d);
    run<3>("v_add_u32", d);
    run<4>("v_sub_u32", d);
    run<5>("v_lshlrev_b32", d);
    run<6>("v_lshrrev_b32", d);
    run<7>("v_max_u32", d);
    run<8>("v_min_i32", d);
    run<9>("v_mov_b32", d);
    run<10>("v_cndmask_b32", d);
    run<11>("v_add_f32", d);
    run<12>("v_mul_f32", d);
    run<13>("v_max_f32", d);
    run<14>("v_min_f32", d);
    run<15>("v_fmac_f32", d);
    run<16>("v_fma_f32", d);
    run<17>("v_pk_add_f32", d);
    run<18>("v_pk_mul_f32", d);
    run<19>("v_pk_fma_f32", d);
    run<20>("v_add3_u32", d);
    run<21>("v_and_or_b32", d);
    run<22>("v_bfe_u32", d);
    run<23>("v_cvt_f32_ubyte0", d);
    run<24>("v_cvt_f32_u32", d);
    run<25>("v_cvt_pk_u8_f32", d);
    run<26>("v_sad_u8", d);
    run<27>("v_mul_u32_u24", d);
    run<28>("v_mad_u32_u24", d);
    run<29>("v_add_u16", d);
    run<30>("v_max_u16", d);
    run<31>("v_min_i16", d);
    run<32>("v_pk_add_u16", d);
    run<33>("v_pk_sub_i16", d);
    run<34>("v_pk_max_i16", d);
    run<35>("v_pk_min_u16", d);
    run<36>("v_pk_add_f16", d);
    run<37>("v_pk_min_f16", d);
    run<38>("v_pk_max_f16", d);
    run<39>("v_pk_minimum3_f16", d);
    run<40>("v_pk_maximum3_f16", d);
    run<41>("v_pk_mad_u16", d);
    run<42>("v_pk_fma_f16", d);
    run<43>("v_med3_i32", d);
    run<44>("v_min3_u32", d);
    run<45>("v_max3_i16", d);
    run<46>("v_min_f16", d);
    run<47>("v_perm_b32", d);
    run<48>("v_alignbit_b32", d);
    run<49>("v_dot4_u32_u8", d);
    run<50>("v_dot2_u32_u16", d);
    run<51>("v_lshl_or_b32", d);
    run<52>("v_lshl_add_u32", d);
    run<53>("v_xor_sdwa_b0", d);
    run<54>("v_sub_u16_sdwa", d);
    run<55>("v_add_u32_dpp", d);
    run<56>("v_cmp_lt_u32", d);
    run<57>("v_bitop3_b32", d);
    run<58>("v_xor_b32 lit", d);
    run<59>("v_xor_b32 sgpr", d);
    run<60>("v_xor_b32 inl", d);
    return 0;
}
