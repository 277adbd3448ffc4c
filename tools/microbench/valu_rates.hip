// Per-opcode VALU issue cost on gfx950: 256 CUs x 8 waves/SIMD, each wave runs N independent copies of one
// instruction in a loop; reports cycles per wave-instruction per SIMD. Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, b = seed | 1, c = seed * 11 + 5;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 1) { REP16(asm volatile("v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %4, %1\n v_bcnt_u32_b32 %2, %4, %2\n v_bcnt_u32_b32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 2) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 3) { REP16(asm volatile("v_pk_sub_i16 %0, %0, %4\n v_pk_sub_i16 %1, %1, %4\n v_pk_sub_i16 %2, %2, %4\n v_pk_sub_i16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 4) { REP16(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 5) { REP16(asm volatile("v_dot4_u32_u8 %0, %4, %5, %0\n v_dot4_u32_u8 %1, %4, %5, %1\n v_dot4_u32_u8 %2, %4, %5, %2\n v_dot4_u32_u8 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 6) { REP16(asm volatile("v_mad_u32_u24 %0, %4, %5, %0\n v_mad_u32_u24 %1, %4, %5, %1\n v_mad_u32_u24 %2, %4, %5, %2\n v_mad_u32_u24 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 7) { REP16(asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 8) { REP16(asm volatile("v_alignbyte_b32 %0, %0, %4, 1\n v_alignbyte_b32 %1, %1, %4, 1\n v_alignbyte_b32 %2, %2, %4, 1\n v_alignbyte_b32 %3, %3, %4, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 9) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 16, %4\n v_lshl_or_b32 %1, %1, 16, %4\n v_lshl_or_b32 %2, %2, 16, %4\n v_lshl_or_b32 %3, %3, 16, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 10) { REP16(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 11) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 12) { REP16(asm volatile("v_pk_min_i16 %0, %0, %4\n v_pk_min_i16 %1, %1, %4\n v_pk_min_i16 %2, %2, %4\n v_pk_min_i16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 13) { REP16(asm volatile("v_xor_b32 %0, s2, %0\n v_xor_b32 %1, s2, %1\n v_xor_b32 %2, s2, %2\n v_xor_b32 %3, s2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s2");) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
template <int OP> double run(const char* name, unsigned* d) {
    const int iters = 2000, blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)blocks * 4 / 1024.0 * iters * 64.0;   // wave-instructions issued per SIMD
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double cyc = ms * 1e-3 * p.clockRate * 1e3 / inst_per_simd;
    printf("%-18s %.2f cycles per wave-instruction per SIMD (%.3f ms)\n", name, cyc, ms);
    return cyc;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<10>("v_fma_f32", d); run<0>("v_xor_b32", d); run<13>("v_xor_b32 sgpr", d); run<1>("v_bcnt_u32_b32", d); run<2>("v_add_u32", d);
    run<3>("v_pk_sub_i16", d); run<12>("v_pk_min_i16", d); run<4>("v_perm_b32", d); run<5>("v_dot4_u32_u8", d);
    run<6>("v_mad_u32_u24", d); run<7>("v_min_u32", d); run<8>("v_alignbyte_b32", d); run<9>("v_lshl_or_b32", d);
    run<11>("v_mul_lo_u32", d);
    return 0;
}
