// What does a matrix instruction cost a VALU-bound wave? (round 4, VERDICT r3 item 1: "take the 7x7 blur off the vector ALU")
//
// k_fast_blur_stream is bound by vector-ALU ISSUE (profiles/r3_fast_blur_arithmetic.md). Moving the blur's column pass onto
// the matrix cores only pays if an MFMA takes fewer of the SIMD's issue cycles than the VALU instructions it replaces.
// This probe runs the walk's instruction mix (a block of 64 VALU instructions: 40 of the 1.75 ns class -- v_perm, v_pk_*_i16,
// v_dot4 -- and 24 of the 1.05 ns class) at 3 waves per SIMD, with M matrix instructions of one shape spread through the
// block, and prints ns per block per SIMD. (time(M) - time(0)) / M = what ONE MFMA of that shape costs the wave's VALU
// stream, in ns and in "slow-class VALU instructions".
//
// Shapes: the small multi-block forms whose lane layout matches the streaming wave (lane = pixel column: 4x4x4_16B bf16 for
// a K = 4 rows chunk, 4x4x1_16B f32 for one row) and the big tiles of band_mfma.hip (32x32x32 i8, 32x32x16 bf16, 16x16x32
// bf16) for reference.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_issue_cost.hip -o mfma_issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

// 16 VALU instructions: 10 slow-class, 6 fast-class, four independent chains
#define FILL16(a0, a1, a2, a3, b, c)                                                                                     \
    asm volatile("v_perm_b32 %0, %0, %4, %5\n v_pk_min_i16 %1, %1, %4\n v_dot4_u32_u8 %2, %4, %5, %2\n v_pk_max_i16 %3, %3, %4\n" \
                 "v_fmac_f32 %0, %4, %5\n v_add_u32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n"                                   \
                 "v_alignbyte_b32 %3, %3, %4, 1\n v_pk_sub_i16 %0, %0, %4\n v_perm_b32 %1, %1, %4, %5\n"                  \
                 "v_fmac_f32 %2, %4, %5\n v_mul_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n"                                   \
                 "v_cvt_pk_u8_f32 %1, %4, 1, %1\n v_pk_add_u16 %2, %2, %4\n v_cvt_f32_u32 %3, %3"                          \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c))

template <int SHAPE>
__device__ __forceinline__ void one_mfma(v4f& c4a, v4f& c4b, v16f& c16f, v16i& c16i, const v4s& h4, const v4i& i4, const v8bf& b8, float f1) {
    if (SHAPE == 1) { c4a = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(h4, h4, c4a, 0, 0, 0); }
    if (SHAPE == 2) { c4b = __builtin_amdgcn_mfma_f32_4x4x1f32(f1, f1, c4b, 0, 0, 0); }
    if (SHAPE == 3) { c16i = __builtin_amdgcn_mfma_i32_32x32x32_i8(i4, i4, c16i, 0, 0, 0); }
    if (SHAPE == 4) { c16f = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8, b8, c16f, 0, 0, 0); }
    if (SHAPE == 5) { c4a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b8, b8, c4a, 0, 0, 0); }
}

// M matrix instructions per block of 64 VALU instructions (M = 0, 1, 2, 4, 8)
template <int SHAPE, int M>
__global__ __launch_bounds__(64) void k(unsigned* out, int iters, unsigned seed) {
    extern __shared__ unsigned pad[];      // LDS padding sets the occupancy: 3 waves per SIMD like the streaming kernel
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, b = seed | 1, c = seed * 11 + 5;
    v4f c4a = {0, 0, 0, 0}, c4b = {0, 0, 0, 0}, c4c = {0, 0, 0, 0}, c4d = {0, 0, 0, 0};
    v16f c16f = {}; v16i c16i = {};
    v4s h4 = {(short)seed, 1, 2, 3}; v4i i4 = {(int)seed, 1, 2, 3};
    v8bf b8; for (int j = 0; j < 8; j++) b8[j] = (__bf16)(float)(j + (int)seed);
    const float f1 = (float)seed;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            FILL16(a0, a1, a2, a3, b, c);
            // two independent accumulators alternate for the small shapes (the blur would have 8: 4 pixels x 2 row quads)
            if (M >= 4 || (M == 2 && (q & 1) == 0) || (M == 1 && q == 0)) {
                if (q & 1) one_mfma<SHAPE>(c4c, c4d, c16f, c16i, h4, i4, b8, f1); else one_mfma<SHAPE>(c4a, c4b, c16f, c16i, h4, i4, b8, f1);
            }
            if (M >= 8) { if (q & 1) one_mfma<SHAPE>(c4a, c4b, c16f, c16i, h4, i4, b8, f1); else one_mfma<SHAPE>(c4c, c4d, c16f, c16i, h4, i4, b8, f1); }
        }
    }
    float s = c4a[0] + c4b[1] + c4c[2] + c4d[3];
    for (int j = 0; j < 16; j++) s += c16f[j] + (float)c16i[j];
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (unsigned)s ^ pad[threadIdx.x & 1];
}

static double t0_ns = 0;
static int g_waves_per_simd = 3;
template <int SHAPE, int M> void run(const char* name, unsigned* d) {
    const int iters = 2000, blocks = 1024 * 12;
    const size_t lds = 160 * 1024 / (4 * g_waves_per_simd);     // single-wave workgroups: LDS padding sets the waves per SIMD
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE, M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, M>), dim3(blocks), dim3(64), lds, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL((k<SHAPE, M>), dim3(blocks), dim3(64), lds, 0, d, iters, 1u); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double blocks_per_simd = (double)blocks / 1024.0 * iters;          // 64-instruction blocks executed per SIMD
    const double ns = ms * 1e6 / blocks_per_simd;
    if (M == 0) t0_ns = ns;
    printf("%-22s M=%d  %8.2f ns per 64-VALU block per SIMD", name, M, ns);
    if (M > 0) printf("   +%.2f ns per MFMA = %.1f slow-class VALU instructions (1.75 ns)", (ns - t0_ns) / M, (ns - t0_ns) / M / 1.75);
    printf("\n");
}

int main() {
    unsigned* d; hipMalloc(&d, 1024 * 12 * 64 * 4);
    // the VALU block alone at 1 .. 8 waves per SIMD: how much of the issue rate the walk's mix reaches at each occupancy
    for (int w : {1, 2, 3, 4, 5, 6, 8}) { g_waves_per_simd = w; char nm[64]; snprintf(nm, sizeof nm, "no MFMA, %d waves/SIMD", w); run<0, 0>(nm, d); }
    g_waves_per_simd = 3;
    run<0, 0>("no MFMA", d);
#define SHAPE_RUNS(S, NAME) run<S, 1>(NAME, d); run<S, 2>(NAME, d); run<S, 4>(NAME, d); run<S, 8>(NAME, d);
    SHAPE_RUNS(1, "4x4x4_16B bf16")
    SHAPE_RUNS(2, "4x4x1_16B f32")
    SHAPE_RUNS(3, "32x32x32 i8")
    SHAPE_RUNS(4, "32x32x16 bf16")
    SHAPE_RUNS(5, "16x16x32 bf16")
    return 0;
}
