// Probe for a Hamming kNN on the FP4 matrix path of gfx950: v_mfma_scale_f32_32x32x64_f8f6f4 with E2M1 operands.
// (1) exactness / operand map: A = 32 rows x 64 bits, B = 64 bits x 32 columns, every bit one FP4 value (A: bit -> -2.0,
//     B: bit -> 1.0); D[i][j] must equal C[i][j] - 2 |a_i & b_j| exactly. Lane l carries row/column l & 31 and the K half
//     l >> 5 (32 bits -> 32 nibbles = 4 dwords), the same map on both sides, so the k order inside a half does not matter.
// (2) rate: ns per instruction per SIMD in a dependent-free loop, on zero operands and on random bits.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_fp4_bits.hip -o mfma_fp4_bits
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// 8 bits -> 8 nibbles (bit k -> nibble k), nibble value `val` (4 bits) where the bit is set
__device__ __host__ inline uint32_t spread8(uint32_t b, uint32_t val) {
    uint32_t x = b & 0xFFu;
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return x * val;
}

__global__ void k_check(const uint32_t* __restrict__ a_bits, const uint32_t* __restrict__ b_bits, float* __restrict__ d, int scale_a, int scale_b) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const uint32_t aw = a_bits[r * 2 + h], bw = b_bits[r * 2 + h];      // 32 bits of row r / column r, K half h
    v8i A = {0, 0, 0, 0, 0, 0, 0, 0}, B = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < 4; q++) {
        A[q] = (int)spread8(aw >> (8 * q), 0xCu);      // -2.0 = 1100b
        B[q] = (int)spread8(bw >> (8 * q), 0x2u);      // +1.0 = 0010b
    }
    v16f C;
    for (int j = 0; j < 16; j++) C[j] = 1000.0f + (float)j + (float)r / 4096.0f;
    v16f D = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, C, 4, 4, 0, scale_a, 0, scale_b);
    for (int j = 0; j < 16; j++) d[lane * 16 + j] = D[j];
}

template <int N>
__global__ void k_rate(const uint32_t* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    v8i A = {0, 0, 0, 0, 0, 0, 0, 0}, B = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < 4; q++) { A[q] = (int)src[lane * 8 + q]; B[q] = (int)src[lane * 8 + 4 + q]; }
    v16f acc[N];
    for (int n = 0; n < N; n++) for (int j = 0; j < 16; j++) acc[n][j] = 0.f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int n = 0; n < N; n++) acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[n], 4, 4, 0, 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < N; n++) for (int j = 0; j < 16; j++) s += acc[n][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    std::vector<uint32_t> a(64), b(64);
    srand(7);
    for (auto& v : a) v = (uint32_t)rand() * 2654435761u;
    for (auto& v : b) v = (uint32_t)rand() * 40503u + 77u;
    uint32_t *da, *db; float* dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 64 * 16 * 4);
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    for (int sc = 0; sc < 2; sc++) {
        const int sa = sc ? 127 : 0, sb = sc ? 127 : 0;
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, da, db, dd, sa, sb);
        std::vector<float> d(64 * 16);
        hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0; double first_got = 0, first_want = 0;
        for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 16; j++) {
                const int col = lane & 31, row = (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);      // C/D map of the 32x32 shapes
                int c = 0;
                for (int hh = 0; hh < 2; hh++) c += __builtin_popcount(a[row * 2 + hh] & b[col * 2 + hh]);
                const float want = 1000.0f + (float)j + (float)col / 4096.0f - 2.0f * (float)c;
                if (d[lane * 16 + j] != want) { if (!bad) { first_got = d[lane * 16 + j]; first_want = want; } bad++; }
            }
        printf("scale operands %d/%d: %d of 1024 results differ (first: got %.6f want %.6f)\n", sa, sb, bad, first_got, first_want);
    }
    // rate
    std::vector<uint32_t> src(64 * 8);
    float* dout; uint32_t* dsrc;
    hipMalloc(&dout, 256 * 4 * 256 * 4); hipMalloc(&dsrc, src.size() * 4);
    for (int pass = 0; pass < 2; pass++) {
        for (auto& v : src) v = pass ? (spread8(rand(), 0xC) ^ 0) : 0u;
        if (pass) for (size_t i = 0; i < src.size(); i++) src[i] = (i & 4) ? spread8(rand(), 0x2) : spread8(rand(), 0xC);
        hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice);
        const int iters = 20000;
        hipLaunchKernelGGL(k_rate<4>, dim3(256 * 2), dim3(256), 0, 0, dsrc, dout, 100);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_rate<4>, dim3(256 * 2), dim3(256), 0, 0, dsrc, dout, iters);      // 2 waves per SIMD
        hipDeviceSynchronize();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%s operands: %.2f ns per MFMA per SIMD (2 waves per SIMD, 4 independent accumulators)\n", pass ? "bit-pattern" : "zero",
               s * 1e9 / ((double)iters * 4 * 2));
    }
    return 0;
}
