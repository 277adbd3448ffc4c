// Does the operand bit pattern set the rate of v_mfma_i32_32x32x32_i8 on gfx950 (clock give-back under load)?
// 2 waves per SIMD, each runs ITER x 16 MFMAs (two accumulators, 8 A and 8 B fragments from a table) + 4 VALU fillers per MFMA.
// Patterns: 0 zeros, 1 train {0x40,0xC0} x query {0x7F,0x81} (+-64 x +-127), 2 single-bit bytes {0,0x80} x {0,0x40},
//           3 uniformly random bytes, 4 +-1 (0x01/0xFF) both
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_i8_power mfma_i8_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(const v4i* __restrict__ ta, const v4i* __restrict__ tb, int iters, int* out) {
    v4i a[8], b[8];
    for (int s = 0; s < 8; s++) { a[s] = ta[(s * 64 + (threadIdx.x & 63))]; b[s] = tb[(s * 64 + (threadIdx.x & 63))]; }
    v16i acc0 = {}, acc1 = {};
    float m[8];
    for (int i = 0; i < 8; i++) m[i] = (float)(threadIdx.x + i + 1);
    float kk = 1.5f + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int s = 0; s < 8; s++) {
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[s], acc0, 0, 0, 0);
#pragma unroll
            for (int f = 0; f < 4; f++) m[f] = __builtin_amdgcn_fmed3f(m[f], m[f + 1], kk);
            __builtin_amdgcn_sched_barrier(0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[7 - s], acc1, 0, 0, 0);
#pragma unroll
            for (int f = 4; f < 8; f++) m[f] = __builtin_amdgcn_fmed3f(m[f], m[(f + 1) & 7], kk);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0;
    for (int i = 0; i < 8; i++) r += m[i];
    int x = 0;
    for (int i = 0; i < 16; i++) x += acc0[i] + acc1[i];
    if (x == 123456789 || r == 1.25f) out[0] = x;
}

int main() {
    int* d; hipMalloc(&d, 4);
    v4i *ta, *tb; hipMalloc(&ta, 8 * 64 * 16); hipMalloc(&tb, 8 * 64 * 16);
    const int iters = 3000;
    for (int pat = 0; pat < 5; pat++) {
        std::vector<unsigned char> ha(8 * 64 * 16), hb(8 * 64 * 16);
        srand(1234);
        for (size_t i = 0; i < ha.size(); i++) {
            const int ra = rand() & 1, rb = (rand() >> 3) & 1;
            switch (pat) {
                case 0: ha[i] = 0; hb[i] = 0; break;
                case 1: ha[i] = ra ? 0xC0 : 0x40; hb[i] = rb ? 0x7F : 0x81; break;
                case 2: ha[i] = ra ? 0x80 : 0x00; hb[i] = rb ? 0x40 : 0x00; break;
                case 3: ha[i] = rand() & 255; hb[i] = (rand() >> 4) & 255; break;
                case 4: ha[i] = ra ? 0xFF : 0x01; hb[i] = rb ? 0x01 : 0xFF; break;
            }
        }
        hipMemcpy(ta, ha.data(), ha.size(), hipMemcpyHostToDevice);
        hipMemcpy(tb, hb.data(), hb.size(), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, ta, tb, 200, d);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, ta, tb, iters, d);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("pattern %d rep %d: %.3f ms -> %.2f ns per MFMA per SIMD\n", pat, rep, ms, ms * 1e6 / (iters * 16.0 * 2));
        }
    }
    return 0;
}
