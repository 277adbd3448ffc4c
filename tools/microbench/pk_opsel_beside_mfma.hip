// Probe for the hazard of DESIGN.md section 4 ("A hazard worth recording"): v_pk_add_f32 with op_sel-swizzled register pairs
// gave wrong blurred pixels in k_fast_blur_stream only while the matcher's MFMA waves shared the SIMDs. This runs a loop of
// exactly that instruction form (and, as a control, the plain form), checked against scalar adds, first alone and then
// beside a kernel that keeps the matrix pipes busy on another stream.
// Build: hipcc --offload-arch=gfx950 -O2 pk_opsel_beside_mfma.hip -o pk_opsel_beside_mfma
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <bool SWZ>
__global__ __launch_bounds__(256) void k_pk(const float* __restrict__ src, unsigned long long* __restrict__ bad, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    f2 a = {src[(tid * 4) & 4095], src[(tid * 4 + 1) & 4095]}, b = {src[(tid * 4 + 2) & 4095], src[(tid * 4 + 3) & 4095]};
    unsigned long long wrong = 0;
    for (int i = 0; i < iters; i++) {
        f2 r;
        if (SWZ) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));      // (a.y + b.x, a.x + b.y)
        else asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        const float w0 = SWZ ? a.y + b.x : a.x + b.x, w1 = SWZ ? a.x + b.y : a.y + b.y;
        wrong += (r.x != w0) + (r.y != w1);
        // next operands: small integers, exact in fp32
        a.x = (float)((int)(a.x + 3.f) & 1023); a.y = (float)((int)(a.y + 5.f) & 1023);
        b.x = (float)((int)(b.x + 7.f) & 1023); b.y = (float)((int)(b.y + 11.f) & 1023);
    }
    if (wrong) atomicAdd(bad, wrong);
}

__global__ __launch_bounds__(256) void k_mfma(int* __restrict__ out, int iters) {
    v4i A = {(int)threadIdx.x * 0x01010101, (int)0x80008000u, 0x00800080, 0x40404040}, B = {0x40004000, 0x00400040, (int)threadIdx.x, 0x40400000};
    v16i acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    for (int i = 0; i < iters; i++) {
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc3, 0, 0, 0);
    }
    int s = 0;
    for (int j = 0; j < 16; j++) s += acc0[j] + acc1[j] + acc2[j] + acc3[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = (float)((i * 37 + 11) & 1023);
    float* src; unsigned long long* bad; int* out;
    hipMalloc(&src, 4096 * 4); hipMalloc(&bad, 8); hipMalloc(&out, 1024 * 256 * 4);
    hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (int swz = 1; swz >= 0; swz--)
        for (int beside = 0; beside < 2; beside++) {
            hipMemset(bad, 0, 8);
            hipDeviceSynchronize();
            if (beside) hipLaunchKernelGGL(k_mfma, dim3(512), dim3(256), 0, s2, out, 400000);
            for (int rep = 0; rep < 8; rep++) {
                if (swz) hipLaunchKernelGGL(k_pk<true>, dim3(2048), dim3(256), 0, s1, src, bad, 20000);
                else hipLaunchKernelGGL(k_pk<false>, dim3(2048), dim3(256), 0, s1, src, bad, 20000);
            }
            hipDeviceSynchronize();
            unsigned long long n = 0;
            hipMemcpy(&n, bad, 8, hipMemcpyDeviceToHost);
            printf("v_pk_add_f32 %s, %s: %llu wrong results of %.3g\n", swz ? "op_sel:[1,0] op_sel_hi:[0,1]" : "plain", beside ? "beside the MFMA kernel" : "alone",
                   n, 8.0 * 2048 * 256 * 20000 * 2);
        }
    return 0;
}
