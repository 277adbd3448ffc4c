// Does gfx950 execute the GFX9 DPP wave shifts (wave_shr:1 / wave_shl:1) that the assembler accepts?
// k_fast_blur_stream takes a lane's left / right neighbour dword with them. Prints the first mismatching lane or OK.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const int* in, int* out) {
    const int v = in[threadIdx.x];
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);          // wave_shr:1: lane i <- lane i-1
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true);     // wave_shl:1: lane i <- lane i+1
}
int main() {
    int h[64], o[128], *di, *dout;
    for (int i = 0; i < 64; i++) h[i] = 1000 + i;
    hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) {
        const int want_r = i > 0 ? h[i - 1] : 0, want_l = i < 63 ? h[i + 1] : 0;
        if (o[i] != want_r || o[64 + i] != want_l) { if (!bad) printf("lane %d: shr %d (want %d), shl %d (want %d)\n", i, o[i], want_r, o[64 + i], want_l); bad++; }
    }
    printf(bad ? "MISMATCH in %d lanes\n" : "OK wave_shr / wave_shl\n", bad);
    return bad ? 1 : 0;
}
