// FETCH_SIZE calibration for 4-byte-per-lane loads (MI355X_MICROARCH.md: "other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern"). k_fast_blur_stream reads one dword per lane per row; this copies n dwords
// the same way (one dword per lane and instruction, 256 contiguous bytes per wave). Built on the GPU box by tools/pmc_traffic.sh
// as libcopy_dword.so and called once by tools/prof_extract.py --calibrate.
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ void k_calib_copy_dword(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}
extern "C" int calib_copy_dword(const void* src, void* dst, size_t bytes, void* stream) {
    hipLaunchKernelGGL(k_calib_copy_dword, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)src, (uint32_t*)dst, bytes / 4);
    return (int)hipGetLastError();
}
