// Device-side helpers shared by the extractor kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "orb_kernels.h"

namespace aria {

__device__ __forceinline__ const uint8_t* raw_level_ptr(const Plan& P, const FrameSrc& S, const uint8_t* raw,
                                                        int frame, int l, int& pitch) {
    if (l == 0) {
        pitch = S.row_stride;
        return S.img + (int64_t)frame * S.frame_stride;
    }
    pitch = P.lv[l].pitch;
    return raw + (int64_t)frame * P.raw_frame_bytes + P.lv[l].raw_off;
}

// Q4 order of a level (the blurred levels of the batch path, written by k_fast_blur_stream and read by k_describe): the
// level is cut into quads of four rows, and inside a quad the four rows of one DWORD COLUMN (4 px) are adjacent --
//   byte (x, y) lives at (y >> 2) * 4 * pitch + (x >> 2) * 16 + (y & 3) * 4 + (x & 3).
// A 128-byte line is then a 32 x 4 px tile instead of 128 x 1 px: the 37 x 37 window of a keypoint touches ~21 lines, not
// ~51, and that line traffic (L2 -> L1) is what bounds k_describe. The writer's cost is nil: a lane of the streaming kernel
// still stores one dword per row, 16 bytes from its neighbour's, and the four rows of a quad fill the same lines.
__device__ __host__ __forceinline__ int64_t q4_offset(int x, int y, int pitch) {
    return (int64_t)(y >> 2) * 4 * pitch + (int64_t)(x >> 2) * 16 + (y & 3) * 4 + (x & 3);
}

__device__ __forceinline__ int reflect101(int i, int n) {
    // BORDER_REFLECT_101; inputs here never lie more than one period outside, clamp guards tiny levels
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return min(max(i, 0), n - 1);
}

}  // namespace aria
