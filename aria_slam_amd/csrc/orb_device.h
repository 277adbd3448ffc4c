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

__device__ __forceinline__ int reflect101(int i, int n) {
    // BORDER_REFLECT_101; inputs here never lie more than one period outside, clamp guards tiny levels
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return min(max(i, 0), n - 1);
}

}  // namespace aria
