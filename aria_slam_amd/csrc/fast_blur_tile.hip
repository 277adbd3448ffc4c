// 64x32-tile form of the FAST + NMS + blur stage: the first correct kernel of round 1,
// kept as a bit-identical alternative (ARIA_FAST_BLUR_IMPL=tile, tests/test_gpu_variants.py). Not on the default path.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <cstdlib>

#include "common.h"
#include "orb_device.h"
#include "orb_kernels.h"

namespace aria {

// ------------------------------------------------------------------------------------------------------
// a6.2 + a6.7  one 64x32 tile of one level of one frame per workgroup.
// ------------------------------------------------------------------------------------------------------
// ring of radius 3 in circular order (fast_score.cpp makeOffsets); only circular adjacency matters
#define RING_DX {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1}
#define RING_DY {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3}

// FAST-9/16 test + cornerScore<16> at LDS patch position c (row pitch kPatchW). Returns 0 if not a corner,
// else the score (>= threshold). fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>.
__device__ __forceinline__ int fast9_score(const uint8_t* c, int t) {
    constexpr int dxs[16] = RING_DX;
    constexpr int dys[16] = RING_DY;
    const int v = c[0];
    const int lo = v - t, hi = v + t;
    // any 9 consecutive ring positions contain at least two of the compass points 0,4,8,12
    int p0 = c[dys[0] * kPatchW + dxs[0]], p4 = c[dys[4] * kPatchW + dxs[4]];
    int p8 = c[dys[8] * kPatchW + dxs[8]], p12 = c[dys[12] * kPatchW + dxs[12]];
    int nd = (p0 < lo) + (p4 < lo) + (p8 < lo) + (p12 < lo);
    int nb = (p0 > hi) + (p4 > hi) + (p8 > hi) + (p12 > hi);
    if (nd < 2 && nb < 2) return 0;
    int d[16];
    uint32_t dark = 0, bright = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int p = c[dys[k] * kPatchW + dxs[k]];
        d[k] = v - p;
        dark |= (uint32_t)(p < lo) << k;
        bright |= (uint32_t)(p > hi) << k;
    }
    uint32_t m = dark | (dark << 16);
    uint32_t a = m & (m >> 1);
    a &= a >> 2;
    a &= a >> 4;
    a &= m >> 8;
    uint32_t mb = bright | (bright << 16);
    uint32_t b = mb & (mb >> 1);
    b &= b >> 2;
    b &= b >> 4;
    b &= mb >> 8;
    if (((a | b) & 0xFFFFu) == 0) return 0;
    // score = max over the 16 nine-arcs of min(d) and of min(-d), minus 1
    int mn2[16], mx2[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn2[k] = min(d[k], d[(k + 1) & 15]); mx2[k] = max(d[k], d[(k + 1) & 15]); }
    int mn4[16], mx4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn4[k] = min(mn2[k], mn2[(k + 2) & 15]); mx4[k] = max(mx2[k], mx2[(k + 2) & 15]); }
    int q0 = -1000, q1 = 1000;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int mn9 = min(min(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
        int mx9 = max(max(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
        q0 = max(q0, mn9);
        q1 = min(q1, mx9);
    }
    return max(q0, -q1) - 1;
}

__global__ __launch_bounds__(256) void k_fast_blur(Plan P, FrameSrc S, const uint8_t* __restrict__ raw,
                                                   uint8_t* __restrict__ blur, uint32_t* __restrict__ cand,
                                                   int* __restrict__ cand_cnt, int* __restrict__ err) {
    __shared__ __attribute__((aligned(16))) uint8_t s_pix[kPatchH * kPatchW];
    __shared__ uint8_t s_score[(kTileH + 2) * (kTileW + 4)];
    __shared__ uint16_t s_row[(kTileH + 6) * kTileW];

    const int tid = threadIdx.x;
    const int frame = blockIdx.y;
    int l = 0;
#pragma unroll
    for (int i = 1; i < kLevels; i++)
        if ((int)blockIdx.x >= P.lv[i].tile_base) l = i;
    const LevelGeom g = P.lv[l];
    const int t_in = blockIdx.x - g.tile_base;
    const int tyi = t_in / g.tiles_x, txi = t_in - tyi * g.tiles_x;
    const int x0 = txi * kTileW, y0 = tyi * kTileH;
    int pitch;
    const uint8_t* img = raw_level_ptr(P, S, raw, frame, l, pitch);
    const bool can_dword = (l > 0) || S.aligned4;

    // ---- stage the (64+8) x (32+8) patch in LDS, BORDER_REFLECT_101 outside the level ----
    for (int i = tid; i < kPatchH * (kPatchW / 4); i += 256) {
        const int r = i / (kPatchW / 4), dcol = i - r * (kPatchW / 4);
        const int gy = reflect101(y0 - kHalo + r, g.h);
        const int gx = x0 - kHalo + dcol * 4;
        const uint8_t* rowp = img + (int64_t)gy * pitch;
        uint32_t w;
        if (can_dword && gx >= 0 && gx + 3 < g.w) {
            w = *reinterpret_cast<const uint32_t*>(rowp + gx);
        } else {
            w = (uint32_t)rowp[reflect101(gx, g.w)] | ((uint32_t)rowp[reflect101(gx + 1, g.w)] << 8) |
                ((uint32_t)rowp[reflect101(gx + 2, g.w)] << 16) | ((uint32_t)rowp[reflect101(gx + 3, g.w)] << 24);
        }
        *reinterpret_cast<uint32_t*>(&s_pix[r * kPatchW + dcol * 4]) = w;
    }
    __syncthreads();

    // ---- FAST score on the tile + 1 ring; only where a kept corner or its NMS neighbour can be ----
    // keypoints.cpp runByImageBorder keeps x in [31, w-31), y in [31, h-31); clears all if the level is <= 62
    const bool level_has_kp = (g.w > 2 * kEdgeThreshold) && (g.h > 2 * kEdgeThreshold);
    const int t = P.fast_threshold;
    constexpr int SW = kTileW + 4;  // score row pitch
    for (int i = tid; i < (kTileH + 2) * (kTileW + 2); i += 256) {
        const int sy = i / (kTileW + 2), sx = i - sy * (kTileW + 2);
        const int X = x0 - 1 + sx, Y = y0 - 1 + sy;
        int sc = 0;
        if (level_has_kp && X >= kEdgeThreshold - 1 && X <= g.w - kEdgeThreshold && Y >= kEdgeThreshold - 1 &&
            Y <= g.h - kEdgeThreshold)
            sc = fast9_score(&s_pix[(sy + kHalo - 1) * kPatchW + (sx + kHalo - 1)], t);
        s_score[sy * SW + sx] = (uint8_t)sc;
    }
    __syncthreads();

    // ---- 3x3 strict-max NMS + border filter + wave-aggregated append (ballot / popcount prefix) ----
    uint32_t* clist = cand + (int64_t)frame * P.cand_frame_entries + g.cand_off;
    int* ccnt = cand_cnt + frame * kLevels + l;
    const int lane = tid & 63;
    for (int i = tid; i < kTileH * kTileW; i += 256) {
        const int py = i >> 6, px = i & 63;
        const int X = x0 + px, Y = y0 + py;
        bool keep = false;
        int sc = 0;
        if (level_has_kp && X >= kEdgeThreshold && X < g.w - kEdgeThreshold && Y >= kEdgeThreshold &&
            Y < g.h - kEdgeThreshold) {
            const uint8_t* s = &s_score[(py + 1) * SW + (px + 1)];
            sc = s[0];
            keep = sc > 0 && sc > s[-1] && sc > s[1] && sc > s[-SW - 1] && sc > s[-SW] && sc > s[-SW + 1] &&
                   sc > s[SW - 1] && sc > s[SW] && sc > s[SW + 1];
        }
        const unsigned long long mask = __ballot(keep);
        if (mask) {
            const int leader = __ffsll((long long)mask) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(ccnt, __popcll(mask));
            base = __shfl(base, leader);
            const int off = __popcll(mask & ((1ull << lane) - 1ull));
            if (keep) {
                if (base + off < g.cand_cap)
                    clist[base + off] = (uint32_t)X | ((uint32_t)Y << 11) | ((uint32_t)sc << 22);
                else
                    atomicOr(err, ERRBIT_CAND_OVERFLOW);
            }
        }
    }

    // ---- 7x7 Gaussian, integer kernel {18,34,49,55,49,34,18} per pass (sum 257, not renormalised) ----
    // row pass: rows y0-3 .. y0+34 of the level = patch rows 1..38
    for (int i = tid; i < (kTileH + 6) * kTileW; i += 256) {
        const int r = i >> 6, cx = i & 63;
        const uint8_t* p = &s_pix[(r + 1) * kPatchW + cx + kHalo];
        const int s = 18 * (p[-3] + p[3]) + 34 * (p[-2] + p[2]) + 49 * (p[-1] + p[1]) + 55 * p[0];
        s_row[r * kTileW + cx] = (uint16_t)s;   // <= 255*257 = 65535
    }
    __syncthreads();
    uint8_t* bl = blur + (int64_t)frame * P.blur_frame_bytes + g.blur_off;
    const int body = g.w & ~3;   // SymmColumnVec_32s8u covers x < (w & ~3) with ties-to-even
    for (int i = tid; i < kTileH * (kTileW / 4); i += 256) {
        const int py = i >> 4, qx = (i & 15) * 4;
        const int X = x0 + qx, Y = y0 + py;
        if (Y >= g.h || X >= g.w) continue;
        uint32_t outw = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint16_t* c = &s_row[py * kTileW + qx + j];
            const int s = 18 * (c[0] + c[6 * kTileW]) + 34 * (c[kTileW] + c[5 * kTileW]) +
                          49 * (c[2 * kTileW] + c[4 * kTileW]) + 55 * c[3 * kTileW];
            int q = s >> 16;
            const int rem = s & 0xFFFF;
            if (rem > 32768) q += 1;
            else if (rem == 32768) q += (P.tie_mode == 1 && (X + j) < body) ? (q & 1) : 1;
            outw |= (uint32_t)min(q, 255) << (8 * j);
        }
        *reinterpret_cast<uint32_t*>(bl + (int64_t)Y * g.pitch + X) = outw;
    }
}

void launch_fast_blur_tile(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st, Profiler* prof) {
    ARIA_LAUNCH(prof, k_fast_blur, dim3(P.total_tiles, n_frames), dim3(256), 0, st, P, S, D.raw, D.blur, D.cand, D.cand_cnt, D.err);
}

}  // namespace aria
