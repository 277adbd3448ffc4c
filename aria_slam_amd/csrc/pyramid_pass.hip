// Stand-alone pyramid pass: level l from level l-1 by INTER_LINEAR_EXACT, one launch per level.
// The batch schedule does not use it (the FAST/blur launch of level l writes the raw rows of level l+1, fast_blur_band.hip);
// the single-frame latency schedule does (orb_api.hip: resize chain, then the 8 levels' FAST/blur launches as parallel
// graph branches), and so do ARIA_RESIZE_FUSE=0 / ARIA_RESIZE_IMPL / ARIA_PYRAMID_IMPL (bit-identical alternatives kept for
// tests/test_gpu_variants.py: direct gathers, shift/mad arithmetic, the all-levels-in-LDS pyramid).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <cstdlib>

#include "common.h"
#include "orb_device.h"
#include "orb_kernels.h"

namespace aria {

// ------------------------------------------------------------------------------------------------------
// a6.1  resize.cpp resize_bitExact<uchar, interpolationLinear>: H = c0*p[o] + c1*p[o+1] (exact, 8 frac bits),
//       out = (cy0*H0 + cy1*H1 + 32768) >> 16. One thread = 4 adjacent output pixels = one dword store.
// ------------------------------------------------------------------------------------------------------
// Two bytes p[o], p[o+1] out of a 12-byte window (w0,w1,w2) that starts at byte `base`; e = o - base in [0, 7].
__device__ __forceinline__ uint32_t window_pair(uint32_t w0, uint32_t w1, uint32_t w2, int e) {
    const uint32_t lo = e < 4 ? w0 : w1, hi = e < 4 ? w1 : w2;
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)e);    // uses e & 3
}

constexpr int kResizeRows = 8;   // output rows per thread

// One thread = 4 adjacent output pixels x 8 output rows. The level's x table and the block's slice of the y table
// are staged in LDS with one coalesced round trip; after that the only global accesses are the source rows
// (3 aligned dwords per source row, 4 rows = 24 loads in flight per lane) and the dword stores.
__global__ __launch_bounds__(256) void k_resize(Plan P, FrameSrc S, uint8_t* __restrict__ raw,
                                                const uint32_t* __restrict__ tab, int l) {
    extern __shared__ uint32_t s_rt[];        // [w] x table, then [kResizeRows * (rc1 - rc0 + 1)] y slice
    const LevelGeom g = P.lv[l];
    const int frame = blockIdx.y;
    const int groups = g.pitch >> 2;
    const int nrc = (g.h + kResizeRows - 1) / kResizeRows;
    const int total = groups * nrc;
    const int gid0 = blockIdx.x * 256;
    const int rc_first = gid0 / groups, rc_last = min(gid0 + 255, total - 1) / groups;
    uint32_t* s_yt = s_rt + g.w;
    for (int i = threadIdx.x; i < g.w; i += 256) s_rt[i] = tab[g.xtab + i];
    {
        const int y0 = rc_first * kResizeRows, ny = min((rc_last + 1) * kResizeRows, g.h) - y0;
        for (int i = threadIdx.x; i < ny; i += 256) s_yt[i] = tab[g.ytab + y0 + i];
    }
    __syncthreads();
    const int gid = gid0 + threadIdx.x;
    if (gid >= total) return;
    const int rc = gid / groups, gx = gid - rc * groups;
    int spitch;
    const uint8_t* src = raw_level_ptr(P, S, raw, frame, l - 1, spitch);
    const int sw = P.lv[l - 1].w, sh = P.lv[l - 1].h;
    uint8_t* dst = raw + (int64_t)frame * P.raw_frame_bytes + g.raw_off + gx * 4;
    const int dx0 = gx * 4;
    const int dy0 = rc * kResizeRows;
    const uint32_t* yt = s_yt + (rc - rc_first) * kResizeRows;
    if (dx0 >= g.w) {   // row padding up to the 16-byte pitch: keep it deterministic
        for (int r = 0; r < kResizeRows && dy0 + r < g.h; r++) *reinterpret_cast<uint32_t*>(dst + (int64_t)(dy0 + r) * g.pitch) = 0u;
        return;
    }
    uint32_t tx[4];
#pragma unroll
    for (int i = 0; i < 4; i++) tx[i] = s_rt[min(dx0 + i, g.w - 1)];
    const int base = (int)(tx[0] & 0xFFFF) & ~3;
    int e[4], cx1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { e[i] = (int)(tx[i] & 0xFFFF) - base; cx1[i] = (int)(tx[i] >> 16); }
    // dword path: 3 aligned dwords per source row cover the <= 9 source bytes four outputs need (scale ~1.2). Only
    // the last threads of a row of an unpadded / unaligned caller image take the byte path.
    const bool fast = (l > 1 || S.aligned4) && (base + 12 <= spitch) && e[3] <= 7;
    if (fast) {
#pragma unroll 4
        for (int r = 0; r < kResizeRows; r++) {
            const int dy = dy0 + r;
            if (dy < g.h) {
                const uint32_t ty = yt[r];
                const int oy = ty & 0xFFFF, cy1 = (int)(ty >> 16);
                const uint32_t* q0 = reinterpret_cast<const uint32_t*>(src + (int64_t)oy * spitch + base);
                const uint32_t* q1 = reinterpret_cast<const uint32_t*>(src + (int64_t)min(oy + 1, sh - 1) * spitch + base);
                const uint32_t a0 = q0[0], a1 = q0[1], a2 = q0[2];
                const uint32_t b0 = q1[0], b1 = q1[1], b2 = q1[2];
                uint32_t outw = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t pa = window_pair(a0, a1, a2, e[i]), pb = window_pair(b0, b1, b2, e[i]);
                    const int p00 = pa & 0xFF, p01 = (pa >> 8) & 0xFF, p10 = pb & 0xFF, p11 = (pb >> 8) & 0xFF;
                    // c0*p0 + c1*p1 with c0 = 256 - c1  ==  256*p0 + c1*(p1 - p0)   (exact, same integers)
                    const int h0 = (p00 << 8) + cx1[i] * (p01 - p00);
                    const int h1 = (p10 << 8) + cx1[i] * (p11 - p10);
                    const uint32_t v = (uint32_t)((h0 << 8) + cy1 * (h1 - h0) + 32768) >> 16;
                    outw |= min(v, 255u) << (8 * i);
                }
                *reinterpret_cast<uint32_t*>(dst + (int64_t)dy * g.pitch) = outw;
            }
        }
    } else {
        for (int r = 0; r < kResizeRows; r++) {
            const int dy = dy0 + r;
            if (dy >= g.h) break;
            const uint32_t ty = yt[r];
            const int oy = ty & 0xFFFF;
            const uint32_t cyy1 = ty >> 16, cyy0 = 256u - cyy1;
            const uint8_t* r0 = src + (int64_t)oy * spitch;
            const uint8_t* r1 = src + (int64_t)min(oy + 1, sh - 1) * spitch;
            uint32_t outw = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (dx0 + i < g.w) {
                    const int ox = tx[i] & 0xFFFF, ox1 = min(ox + 1, sw - 1);
                    const uint32_t c1 = tx[i] >> 16, c0 = 256u - c1;
                    const uint32_t h0 = c0 * r0[ox] + c1 * r0[ox1];
                    const uint32_t h1 = c0 * r1[ox] + c1 * r1[ox1];
                    const uint32_t v = (cyy0 * h0 + cyy1 * h1 + 32768u) >> 16;
                    outw |= min(v, 255u) << (8 * i);
                }
            }
            *reinterpret_cast<uint32_t*>(dst + (int64_t)dy * g.pitch) = outw;
        }
    }
}

// a6.1, LDS-staged form: a workgroup produces kResizeBand output rows of level l. The source rows it needs
// (~1.2 * band + 2) are staged with coalesced 16-byte loads, 4 in flight per lane, so the texture path sees wide
// contiguous requests instead of three overlapping dword gathers per lane; the x table and the band's y slice ride
// along in LDS; the bilinear taps are then dword windows read from LDS.
//
// DOT2 (default): the per-pixel arithmetic runs on v_perm_b32 + v_dot2_u32_u16. The x table is widened at staging
// time to three words per output column -- (256-cx1) | cx1 << 16, the dword-aligned source offset, and the v_perm
// selector that lifts the two source bytes into two u16 lanes -- kept as three arrays so that a lane fetches the four
// columns of its output dword with three conflict-free ds_read_b128. A pixel is then: two ds_read2_b32 (source
// dword pair in both rows), 2 v_perm + 2 v_dot2 (horizontal pass, both rows), 1 v_lshl_or + 1 v_dot2 with
// the rounding constant as accumulator (vertical pass), 1 v_perm that drops the result byte into the output dword:
// 9 vector-ALU instructions per pixel where the shift/extract/mad form needed ~30 (this kernel is VALU-issue bound).
// Same integers as the reference's ufixedpoint16 arithmetic: (256-c)*p0 + c*p1 is what it evaluates.
constexpr int kResizeBand = 16;
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

template <bool DOT2>
__global__ __launch_bounds__(256) void k_resize_lds(Plan P, FrameSrc S, uint8_t* __restrict__ raw,
                                                    const uint32_t* __restrict__ tab, int l) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_rs[];
    const LevelGeom g = P.lv[l];
    const int tid = threadIdx.x;
    const int frame = blockIdx.y;
    const int dy0 = blockIdx.x * kResizeBand, ndy = min(kResizeBand, g.h - dy0);
    int spitch;
    const uint8_t* src = raw_level_ptr(P, S, raw, frame, l - 1, spitch);
    const int sw = P.lv[l - 1].w, sh = P.lv[l - 1].h;
    const int lp = (sw + 15) / 16 * 16 + 16;                     // LDS pitch of a staged source row (+16: window slack)
    const int max_rows = (kResizeBand * 13) / 10 + 4;            // rows a band can need at scale ~1.2
    uint32_t* s_xt = reinterpret_cast<uint32_t*>(smem_rs + max_rows * lp);      // 16-byte aligned: lp is a multiple of 16
    const int w4 = (g.w + 3) & ~3;
    uint32_t* s_yt = s_xt + (DOT2 ? 3 * w4 : g.w);
    const uint32_t* ytg = tab + g.ytab + dy0;
    const int oy_lo = (int)(ytg[0] & 0xFFFF);
    const int oy_hi = min((int)(ytg[ndy - 1] & 0xFFFF) + 1, sh - 1);
    const int nrows = min(oy_hi - oy_lo + 1, max_rows);
    if (DOT2) {
        for (int i = tid; i < w4; i += 256) {                 // columns past the level repeat the last one
            const uint32_t t = tab[g.xtab + min(i, g.w - 1)];
            const uint32_t ox = t & 0xFFFFu, cx1 = t >> 16;
            s_xt[i] = (256u - cx1) | (cx1 << 16);
            s_xt[w4 + i] = ox & ~3u;
            s_xt[2 * w4 + i] = 0x0C010C00u + (ox & 3u) * 0x00010001u;
        }
    } else {
        for (int i = tid; i < g.w; i += 256) s_xt[i] = tab[g.xtab + i];
    }
    if (tid < ndy) s_yt[tid] = ytg[tid];
    const bool a16 = (l > 1) || S.aligned16;
    const int nch = a16 ? (sw >> 4) : 0;
    if (nch > 0 && nch <= 256) {
        const int rpp = 256 / nch;
        const int my_r = tid / nch, my_c = tid - my_r * nch;
        if (my_r < rpp) {
            for (int r0 = my_r; r0 < nrows; r0 += 4 * rpp) {
                uint4 v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int r = r0 + k * rpp;
                    if (r < nrows) v[k] = *reinterpret_cast<const uint4*>(src + (int64_t)(oy_lo + r) * spitch + 16 * my_c);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int r = r0 + k * rpp;
                    if (r < nrows) *reinterpret_cast<uint4*>(smem_rs + r * lp + 16 * my_c) = v[k];
                }
            }
        }
    }
    {
        const int xe0 = (nch > 0 && nch <= 256) ? nch * 16 : 0;
        const int ne = sw - xe0;
        for (int i = tid; i < nrows * ne; i += 256) {
            const int r = i / ne, c = xe0 + (i - r * ne);
            smem_rs[r * lp + c] = src[(int64_t)(oy_lo + r) * spitch + c];
        }
    }
    __syncthreads();

    uint8_t* dg = raw + (int64_t)frame * P.raw_frame_bytes + g.raw_off;
    const int groups = g.pitch >> 2;
    const int items = ndy * groups;
    const float inv_groups = 1.0f / (float)groups;
    for (int it = tid; it < items; it += 256) {
        int r = (int)((float)it * inv_groups);
        if (r * groups > it) r--;
        else if ((r + 1) * groups <= it) r++;
        const int gx = it - r * groups;
        const int dx0 = gx * 4;
        uint32_t outw = 0;
        if (DOT2) {
            if (dx0 < g.w) {
                const uint32_t ty = s_yt[r];
                const int oy = ty & 0xFFFF;
                const uint32_t cy1 = ty >> 16, cyp = (256u - cy1) | (cy1 << 16);
                const int ra = min(oy - oy_lo, nrows - 1), rb = min(min(oy + 1, sh - 1) - oy_lo, nrows - 1);
                const uint8_t* rowa = smem_rs + ra * lp;
                const uint8_t* rowb = smem_rs + rb * lp;
                const uint4 xw = *reinterpret_cast<const uint4*>(s_xt + dx0);
                const uint4 xo = *reinterpret_cast<const uint4*>(s_xt + w4 + dx0);
                const uint4 xs = *reinterpret_cast<const uint4*>(s_xt + 2 * w4 + dx0);
                const uint32_t xwv[4] = {xw.x, xw.y, xw.z, xw.w}, xov[4] = {xo.x, xo.y, xo.z, xo.w},
                               xsv[4] = {xs.x, xs.y, xs.z, xs.w};
                constexpr uint32_t put[4] = {0x03020106u, 0x03020600u, 0x03060100u, 0x06020100u};   // byte 2 of v -> byte i
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t* qa = reinterpret_cast<const uint32_t*>(rowa + xov[i]);
                    const uint32_t* qb = reinterpret_cast<const uint32_t*>(rowb + xov[i]);
                    const uint32_t top = __builtin_amdgcn_perm(qa[1], qa[0], xsv[i]);     // p00 | p01 << 16
                    const uint32_t bot = __builtin_amdgcn_perm(qb[1], qb[0], xsv[i]);
                    const us2 wx = __builtin_bit_cast(us2, xwv[i]);
                    const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, top), wx, 0u, false);
                    const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, bot), wx, 0u, false);
                    const uint32_t v = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, h0 | (h1 << 16)),
                                                              __builtin_bit_cast(us2, cyp), 32768u, false);   // < 2^24
                    outw = __builtin_amdgcn_perm(v, outw, put[i]);
                }
            }
        } else if (dx0 < g.w) {
            const uint32_t ty = s_yt[r];
            const int oy = ty & 0xFFFF, cy1 = (int)(ty >> 16);
            uint32_t tx[4];
#pragma unroll
            for (int i = 0; i < 4; i++) tx[i] = s_xt[min(dx0 + i, g.w - 1)];
            const int base = (int)(tx[0] & 0xFFFF) & ~3;
            const int ra = min(oy - oy_lo, nrows - 1), rb = min(min(oy + 1, sh - 1) - oy_lo, nrows - 1);
            const uint32_t* q0 = reinterpret_cast<const uint32_t*>(smem_rs + ra * lp + base);
            const uint32_t* q1 = reinterpret_cast<const uint32_t*>(smem_rs + rb * lp + base);
            const uint32_t a0 = q0[0], a1 = q0[1], a2 = q0[2];
            const uint32_t b0 = q1[0], b1 = q1[1], b2 = q1[2];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int e = min((int)(tx[i] & 0xFFFF) - base, 7), cx1 = (int)(tx[i] >> 16);
                const uint32_t pa = window_pair(a0, a1, a2, e), pb = window_pair(b0, b1, b2, e);
                const int p00 = pa & 0xFF, p01 = (pa >> 8) & 0xFF, p10 = pb & 0xFF, p11 = (pb >> 8) & 0xFF;
                const int h0 = (p00 << 8) + cx1 * (p01 - p00);      // == (256-cx1)*p00 + cx1*p01
                const int h1 = (p10 << 8) + cx1 * (p11 - p10);
                const uint32_t v = (uint32_t)((h0 << 8) + cy1 * (h1 - h0) + 32768) >> 16;
                outw |= min(v, 255u) << (8 * i);
            }
        }
        *reinterpret_cast<uint32_t*>(dg + (int64_t)(dy0 + r) * g.pitch + dx0) = outw;
    }
}

// a6.1 fused: one workgroup builds levels 1..7 for a band of level-0 rows entirely in LDS: the level-0 band, the x
// tables and each level's y-table slice are staged once, level l is resized from the level l-1 rows the same workgroup
// just produced, and every row it owns goes to HBM exactly once. Nothing is re-read from HBM between levels, no item
// waits on a global load, and the 7 dependent launches collapse into one.
__global__ __launch_bounds__(256) void k_pyramid(Plan P, FrameSrc S, uint8_t* __restrict__ raw,
                                                 const uint32_t* __restrict__ tab, const int* __restrict__ bands,
                                                 int* __restrict__ clr_a, int n_a, int* __restrict__ clr_b, int n_b,
                                                 const uint8_t* __restrict__ src_alt, uint8_t* __restrict__ copy_dst) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_pyr[];
    const int tid = threadIdx.x;
    const int frame = blockIdx.y;
    // single-frame latency schedule: this is the first kernel of the pass, the counters of the later ones are zeroed here
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (tid < n_a) clr_a[tid] = 0;
        if (tid < n_b) clr_b[tid] = 0;
    }
    const int* B = bands + (int)blockIdx.x * kLevels * 4;
    uint32_t* s_xt = reinterpret_cast<uint32_t*>(smem_pyr + P.pyr_xtab_off);
    uint32_t* s_yt = reinterpret_cast<uint32_t*>(smem_pyr + P.pyr_ytab_off);
    const int xt_lo = P.lv[1].xtab;

    // ---- stage: x tables, y slice of level 1, level-0 rows [comp_lo, comp_lo + comp_n) ----
    for (int i = tid; i < P.pyr_xtab_n; i += 256) s_xt[i] = tab[xt_lo + i];
    {
        const int clo = B[4 + 0], cn = B[4 + 1];
        if (tid < cn) s_yt[tid] = tab[P.lv[1].ytab + clo + tid];
    }
    {
        const int lo0 = B[0], n0 = B[1], w0 = P.lv[0].w, p0 = P.pyr_p0;
        // src_alt (single-frame latency schedule): level 0 comes from the pinned host copy of the frame, and the rows this
        // band owns are also written to the device copy (copy_dst) for the kernels that follow -- the upload rides along
        const uint8_t* img = (src_alt ? src_alt : S.img) + (int64_t)frame * S.frame_stride;
        uint8_t* cpy = copy_dst ? copy_dst + (int64_t)frame * S.frame_stride : nullptr;
        const int own0 = B[3];
        uint8_t* d0 = smem_pyr + P.pyr_off[0];
        const int nch = S.aligned16 ? (w0 >> 4) : 0;
        if (nch > 0) {
            const int rpp = 256 / nch > 0 ? 256 / nch : 1;
            const int my_r = tid / nch, my_c = tid - my_r * nch;
            if (my_r < rpp && nch <= 256) {
                for (int r0 = my_r; r0 < n0; r0 += 4 * rpp) {
                    uint4 v[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int r = r0 + k * rpp;
                        if (r < n0) v[k] = *reinterpret_cast<const uint4*>(img + (int64_t)(lo0 + r) * S.row_stride + 16 * my_c);
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int r = r0 + k * rpp;
                        if (r < n0) *reinterpret_cast<uint4*>(d0 + r * p0 + 16 * my_c) = v[k];
                        if (cpy && r < own0) *reinterpret_cast<uint4*>(cpy + (int64_t)(lo0 + r) * S.row_stride + 16 * my_c) = v[k];
                    }
                }
            }
        }
        const int xe0 = (nch > 0 && nch <= 256) ? nch * 16 : 0;     // columns not covered by the 16-byte chunks
        const int ne = w0 - xe0;
        for (int i = tid; i < n0 * ne; i += 256) {
            const int r = i / ne, c = xe0 + (i - r * ne);
            const uint8_t px = img[(int64_t)(lo0 + r) * S.row_stride + c];
            d0[r * p0 + c] = px;
            if (cpy && r < own0) cpy[(int64_t)(lo0 + r) * S.row_stride + c] = px;
        }
    }
    __syncthreads();

    for (int l = 1; l < kLevels; l++) {
        const LevelGeom g = P.lv[l];
        const int comp_lo = B[l * 4 + 0], comp_n = B[l * 4 + 1], own_n = B[l * 4 + 3];
        const int src_lo = B[(l - 1) * 4 + 0];
        const int sh = P.lv[l - 1].h;
        const int spitch = l == 1 ? P.pyr_p0 : P.lv[l - 1].pitch;
        const uint8_t* src = smem_pyr + P.pyr_off[l - 1];      // LDS row 0 is level row src_lo
        uint8_t* dl = smem_pyr + P.pyr_off[l];
        uint8_t* dg = raw + (int64_t)frame * P.raw_frame_bytes + g.raw_off;
        const uint32_t* xt = s_xt + (g.xtab - xt_lo);
        const int groups = g.pitch >> 2;
        const int items = comp_n * groups;
        const float inv_groups = 1.0f / (float)groups;
        for (int it = tid; it < items; it += 256) {
            int r = (int)((float)it * inv_groups);
            if (r * groups > it) r--;
            else if ((r + 1) * groups <= it) r++;
            const int gx = it - r * groups;
            const int dy = comp_lo + r;
            const int dx0 = gx * 4;
            uint32_t outw = 0;
            if (dx0 < g.w) {
                const uint32_t ty = s_yt[r];
                const int oy = ty & 0xFFFF, cy1 = (int)(ty >> 16);
                uint32_t tx[4];
#pragma unroll
                for (int i = 0; i < 4; i++) tx[i] = xt[min(dx0 + i, g.w - 1)];
                const int base = (int)(tx[0] & 0xFFFF) & ~3;
                // 3 aligned dwords per source row cover the <= 9 source bytes four outputs need (scale ~1.2); a window
                // may run past the row's end into the next LDS row: those bytes only ever meet weight 0
                const uint32_t* q0 = reinterpret_cast<const uint32_t*>(src + (oy - src_lo) * spitch + base);
                const uint32_t* q1 = reinterpret_cast<const uint32_t*>(src + (min(oy + 1, sh - 1) - src_lo) * spitch + base);
                const uint32_t a0 = q0[0], a1 = q0[1], a2 = q0[2];
                const uint32_t b0 = q1[0], b1 = q1[1], b2 = q1[2];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int e = min((int)(tx[i] & 0xFFFF) - base, 7), cx1 = (int)(tx[i] >> 16);
                    const uint32_t pa = window_pair(a0, a1, a2, e), pb = window_pair(b0, b1, b2, e);
                    const int p00 = pa & 0xFF, p01 = (pa >> 8) & 0xFF, p10 = pb & 0xFF, p11 = (pb >> 8) & 0xFF;
                    const int h0 = (p00 << 8) + cx1 * (p01 - p00);      // == (256-cx1)*p00 + cx1*p01
                    const int h1 = (p10 << 8) + cx1 * (p11 - p10);
                    const uint32_t v = (uint32_t)((h0 << 8) + cy1 * (h1 - h0) + 32768) >> 16;
                    outw |= min(v, 255u) << (8 * i);
                }
            }
            *reinterpret_cast<uint32_t*>(dl + r * g.pitch + dx0) = outw;
            if (r < own_n) *reinterpret_cast<uint32_t*>(dg + (int64_t)dy * g.pitch + dx0) = outw;
        }
        __syncthreads();
        if (l + 1 < kLevels) {       // y slice of the next level (the slice just used is dead after the barrier)
            const int clo = B[(l + 1) * 4 + 0], cn = B[(l + 1) * 4 + 1];
            if (tid < cn) s_yt[tid] = tab[P.lv[l + 1].ytab + clo + tid];
            __syncthreads();
        }
    }
}

int pyramid_set_attributes() {
    ARIA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pyramid), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    return ARIA_OK;
}

// the whole pyramid in one launch (levels kept in LDS, bands of level-0 rows); false when its bands do not fit the LDS
bool pyramid_fused_available(const Plan& P) { return P.pyr_nbands > 0 && P.pyr_lds_bytes <= 150 * 1024; }

bool launch_pyramid_fused(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st, Profiler* prof,
                          int* clr_a, int n_a, int* clr_b, int n_b, const uint8_t* src_alt, uint8_t* copy_dst) {
    if (!pyramid_fused_available(P) || n_a > 256 || n_b > 256) return false;
    ARIA_LAUNCH(prof, k_pyramid, dim3(P.pyr_nbands, n_frames), dim3(256), (size_t)P.pyr_lds_bytes, st, P, S, D.raw, D.tab, D.pyr_bands,
                clr_a, n_a, clr_b, n_b, src_alt, copy_dst);
    return true;
}

// one level of the default stand-alone resize (dot2 LDS bands; direct gathers when a band does not fit 64 KB of LDS)
void launch_pyramid_level(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st, Profiler* prof, int l) {
    const int lp = (P.lv[l - 1].w + 15) / 16 * 16 + 16;
    const size_t xt_words = 3 * (size_t)((P.lv[l].w + 3) & ~3);
    const size_t lds = (size_t)((kResizeBand * 13) / 10 + 4) * lp + sizeof(uint32_t) * (xt_words + kResizeBand) + 16;
    if (lds <= 64 * 1024) {
        const dim3 grid((P.lv[l].h + kResizeBand - 1) / kResizeBand, n_frames);
        ARIA_LAUNCH(prof, k_resize_lds<true>, grid, dim3(256), lds, st, P, S, D.raw, D.tab, l);
        return;
    }
    const int items = (P.lv[l].pitch >> 2) * ((P.lv[l].h + kResizeRows - 1) / kResizeRows);
    const int groups = P.lv[l].pitch >> 2;
    const size_t lds2 = sizeof(uint32_t) * ((size_t)P.lv[l].w + (size_t)kResizeRows * (256 / groups + 2));
    ARIA_LAUNCH(prof, k_resize, dim3((items + 255) / 256, n_frames), dim3(256), lds2, st, P, S, D.raw, D.tab, l);
}

#ifdef ARIA_VARIANTS
// the stand-alone pyramid pass in its measured forms (variants build only: the product fuses the pyramid step into the
// FAST/blur launches and uses launch_pyramid_level / k_pyramid in the single-frame schedule)
void launch_pyramid_pass(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st, Profiler* prof) {
    const EnvConfig& E = env_config();
    const int pyr_impl = E.pyr_impl;   // 0 = one resize launch per level (default); 1 = fused LDS pyramid (ARIA_PYRAMID_IMPL=fused,
                                       // measured slower at 640x480: the top-down halo makes small bands recompute too much)
    if (pyr_impl == 1 && P.pyr_lds_bytes <= 150 * 1024) {
        ARIA_LAUNCH(prof, k_pyramid, dim3(P.pyr_nbands, n_frames), dim3(256), (size_t)P.pyr_lds_bytes, st, P, S,
                    D.raw, D.tab, D.pyr_bands, (int*)nullptr, 0, (int*)nullptr, 0, (const uint8_t*)nullptr, (uint8_t*)nullptr);
    } else {
        // 2 = LDS-staged bands with dot2 arithmetic (default), 1 = LDS-staged bands with shift/mad arithmetic
        // (ARIA_RESIZE_IMPL=lds), 0 = direct global gathers (ARIA_RESIZE_IMPL=direct)
        const int rs_impl = E.rs_impl;
        for (int l = 1; l < kLevels; l++) {
            if (rs_impl >= 1) {
                const int lp = (P.lv[l - 1].w + 15) / 16 * 16 + 16;
                const size_t xt_words = rs_impl == 2 ? 3 * (size_t)((P.lv[l].w + 3) & ~3) : (size_t)P.lv[l].w;
                const size_t lds = (size_t)((kResizeBand * 13) / 10 + 4) * lp + sizeof(uint32_t) * (xt_words + kResizeBand) + 16;
                if (lds <= 64 * 1024) {
                    const dim3 grid((P.lv[l].h + kResizeBand - 1) / kResizeBand, n_frames);
                    if (rs_impl == 2) ARIA_LAUNCH(prof, k_resize_lds<true>, grid, dim3(256), lds, st, P, S, D.raw, D.tab, l);
                    else ARIA_LAUNCH(prof, k_resize_lds<false>, grid, dim3(256), lds, st, P, S, D.raw, D.tab, l);
                    continue;
                }
            }
            const int items = (P.lv[l].pitch >> 2) * ((P.lv[l].h + kResizeRows - 1) / kResizeRows);
            dim3 grid((items + 255) / 256, n_frames);
            // LDS: the level's x table + the y entries of the row-chunks a block can touch (256 items span <= 256/groups + 2 chunks)
            const int groups = P.lv[l].pitch >> 2;
            const size_t lds = sizeof(uint32_t) * ((size_t)P.lv[l].w + (size_t)kResizeRows * (256 / groups + 2));
            ARIA_LAUNCH(prof, k_resize, grid, dim3(256), lds, st, P, S, D.raw, D.tab, l);
        }
    }

}
#endif  // ARIA_VARIANTS

}  // namespace aria
