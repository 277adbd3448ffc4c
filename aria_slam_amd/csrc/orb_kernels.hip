// HIP kernels of the ORB extractor for gfx950 (MI355X). Wave = 64 lanes throughout.
//
// Stage map (algorithm = CPU cv::ORB of OpenCV 4.9.0):
//   (fast_blur_band.hip: a6.1 pyramid step, a6.2 FAST-9 + NMS, a6.3 border filter, a6.7 7x7 Gaussian -- the dominant kernel;
//    pyramid_pass.hip: the stand-alone a6.1 resize pass of the single-frame latency schedule)
//   k_select      a6.3  retainBest(2*quota) by FAST score (256-bin histogram cut, ties kept)
//                 a6.4  Harris response (block 7, k 0.04) of the survivors
//                 a6.5  retainBest(quota) by Harris (LDS bitonic sort, ties kept) -> canonical order
//   k_select_ovf  the same selection in global memory for (frame, level) pairs whose ties overflow k_select's LDS capacities
//   k_describe    a6.6  intensity-centroid angle (one keypoint per 16-lane DPP row, integer moments, fastAtan2)
//                 a6.8  256-bit rBRIEF: lane l16 of a row evaluates tests 16*it + l16; one 64-bit ballot per iteration
//                       carries 16 descriptor bits for each of the wave's four keypoints (LSB-first bytes)
// All of it is HBM/LDS/VALU-integer work; nothing here is a dense contraction, so no MFMA.
// Float stages are compiled with -ffp-contract=off so that they round exactly like the SSE3-baseline
// OpenCV build the CPU reference path uses (no FMA).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <cstdlib>

#include "common.h"
#include "orb_device.h"
#include "orb_kernels.h"

namespace aria {

// The rBRIEF pattern, 4 signed bytes (x0, y0, x1, y1) per test, laid out for k_describe: lane l16 evaluates tests
// 16*it + l16, it = 0..15, and fetches its 16 dwords with four 16-byte loads (the texture-address path charges per
// instruction, not per byte): word [l16][it] = test 16*it + l16.
struct PatternLaneMajor { uint32_t w[16][16]; };
constexpr PatternLaneMajor make_pattern_lane_major() {
    constexpr int src[1024] = {
#include "orb_pattern_31.inc"
    };
    PatternLaneMajor p{};
    for (int l = 0; l < 16; l++)
        for (int it = 0; it < 16; it++) {
            const int t = 16 * it + l;
            p.w[l][it] = ((uint32_t)(src[4 * t] & 0xFF)) | ((uint32_t)(src[4 * t + 1] & 0xFF) << 8) |
                         ((uint32_t)(src[4 * t + 2] & 0xFF) << 16) | ((uint32_t)(src[4 * t + 3] & 0xFF) << 24);
        }
    return p;
}
__device__ __attribute__((aligned(16))) const PatternLaneMajor kPattern31 = make_pattern_lane_major();

// end-of-row table of the radius-15 disc (orb.cpp computeKeyPoints umax; verified against the host
// computation in tests/test_host_logic.py)
#define ARIA_UMAX_LIST {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3}

// Byte masks of the radius-15 disc for k_describe's dword-wise IC moments, laid out like the pattern: lane
// (hp = l16 >> 3, j = l16 & 7) owns patch columns u = 4j-15 .. 4j-12 of the rows v = 2k + hp - 15, k = 0..15, and
// fetches its 16 masks [l16][k] with four 16-byte loads. 0xFF where |u| <= umax[|v|]; row 31 (k = 15, hp = 1) is zero.
struct IcMaskLaneMajor { uint32_t m[16][16]; };
constexpr IcMaskLaneMajor make_ic_mask() {
    constexpr int umax[16] = ARIA_UMAX_LIST;
    IcMaskLaneMajor t{};
    for (int l = 0; l < 16; l++)
        for (int k = 0; k < 16; k++) {
            const int r = 2 * k + (l >> 3), j = l & 7;
            uint32_t w = 0;
            for (int b = 0; b < 4 && r < 31; b++) {
                const int u = 4 * j + b - 15, v = r - 15;
                const int au = u < 0 ? -u : u, av = v < 0 ? -v : v;
                if (au <= 15 && au <= umax[av]) w |= 0xFFu << (8 * b);
            }
            t.m[l][k] = w;
        }
    return t;
}
__device__ __attribute__((aligned(16))) const IcMaskLaneMajor kIcMask = make_ic_mask();

// ------------------------------------------------------------------------------------------------------
// a6.3-a6.5  one workgroup per (frame, level)
// ------------------------------------------------------------------------------------------------------
// One 9-pixel row of the Harris neighbourhood as five int16 pairs E[k] = (p[2k], p[2k+1]) (E[4] = (p[8], 0)): the row is
// fetched as 3 aligned dwords (scattered byte loads are texture-path bound), cut out with v_alignbyte and widened with
// v_perm. dword_ok = the level's rows are 4-byte aligned.
typedef short s2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void harris_row_pairs(const uint8_t* rowp, int xs, int sh, bool dword_ok, uint32_t* E) {
    uint32_t w0, w1, w2;
    if (dword_ok) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(rowp + xs);
        w0 = q[0]; w1 = q[1]; w2 = q[2];
    } else {
        const uint8_t* q = rowp + xs;
        w0 = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
        w1 = (uint32_t)q[4] | ((uint32_t)q[5] << 8) | ((uint32_t)q[6] << 16) | ((uint32_t)q[7] << 24);
        w2 = (uint32_t)q[8] | ((uint32_t)q[9] << 8) | ((uint32_t)q[10] << 16) | ((uint32_t)q[11] << 24);
    }
    const uint32_t a = __builtin_amdgcn_alignbyte(w1, w0, (uint32_t)sh);   // bytes sh .. sh+3   = columns 0..3
    const uint32_t b = __builtin_amdgcn_alignbyte(w2, w1, (uint32_t)sh);   // bytes sh+4 .. sh+7 = columns 4..7
    E[0] = __builtin_amdgcn_perm(0u, a, 0x0c010c00u);
    E[1] = __builtin_amdgcn_perm(0u, a, 0x0c030c02u);
    E[2] = __builtin_amdgcn_perm(0u, b, 0x0c010c00u);
    E[3] = __builtin_amdgcn_perm(0u, b, 0x0c030c02u);
    E[4] = (w2 >> (8 * sh)) & 0xFFu;                                        // byte sh+8 = column 8
}

__device__ __forceinline__ uint32_t pk16_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2v, a) + __builtin_bit_cast(s2v, b));
}
__device__ __forceinline__ uint32_t pk16_sub(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2v, a) - __builtin_bit_cast(s2v, b));
}

// orb.cpp HarrisResponses(blockSize 7, k 0.04f): integer Sobel sums a = sum Ix^2, b = sum Iy^2, c = sum Ix Iy over the
// 7 x 7 block, then the float formula. Separable and two columns per op in packed int16 (|Ix|, |Iy| <= 1020):
//   s = p[y-1] + 2 p[y] + p[y+1],  Ix[j] = s[j+1] - s[j-1];   d = p[y+1] - p[y-1],  Iy[j] = d[j-1] + 2 d[j] + d[j+1]
// with the even-aligned pairs E[k] = (col 2k, 2k+1) and the odd-aligned O[k] = (col 2k+1, 2k+2) cut out of them by
// v_alignbit; the products are accumulated two at a time by v_dot2_i32_i16. Same integers as the scalar form.
__device__ __forceinline__ float harris_response(const uint8_t* img, int pitch, int x, int y, bool dword_ok) {
    int a = 0, b = 0, c = 0;
    const int xs = (x - 4) & ~3, sh = (x - 4) & 3;
    const uint8_t* base = img + (int64_t)(y - 4) * pitch;
    uint32_t R[9][5];
#pragma unroll
    for (int r = 0; r < 9; r++) harris_row_pairs(base + (int64_t)r * pitch, xs, sh, dword_ok, R[r]);
#pragma unroll
    for (int i = 1; i <= 7; i++) {
        uint32_t sE[5], dE[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            sE[k] = pk16_add(pk16_add(R[i - 1][k], R[i + 1][k]), pk16_add(R[i][k], R[i][k]));
            dE[k] = pk16_sub(R[i + 1][k], R[i - 1][k]);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // columns j = 2k+1, 2k+2 (the pair k = 3 holds j = 7 and a column-8 term that is masked off)
            uint32_t ix = pk16_sub(sE[k + 1], sE[k]);                                   // s[j+1] - s[j-1]
            const uint32_t dO = __builtin_amdgcn_alignbit(dE[k + 1], dE[k], 16);        // (d[2k+1], d[2k+2])
            uint32_t iy = pk16_add(pk16_add(dE[k], dE[k + 1]), pk16_add(dO, dO));       // d[j-1] + 2 d[j] + d[j+1]
            if (k == 3) { ix &= 0xFFFFu; iy &= 0xFFFFu; }
            const s2v vx = __builtin_bit_cast(s2v, ix), vy = __builtin_bit_cast(s2v, iy);
            a = __builtin_amdgcn_sdot2(vx, vx, a, false);
            b = __builtin_amdgcn_sdot2(vy, vy, b, false);
            c = __builtin_amdgcn_sdot2(vx, vy, c, false);
        }
    }
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    const float k = 0.04f;
    return ((float)a * (float)b - (float)c * (float)c - k * ((float)a + (float)b) * ((float)a + (float)b)) * scale_sq_sq;
}

__device__ void select_ovf_item(const Plan& P, const FrameSrc& S, const uint8_t* __restrict__ raw,
                                const uint32_t* __restrict__ cand, const int* __restrict__ cand_cnt,
                                uint4* __restrict__ sel, int* __restrict__ sel_cnt, int* __restrict__ err,
                                int* __restrict__ ovf, unsigned long long* __restrict__ keys, long long keys_cap,
                                uint4* __restrict__ osel, int osel_cap, int frame, int l, int* s_hist, int* s_misc);

// INLINE_OVF = false (batches): tie storms go to a work list and k_select_ovf. true (single-frame latency schedule: one
// launch less): the overflowing workgroup runs the global-memory selection itself; ovf_items then carries the key arena.
// NT = threads per workgroup (the three prefix sums involve the first 256 threads only). 256 everywhere: 1024-thread
// workgroups were tried for the single-frame latency schedule (8 workgroups on the whole chip) and lost -- the phases are
// chains of dependent round trips, not trip counts (Harris 20k cycles in 2 rounds against 17k in 5), and with the inline
// tie-storm path's scratch a 16-wave workgroup waits ~40k cycles before its first instruction (37 vs 25 us).
template <bool INLINE_OVF, int NT>
__global__ __launch_bounds__(NT) void k_select(Plan P, FrameSrc S, const uint8_t* __restrict__ raw,
                                                const uint32_t* __restrict__ cand, const int* __restrict__ cand_cnt,
                                                uint4* __restrict__ sel, int* __restrict__ sel_cnt,
                                                int* __restrict__ err, unsigned long long* __restrict__ stamps,
                                                int* __restrict__ ovf, int2* __restrict__ ovf_items,
                                                unsigned long long* __restrict__ ovf_keys, long long ovf_keys_cap,
                                                uint4* __restrict__ osel, int osel_cap, int force_bitonic) {
    // Workgroups go round-robin to the 8 XCDs by linear id, and blockIdx.x has 8 values: with level = blockIdx.x one XCD
    // would get every level-0 workgroup (5x the work of a level-7 one) and set the pace. Rotating by the frame index gives
    // every XCD every level.
    static_assert(kLevels == 8, "level rotation below assumes 8 levels");
    const int l = (int)((blockIdx.x + blockIdx.y) & 7u), frame = blockIdx.y;
#define SSTAMP(k) do { if (stamps && threadIdx.x == 0) stamps[((size_t)blockIdx.y * kLevels + l) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    SSTAMP(0);
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];   // P.sort_cap entries
    __shared__ int s_hist[256];
    __shared__ int s_misc[4];   // [0] threshold score, [1] n1, [2] n2
    __shared__ int s_wsum[NT / 64];
    __shared__ int s_bin[1024];   // keypoints per tile of the level

    const int tid = threadIdx.x;
    const bool lead = tid < 256;          // (always true for NT = 256)
    const LevelGeom g = P.lv[l];
    const uint32_t* clist = cand + (int64_t)frame * P.cand_frame_entries + g.cand_off;
    const int n = min(cand_cnt[frame * kLevels + l], g.cand_cap);
    const int q = g.quota;
    const int kSortCap = P.sort_cap;

    if (q == 0 || n == 0) {   // retainBest(keypoints, 0) clears
        if (tid == 0) sel_cnt[frame * kLevels + l] = 0;
        return;
    }
    if (lead) s_hist[tid] = 0;
    if (tid == 0) { s_misc[1] = 0; s_misc[2] = 0; }
    __syncthreads();
    // The candidate list is walked twice. Eight independent loads per lane and trip: a load-then-use loop pays one
    // global round trip per 256 candidates, which is what this kernel used to spend most of its time on.
    constexpr int kUnroll = 8;
    for (int i0 = 0; i0 < n; i0 += NT * kUnroll) {
        uint32_t v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const int i = i0 + u * NT + tid;
            v[u] = i < n ? clist[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; u++)
            if (i0 + u * NT + tid < n) atomicAdd(&s_hist[v[u] >> 22], 1);
    }
    __syncthreads();
    SSTAMP(1);
    // keypoint.cpp retainBest(2*quota) on the FAST score: keep every score >= the (2q)-th largest
    // = the largest score s whose suffix count sum_{s' >= s} hist[s'] reaches 2q (0 if there are at most 2q). Thread
    // t owns bin 255 - t, so the suffix count is an inclusive prefix sum over threads: wave scan (DPP-free shuffles)
    // + three wave totals through LDS. (A single lane walking the 256 bins cost every workgroup ~20k cycles.)
    {
        const int lane = tid & 63, wv = tid >> 6;
        int c = s_hist[255 - (tid & 255)];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(c, d);
            if (lane >= d) c += o;
        }
        if (lane == 63) s_wsum[wv] = c;
        if (tid == 0) s_misc[0] = 0;
        __syncthreads();
        int before = 0;
#pragma unroll
        for (int w = 0; w < 3; w++)
            if (w < wv) before += s_wsum[w];
        const int cum = c + before;                                     // sum of hist[255 - tid ..255]
        const int own = s_hist[255 - (tid & 255)];
        // exactly one thread sees the count cross 2q (if it crosses at all)
        if (lead && n > 2 * q && cum >= 2 * q && cum - own < 2 * q) s_misc[0] = 255 - tid;
        __syncthreads();
    }
    const int thr = s_misc[0];
    // survivors of the cut are first compacted into the key array (record in the low word) ...
    for (int i0 = 0; i0 < n; i0 += NT * kUnroll) {
        uint32_t v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const int i = i0 + u * NT + tid;
            v[u] = i < n ? clist[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const bool keep = i0 + u * NT + tid < n && (int)(v[u] >> 22) >= thr;
            const unsigned long long m = __ballot(keep);                // one LDS atomic per wave, not per survivor
            if (m) {
                const int lane = tid & 63, leader = __ffsll((long long)m) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(&s_misc[1], __popcll(m));
                base = __shfl(base, leader);
                const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
                if (keep && slot < kSortCap) s_keys[slot] = (unsigned long long)v[u];
            }
        }
    }
    __syncthreads();
    SSTAMP(2);
    // ... then every lane scores its share of them: balanced, and the 27 window loads of a Harris response are the
    // only latency left in the loop
    int pitch;
    const uint8_t* img = raw_level_ptr(P, S, raw, frame, l, pitch);
    {
        const int ns = min(s_misc[1], kSortCap);
        for (int i = tid; i < ns; i += NT) {
            const uint32_t cd = (uint32_t)s_keys[i];
            const int x = cd & 0x7FF, y = (cd >> 11) & 0x7FF;
            const float r = harris_response(img, pitch, x, y, (l > 0) || S.aligned4);
            uint32_t u = __float_as_uint(r);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending-order integer image of the float
            s_keys[i] = ((unsigned long long)(~u) << 32) | ((uint32_t)y << 16) | (uint32_t)x;
        }
    }
    __syncthreads();
    SSTAMP(3);
    int n1 = s_misc[1];
    if (n1 > kSortCap) {
        // more candidates tie at the FAST cut than the LDS sort holds (a tie storm: checkerboards, synthetic patterns).
        // OpenCV's retainBest keeps them all, so this (frame, level) is redone by k_select_ovf in global memory.
        if (INLINE_OVF) {
            __shared__ __attribute__((aligned(8))) int s_ovm[8];
            if (!lead) return;       // select_ovf_item is written for 256 threads; finished waves do not count at barriers
            select_ovf_item(P, S, raw, cand, cand_cnt, sel, sel_cnt, err, ovf, ovf_keys, ovf_keys_cap, osel, osel_cap, frame, l,
                            s_hist, s_ovm);
            return;
        }
        if (tid == 0) {
            sel_cnt[frame * kLevels + l] = 0;
            const int it = atomicAdd(&ovf[0], 1);
            if (it < kOvfItems) ovf_items[it] = make_int2(frame, l);
            else atomicOr(err, ERRBIT_SORT_OVERFLOW);
        }
        return;
    }
    // ---- order the keys: ascending on (~harris_order, y, x) == Harris descending, then y, then x ----
    // Only the best q (+ ties) by Harris are kept, but retainBest(2q) on the integer FAST score hands over 2q plus a
    // whole score bin of ties (often > 1024 keys at level 0), and a bitonic sort in LDS is a chain of ~50 dependent
    // round trips + barriers whatever the size (27k cycles at level 0, 16k at level 7: half of this kernel). Instead:
    //  1. 1024-bin histogram of the keys' Harris order bits (sign + exponent + 2 mantissa bits) and its prefix sum;
    //  2. the bin that holds the q-th key closes the kept set -- a superset of "first q + ties with the q-th";
    //  3. the kept keys are scattered to their bins (upper half of the key array), which orders them coarsely;
    //  4. every key finds its exact place by counting the smaller keys of its own bin (a few dozen at most).
    // Bins too full for that (> 512 keys of near-equal response) or a kept set beyond half the array: bitonic sort.
    __shared__ int s_srt[2];   // [0] bin of the cut, [1] largest kept bin
    auto hbin = [](unsigned long long kk) { return (int)min((uint32_t)(kk >> 53), 1023u); };
    const bool cut = n1 > q + 64;
    for (int i = tid; i < 1024; i += NT) s_bin[i] = 0;
    if (tid == 0) { s_srt[0] = 1023; s_srt[1] = 0; }
    __syncthreads();
    for (int i = tid; i < n1; i += NT) atomicAdd(&s_bin[hbin(s_keys[i])], 1);
    __syncthreads();
    {   // thread t owns bins 4t .. 4t+3: exclusive prefix sum (= first slot of every bin), bin where the count crosses q
        const int lane = tid & 63, wv = tid >> 6;
        int b[4], tot = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { b[k] = lead ? s_bin[4 * tid + k] : 0; tot += b[k]; }
        int c = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(c, d);
            if (lane >= d) c += o;
        }
        if (lane == 63) s_wsum[wv] = c;
        if (tid == 0) s_misc[1] = 0;          // (every wave has read n1 from it: two barriers ago)
        __syncthreads();
        int run = c - tot;
#pragma unroll
        for (int w = 0; w < 3; w++)
            if (w < wv) run += s_wsum[w];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (lead) s_bin[4 * tid + k] = run;
            if (lead && cut && run < q && run + b[k] >= q) s_srt[0] = 4 * tid + k;
            run += b[k];
        }
        __syncthreads();
        const int bc = s_srt[0];
        int mx = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (lead && 4 * tid + k <= bc) mx = max(mx, b[k]);
        if (mx > 0) atomicMax(&s_srt[1], mx);
    }
    const int bcut = s_srt[0];
    if (cut) {
        // in-place compaction of the kept keys to the front, 256 keys per round: a round's keys are in registers before
        // anyone writes (barrier), and writes only reach slots that this or an earlier round has consumed
        for (int i0 = 0; i0 < n1; i0 += NT) {
            const int i = i0 + tid;
            const unsigned long long kk = i < n1 ? s_keys[i] : ~0ull;
            const bool keep = i < n1 && hbin(kk) <= bcut;
            __syncthreads();
            const unsigned long long m = __ballot(keep);
            if (m) {
                const int lane = tid & 63, leader = __ffsll((long long)m) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(&s_misc[1], __popcll(m));
                base = __shfl(base, leader);
                if (keep) s_keys[base + __popcll(m & ((1ull << lane) - 1ull))] = kk;
            }
        }
        __syncthreads();
        n1 = s_misc[1];
    } else {
        __syncthreads();
    }
    const int upper = kSortCap >> 1;
    if (n1 <= upper && s_srt[1] <= 512 && !force_bitonic) {
        unsigned long long* s_up = s_keys + upper;
        for (int i = tid; i < n1; i += NT) {
            const unsigned long long kk = s_keys[i];
            s_up[atomicAdd(&s_bin[hbin(kk)], 1)] = kk;      // s_bin[b]: first slot -> one past the last slot of bin b
        }
        __syncthreads();
        for (int i = tid; i < n1; i += NT) {
            const unsigned long long kk = s_up[i];
            const int b = hbin(kk);
            const int lo = b ? s_bin[b - 1] : 0, hi = s_bin[b];
            int r = lo;
            for (int j0 = lo; j0 < hi; j0 += 8) {       // eight independent LDS reads per trip (the reads past the bin's end
                unsigned long long o[8];                 // stay inside the key array and are not counted)
#pragma unroll
                for (int u = 0; u < 8; u++) o[u] = s_up[min(j0 + u, hi - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++) r += (j0 + u < hi && o[u] < kk) ? 1 : 0;
            }
            s_keys[r] = kk;
        }
    } else {
        int np = 1;
        while (np < n1) np <<= 1;
        for (int i = n1 + tid; i < np; i += NT) s_keys[i] = ~0ull;
        __syncthreads();
        // bitonic sort, ascending on (~harris_order, y, x) == Harris descending, then y, then x
        // Bitonic sort. Steps are taken two at a time: a thread loads the four keys (i, i+h, i+2h, i+3h) of a radix-4
        // butterfly, does the compare-exchanges of step j = 2h and of step h in registers and stores them back -- one LDS
        // round trip per two steps. (When a merge phase k has an odd number of steps its first one, j = k/2, goes alone, two
        // pairs per thread, between workgroup barriers.) In butterfly steps a wave owns one aligned block of 256 keys; while
        // consecutive butterflies stay inside it (span 2j <= 256) no workgroup barrier is needed -- the wave's LDS
        // accesses are served in order.
        auto steps_of = [](int kk) { int n = 0; for (int jj = kk >> 1; jj > 0; jj >>= 1) n++; return n; };
        for (int k = 2; k <= np; k <<= 1) {
            int j = k >> 1;
            if (steps_of(k) & 1) {
                __syncthreads();
                for (int t0 = tid; t0 < (np >> 1); t0 += 2 * NT) {
                    const int ta = t0, tb = t0 + NT;
                    const bool hb = tb < (np >> 1);
                    const int a0 = ((ta & ~(j - 1)) << 1) | (ta & (j - 1)), a1 = a0 | j;
                    const int b0 = hb ? (((tb & ~(j - 1)) << 1) | (tb & (j - 1))) : a0, b1 = hb ? (b0 | j) : a1;
                    const unsigned long long xa = s_keys[a0], ya = s_keys[a1], xb = s_keys[b0], yb = s_keys[b1];
                    if ((xa > ya) == ((a0 & k) == 0)) { s_keys[a0] = ya; s_keys[a1] = xa; }
                    if (hb && (xb > yb) == ((b0 & k) == 0)) { s_keys[b0] = yb; s_keys[b1] = xb; }
                }
                __syncthreads();
                j >>= 1;
            }
            for (; j > 1; j >>= 2) {
                const int h = j >> 1;
                for (int t = tid; t < (np >> 2); t += NT) {
                    const int i = ((t & ~(h - 1)) << 2) | (t & (h - 1));
                    unsigned long long e0 = s_keys[i], e1 = s_keys[i + h], e2 = s_keys[i + j], e3 = s_keys[i + j + h];
                    const bool up = (i & k) == 0;
                    unsigned long long tmp;
                    if ((e0 > e2) == up) { tmp = e0; e0 = e2; e2 = tmp; }       // step j
                    if ((e1 > e3) == up) { tmp = e1; e1 = e3; e3 = tmp; }
                    if ((e0 > e1) == up) { tmp = e0; e0 = e1; e1 = tmp; }       // step h
                    if ((e2 > e3) == up) { tmp = e2; e2 = e3; e3 = tmp; }
                    s_keys[i] = e0; s_keys[i + h] = e1; s_keys[i + j] = e2; s_keys[i + j + h] = e3;
                }
                // what follows: a butterfly with top step j/4 of this phase, or the next phase (a butterfly with top step k
                // if it has an even number of steps; a single step syncs for itself)
                const int next_span = (j >> 2) > 1 ? 2 * (j >> 2) : ((k < np && !(steps_of(2 * k) & 1)) ? 2 * k : 0);
                if (2 * j > 256 || next_span > 256 || np > 4 * NT) __syncthreads();   // np > 4 NT: a thread makes several trips
                else { __threadfence_block(); __builtin_amdgcn_wave_barrier(); }
            }
        }
    }
    __syncthreads();
    SSTAMP(4);
    // retainBest(quota) on Harris: first q plus everything tying with the q-th
    int n2 = n1;
    if (n1 > q) {
        const uint32_t cut = (uint32_t)(s_keys[q - 1] >> 32);
        for (int i = q + tid; i < n1; i += NT)
            if ((uint32_t)(s_keys[i] >> 32) == cut) atomicAdd(&s_misc[2], 1);
        __syncthreads();
        n2 = q + s_misc[2];
    }
    if (n2 > g.sel_cap) {      // more ties at the Harris cut than the level's slots hold: same fallback
        if (INLINE_OVF) {
            __shared__ __attribute__((aligned(8))) int s_ovm[8];
            if (!lead) return;       // select_ovf_item is written for 256 threads; finished waves do not count at barriers
            select_ovf_item(P, S, raw, cand, cand_cnt, sel, sel_cnt, err, ovf, ovf_keys, ovf_keys_cap, osel, osel_cap, frame, l,
                            s_hist, s_ovm);
            return;
        }
        if (tid == 0) {
            sel_cnt[frame * kLevels + l] = 0;
            const int it = atomicAdd(&ovf[0], 1);
            if (it < kOvfItems) ovf_items[it] = make_int2(frame, l);
            else atomicOr(err, ERRBIT_SEL_OVERFLOW);
        }
        return;
    }
    // Emit in raster order of 32 x 32-px tiles (counting sort, order inside a tile arbitrary), each record carrying its rank i in
    // the canonical order: k_describe then works on spatially close keypoints at the same time -- their 37-row
    // windows overlap, so the 128-byte lines they need are fetched from L2 once instead of once per keypoint --
    // and writes keypoint i to row i of the output whatever order it was processed in.
    uint4* out = sel + (int64_t)frame * P.sel_frame_entries + g.sel_off;
    // 32-px tiles while the level has at most 1024 of them, else 64-px tiles
    const int ts = (((g.w + 31) >> 5) * ((g.h + 31) >> 5) <= 1024) ? 5 : 6;
    const int tiles_x = (g.w + (1 << ts) - 1) >> ts, n_bins = tiles_x * ((g.h + (1 << ts) - 1) >> ts);
    for (int i = tid; i < n_bins; i += NT) s_bin[i] = 0;
    __syncthreads();
    for (int i = tid; i < n2; i += NT) {
        const uint32_t xy = (uint32_t)s_keys[i];
        atomicAdd(&s_bin[((xy >> (16 + ts)) * tiles_x) + ((xy & 0xFFFFu) >> ts)], 1);
    }
    __syncthreads();
    {   // exclusive prefix sum of the bins: thread t owns bins 4t .. 4t+3
        const int lane = tid & 63, wv = tid >> 6;
        int b[4], tot = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { b[k] = (lead && 4 * tid + k < n_bins) ? s_bin[4 * tid + k] : 0; tot += b[k]; }
        int c = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(c, d);
            if (lane >= d) c += o;
        }
        __syncthreads();                      // s_wsum was read by the threshold scan above
        if (lane == 63) s_wsum[wv] = c;
        __syncthreads();
        int run = c - tot;                    // exclusive within the wave
#pragma unroll
        for (int w = 0; w < 3; w++)
            if (w < wv) run += s_wsum[w];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (lead && 4 * tid + k < n_bins) s_bin[4 * tid + k] = run;
            run += b[k];
        }
    }
    __syncthreads();
    for (int i = tid; i < n2; i += NT) {
        const unsigned long long kk = s_keys[i];
        uint32_t u = ~(uint32_t)(kk >> 32);
        u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
        const uint32_t xy = (uint32_t)kk;
        const int pos = atomicAdd(&s_bin[((xy >> (16 + ts)) * tiles_x) + ((xy & 0xFFFFu) >> ts)], 1);
        out[pos] = make_uint4(xy, u, (uint32_t)i, 0u);   // (x | y << 16, harris bits, canonical rank)
    }
    if (tid == 0) sel_cnt[frame * kLevels + l] = n2;
    SSTAMP(5);
#undef SSTAMP
}

// ------------------------------------------------------------------------------------------------------
// a6.3-a6.5 for the (frame, level) pairs whose ties did not fit k_select's LDS capacities: the same selection --
// retainBest(2q) on the FAST score, Harris, retainBest(q) on Harris, ties kept at both cuts (keypoint.cpp
// KeyPointsFilter::retainBest) -- with the keys in a global-memory arena: histogram cut, Harris keys, a bitonic sort
// by one workgroup in global memory, then the selected keypoints go to the level's regular slots (first sel_cap) and to
// the overflow arena `osel` (the rest, in blocks of four entries of one (frame, level): k_describe's arena pass).
// Slow (milliseconds per item) and rare by construction; what matters is that the result equals the reference's.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long ovf_ld(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // bypasses this CU's L1: other waves' stores
}
__device__ __forceinline__ void ovf_st(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One (frame, level) through the global-memory selection (all 256 threads of the calling workgroup, uniformly).
__device__ __attribute__((noinline)) void select_ovf_item(const Plan& P, const FrameSrc& S, const uint8_t* __restrict__ raw,
                                const uint32_t* __restrict__ cand, const int* __restrict__ cand_cnt,
                                uint4* __restrict__ sel, int* __restrict__ sel_cnt, int* __restrict__ err,
                                int* __restrict__ ovf, unsigned long long* __restrict__ keys, long long keys_cap,
                                uint4* __restrict__ osel, int osel_cap, int frame, int l, int* s_hist, int* s_misc) {
    const int tid = threadIdx.x;
    {
        const LevelGeom g = P.lv[l];
        const uint32_t* clist = cand + (int64_t)frame * P.cand_frame_entries + g.cand_off;
        const int n = min(cand_cnt[frame * kLevels + l], g.cand_cap);
        const int q = g.quota;
        __syncthreads();
        s_hist[tid] = 0;
        if (tid < 8) s_misc[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 256) atomicAdd(&s_hist[clist[i] >> 22], 1);
        __syncthreads();
        if (tid == 0) {   // retainBest(2q): largest score whose suffix count reaches 2q (0 if there are at most 2q)
            int thr = 0, n1 = n;
            if (n > 2 * q) {
                int cum = 0;
                for (int sc = 255; sc >= 0; sc--) {
                    cum += s_hist[sc];
                    if (cum >= 2 * q) { thr = sc; n1 = cum; break; }
                }
            }
            long long np = 1;
            while (np < n1) np <<= 1;
            const long long base = (long long)atomicAdd(reinterpret_cast<unsigned long long*>(ovf + 2), (unsigned long long)np);
            s_misc[0] = thr;
            if (base + np > keys_cap) { atomicOr(err, ERRBIT_SORT_OVERFLOW); s_misc[3] = -1; }
            else { s_misc[3] = 0; }
            reinterpret_cast<long long*>(s_misc + 6)[0] = base;
            s_misc[5] = n1;
        }
        __syncthreads();
        if (s_misc[3] < 0) { if (tid == 0) sel_cnt[frame * kLevels + l] = 0; return; }
        const int thr = s_misc[0], n1 = s_misc[5];
        unsigned long long* K = keys + reinterpret_cast<long long*>(s_misc + 6)[0];
        long long np = 1;
        while (np < n1) np <<= 1;
        int pitch;
        const uint8_t* img = raw_level_ptr(P, S, raw, frame, l, pitch);
        for (int i0 = 0; i0 < n; i0 += 256) {
            const int i = i0 + tid;
            const uint32_t cd = i < n ? clist[i] : 0u;
            const bool keep = i < n && (int)(cd >> 22) >= thr;
            unsigned long long key = 0;
            if (keep) {
                const int x = cd & 0x7FF, y = (cd >> 11) & 0x7FF;
                const float r = harris_response(img, pitch, x, y, (l > 0) || S.aligned4);
                uint32_t u = __float_as_uint(r);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                key = ((unsigned long long)(~u) << 32) | ((uint32_t)y << 16) | (uint32_t)x;
            }
            const unsigned long long m = __ballot(keep);
            if (m) {
                const int lane = tid & 63, leader = __ffsll((long long)m) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(&s_misc[1], __popcll(m));
                base = __shfl(base, leader);
                if (keep) ovf_st(&K[base + __popcll(m & ((1ull << lane) - 1ull))], key);
            }
        }
        for (long long i = n1 + tid; i < np; i += 256) ovf_st(&K[i], ~0ull);
        __syncthreads();
        // bitonic sort in global memory, ascending on (~harris_order, y, x) == Harris descending, then y, then x
        for (long long k = 2; k <= np; k <<= 1)
            for (long long j = k >> 1; j > 0; j >>= 1) {
                for (long long t = tid; t < (np >> 1); t += 256) {
                    const long long a = ((t & ~(j - 1)) << 1) | (t & (j - 1)), b = a | j;
                    const unsigned long long xa = ovf_ld(&K[a]), xb = ovf_ld(&K[b]);
                    if ((xa > xb) == ((a & k) == 0)) { ovf_st(&K[a], xb); ovf_st(&K[b], xa); }
                }
                __syncthreads();
            }
        // retainBest(quota) on Harris: first q plus everything tying with the q-th
        int n2 = n1;
        if (n1 > q) {
            const uint32_t cut = (uint32_t)(ovf_ld(&K[q - 1]) >> 32);
            for (int i = q + tid; i < n1; i += 256)
                if ((uint32_t)(ovf_ld(&K[i]) >> 32) == cut) atomicAdd(&s_misc[2], 1);
            __syncthreads();
            n2 = q + s_misc[2];
        }
        const int extra = max(n2 - g.sel_cap, 0), extra4 = (extra + 3) & ~3;
        if (tid == 0) {
            int ob = 0;
            if (extra4 > 0) {
                ob = atomicAdd(&ovf[1], extra4);
                if (ob + extra4 > osel_cap) { atomicOr(err, ERRBIT_SEL_OVERFLOW); ob = -1; }
            }
            s_misc[4] = ob;
        }
        __syncthreads();
        const int ob = s_misc[4];
        if (ob < 0) n2 = g.sel_cap;
        uint4* out = sel + (int64_t)frame * P.sel_frame_entries + g.sel_off;
        const uint32_t tag = (uint32_t)frame | ((uint32_t)l << 24);
        for (int i = tid; i < n2; i += 256) {
            const unsigned long long kk = ovf_ld(&K[i]);
            uint32_t u = ~(uint32_t)(kk >> 32);
            u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
            const uint4 rec = make_uint4((uint32_t)kk, u, (uint32_t)i, tag);
            if (i < g.sel_cap) out[i] = rec;
            else osel[ob + (i - g.sel_cap)] = rec;
        }
        if (ob >= 0)
            for (int i = extra + tid; i < extra4; i += 256) osel[ob + i] = make_uint4(0u, 0u, 0xFFFFFFFFu, tag);   // padding
        if (tid == 0) sel_cnt[frame * kLevels + l] = n2;
    }
}

__global__ __launch_bounds__(256) void k_select_ovf(Plan P, FrameSrc S, const uint8_t* __restrict__ raw,
                                                    const uint32_t* __restrict__ cand, const int* __restrict__ cand_cnt,
                                                    uint4* __restrict__ sel, int* __restrict__ sel_cnt, int* __restrict__ err,
                                                    int* __restrict__ ovf, const int2* __restrict__ ovf_items,
                                                    unsigned long long* __restrict__ keys, long long keys_cap,
                                                    uint4* __restrict__ osel, int osel_cap) {
    __shared__ int s_hist[256];
    __shared__ __attribute__((aligned(8))) int s_misc[8];   // [0] FAST cut, [1] keys written, [2] ties beyond q, [3] key arena ok, [4] osel base, [5] n1, [6..7] key base
    const int n_items = min(ovf[0], kOvfItems);
    for (int it = blockIdx.x; it < n_items; it += gridDim.x)
        select_ovf_item(P, S, raw, cand, cand_cnt, sel, sel_cnt, err, ovf, keys, keys_cap, osel, osel_cap, ovf_items[it].x,
                        ovf_items[it].y, s_hist, s_misc);
}

// ------------------------------------------------------------------------------------------------------
// a6.6 + a6.8  one wave per selected keypoint
// ------------------------------------------------------------------------------------------------------
// mathfuncs_core.dispatch.cpp fastAtan2 (scalar atan_f32)
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// Double-precision sin/cos from IEEE add/mul/floor only (Cody-Waite by pi/2 + fdlibm kernel polynomials):
// the same operation sequence on x86 and gfx950 gives the same bits. Stands in for orb.cpp's
// (float)cos(angle) / (float)sin(angle), whose libm call is platform-defined in its last double ulp.
__device__ __forceinline__ void det_sincos(double x, double& s, double& c) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;
    const double pio2_lo = 6.07710050650619224932e-11;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double fn = floor(x * invpio2 + 0.5);
    const double r = (x - fn * pio2_hi) - fn * pio2_lo;
    const double z = r * r;
    const double sp = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double sn = r + (z * r) * (S1 + z * sp);
    const double cp = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double cs = 1.0 - (0.5 * z - z * cp);
    const int qd = (int)fn & 3;
    s = (qd == 0) ? sn : (qd == 1) ? cs : (qd == 2) ? -sn : -cs;
    c = (qd == 0) ? cs : (qd == 1) ? -sn : (qd == 2) ? -cs : sn;
}

// Full-wave integer sum on the DPP crossbar (no LDS traffic): quad swaps, half-row and row mirrors give every lane its
// 16-lane row total; row_bcast:15 / row_bcast:31 chain the four rows; lane 63 holds the wave total.
__device__ __forceinline__ int wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm:[1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}

// The rotated pattern never leaves [-18, 18]^2 (|x|,|y| <= 13*sqrt(2)), so a 37-row x 48-byte window of the blurred
// level (dword-aligned start) holds every sample of a keypoint.
constexpr int kDescR = 18;
constexpr int kDescRows = 2 * kDescR + 1;   // 37
constexpr int kDescPitch = 64;              // four 16-byte-ALIGNED pieces per row (>= 37 + 15 bytes of alignment slack)
constexpr int kDescWaves = 2;                // waves per workgroup (4 measured slightly slower); consecutive slots are spatially close
constexpr int kDescKp = 4 * kDescWaves;     // keypoints per workgroup: 4 DPP rows per wave
constexpr int kIcPitch = 48;                // raw window: 31 rows x three 16-byte pieces (>= 31 + 3 bytes of alignment slack)
struct __attribute__((packed, aligned(4))) DwordQuad { uint32_t a, b, c, d; };   // 16-byte load from a 4-byte-aligned address

// sum over the aligned 8 lanes of a DPP half row
__device__ __forceinline__ int row8_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm:[1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    return v;
}

// sum over the 16 lanes of a DPP row (every lane of the row gets the total)
__device__ __forceinline__ int row16_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm:[1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
    return v;
}

// a6.6 + a6.8. One keypoint per 16-lane DPP row, four per wave: the per-keypoint scalar work (fastAtan2, the
// deterministic sincos, bookkeeping) is done once per instruction stream for four keypoints instead of once per
// wave, which is where a wave-per-keypoint layout spent half its instructions. Lane (row r, l16):
//   IC moments : columns u = 2*l16-15, 2*l16-14 over the 31 rows of the disc, DPP row reduction
//   rBRIEF     : test 16*it + l16 for it = 0..15; one 64-bit ballot per iteration carries 16 descriptor bits for each
//                of the four keypoints; lane l16 keeps word l16 and stores its two bytes
// MODE 0: the frames' regular slots (batches). MODE 1: only the tie-storm arena (batches: a second, tiny launch). MODE 2:
// both in one launch -- the regular blocks, then kDescArenaBlocks blocks that stride over the arena (single-frame latency
// schedule: one launch less; in a batch the extra code costs the regular blocks registers).
constexpr int kDescArenaBlocks = 32;
// (130 VGPRs = three waves per SIMD. Capping at 128 -- amdgpu_waves_per_eu(4), 120 used, no spills -- makes this stage 7 %
// faster on its own, 0.96 -> 0.89 us/frame, but not the two-stream pipeline: 268.0-269.0k frames/s without the cap,
// 266.3-269.1k with it, and the FAST/blur kernel beside the matcher then measures 0.160 instead of 0.163. Left uncapped.)
// Q4: the blurred levels are in Q4 order (orb_device.h; the batch path with k_fast_blur_stream): the window is 10 x 10
// pieces of 16 bytes (4 rows x 4 px each, 16-byte aligned by construction), 7 loads per lane instead of 10, and lands in a
// 40 x 40 row-major LDS window. Row-major levels (band kernel: single-frame schedule, tiny images): 37 rows x 64 bytes.
constexpr int kDescQ = 10;                   // row quads and dword columns of the Q4 window
constexpr int kDescPitchQ = 4 * kDescQ;      // 40
template <int MODE, bool Q4>
__global__ __launch_bounds__(64 * kDescWaves) void k_describe(Plan P, FrameSrc S, const uint8_t* __restrict__ raw,
                                                  const uint8_t* __restrict__ blur, const uint4* __restrict__ sel,
                                                  const int* __restrict__ sel_cnt, aria_keypoint* __restrict__ kps,
                                                  uint8_t* __restrict__ desc, int* __restrict__ counts, int kp_cap,
                                                  int* __restrict__ err, int n_frames, int blocks_per_frame,
                                                  unsigned long long* __restrict__ stamps,
                                                  const uint4* __restrict__ osel, const int* __restrict__ ovf, int osel_cap) {
    // diagnostic only (ARIA_DESC_STAMPS=1): s_memtime at the phase boundaries of every wave
#define DSTAMP(k) do { if (stamps && (threadIdx.x & 63) == 0) stamps[((size_t)blockIdx.x * kDescWaves + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    DSTAMP(0);
    // One LDS window per keypoint, used twice: first the raw 31 x 48 window (IC moments), then -- once the moments are
    // reduced -- the blurred 37 x 64 window, which has been waiting in registers since both were requested together.
    constexpr int kPatchBytes = Q4 ? kDescPitchQ * kDescPitchQ : kDescRows * kDescPitch;      // >= 31 * kIcPitch either way
    static_assert(kPatchBytes >= 31 * kIcPitch && kPatchBytes % 16 == 0, "the raw window shares the allocation");
    __shared__ __attribute__((aligned(16))) uint8_t s_patch[kDescKp][kPatchBytes];
    uint8_t (*s_raw)[kPatchBytes] = s_patch;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grp = lane >> 4, l16 = lane & 15;
    // regular pass: one call (the grid is the work); arena pass: a small fixed grid strides over the arena's blocks. The body
    // is a lambda so that the regular instantiation stays straight-line code.
    const int n_regular = MODE == 2 ? (int)gridDim.x - kDescArenaBlocks : (int)gridDim.x;
    auto describe_block = [&](const int blk_in, const bool ARENA) {
    const int blk = __builtin_amdgcn_readfirstlane(blk_in);     // block-uniform: keep the frame / level bookkeeping on the scalar unit
    int frame, l = 0, base = 0, nk;
    uint4 sv;
    if (!ARENA) {
        // XCD-aware block -> (frame, slot) map: workgroups are dealt round-robin over the 8 XCDs (speed only, never
        // correctness); give every XCD whole frames so a frame's patches are fetched into one L2.
        const int xcd = blk & 7, j = blk >> 3;
        // (the division expands to vector code even for uniform operands: pin the result back to the scalar unit, or every
        // frame-dependent address below is computed per lane)
        frame = __builtin_amdgcn_readfirstlane((j / blocks_per_frame) * 8 + xcd);
        if (frame >= n_frames) return;
        const int jm = __builtin_amdgcn_readfirstlane(j % blocks_per_frame);
        const int slot0 = __builtin_amdgcn_readfirstlane(jm * kDescKp + wv * 4);      // first of this wave's four slots (one level)
        const int* cnt = sel_cnt + frame * kLevels;
        int total = 0;
#pragma unroll
        for (int i = 0; i < kLevels; i++) {
            const int ci = cnt[i];
            if (i > 0 && slot0 >= P.lv[i].sel_off) { l = i; base = total; }
            total += ci;
        }
        // counts stay within kp_cap (the matcher trusts them); a frame that needs more rows raises ERRBIT_KPCAP below and
        // leaves the number it needs in err[2]
        if (slot0 == 0 && lane == 0) {
            counts[frame] = min(total, kp_cap);
            if (total > kp_cap) atomicMax(err + 2, total);
        }
        if (slot0 >= P.sel_frame_entries) return;
        l = __builtin_amdgcn_readfirstlane(l);     // the wave's four slots are in one level: keep the level geometry scalar
        base = __builtin_amdgcn_readfirstlane(base);
        const int i0 = slot0 - P.lv[l].sel_off;
        nk = min(4, min(cnt[l], P.lv[l].sel_cap) - i0);                    // valid keypoints of this wave
        if (nk <= 0) return;
        // slots hold the level's keypoints in tile order; the record says which row of the output it is
        sv = sel[(int64_t)frame * P.sel_frame_entries + slot0 + min(grp, nk - 1)];
    } else {
        // arena pass: keypoints of tie-storm levels beyond the level's regular slots (k_select_ovf), in blocks of four
        // entries of one (frame, level); .w = frame | level << 24, rank 0xFFFFFFFF = padding (at the end of a block)
        const int e0 = blk * kDescKp + wv * 4;
        if (e0 >= min(ovf[1], osel_cap)) return;
        const uint4 mine = osel[e0 + grp];
        const unsigned long long vm = __ballot(mine.z != 0xFFFFFFFFu && l16 == 0);
        nk = __popcll(vm);
        if (nk <= 0) return;
        const uint32_t tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)mine.w);
        frame = (int)(tag & 0xFFFFFFu);
        l = (int)(tag >> 24);
        const int* cnt = sel_cnt + frame * kLevels;
        for (int i = 0; i < l; i++) base += cnt[i];
        sv = osel[e0 + min(grp, nk - 1)];
    }
    const LevelGeom g = P.lv[l];
    const bool valid = grp < nk;
    const int oidx = base + (int)sv.z;
    const bool fits = oidx < kp_cap;
    if (valid && !fits && l16 == 0) atomicOr(err, ERRBIT_KPCAP);
    const int x = sv.x & 0xFFFF, y = sv.x >> 16;
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    DSTAMP(1);

    // pattern rows and disc masks of this lane (independent of the keypoint): issue early
    uint32_t icm[16];
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
        const uint4 m4 = *reinterpret_cast<const uint4*>(&kIcMask.m[l16][k]);
        icm[k] = m4.x; icm[k + 1] = m4.y; icm[k + 2] = m4.z; icm[k + 3] = m4.w;
    }
    int pat[16];
#pragma unroll
    for (int it = 0; it < 16; it += 4) {
        const uint4 p4 = *reinterpret_cast<const uint4*>(&kPattern31.w[l16][it]);
        pat[it] = (int)p4.x; pat[it + 1] = (int)p4.y; pat[it + 2] = (int)p4.z; pat[it + 3] = (int)p4.w;
    }

    // ---- request the blurred 37 x 64 and the raw 31 x 48 window of each keypoint with 16-byte loads. What bounds this
    //      kernel is the L1 (TCP) access rate: one 64-byte access per clock per CU, and a 16-byte lane load that is not
    //      16-byte aligned, or whose neighbours in the same 64-byte chunk sit in other instructions, costs an access of
    //      its own (245 accesses per keypoint = one per lane load, measured). So: pieces are 16-byte ALIGNED (window
    //      origin rounded down to 16), and the lanes of an instruction cover whole rows, piece by piece, so that the
    //      pieces of one 64-byte chunk are adjacent lanes and coalesce. No lane is ever masked: lanes past the last
    //      row repeat a (row, piece) that another lane or step also holds. The windows end <= 45 px right of a
    //      keypoint that is >= 31 px inside the level, on rows >= 16 above the last: never past the image. ----
    const uint8_t* bl = blur + (int64_t)frame * P.blur_frame_bytes + g.blur_off;
    const int xs = Q4 ? ((x - kDescR) & ~3) : ((x - kDescR) & ~15);     // first window column: dword / 16-byte aligned (pitch and bases are too)
    const int ys = Q4 ? ((y - kDescR) & ~3) : (y - kDescR);            // first window row
    constexpr int kPv = Q4 ? 7 : 10;
    DwordQuad pv[kPv];                                     // blurred window, parked in registers until the moments are done
    int pq[kPv], pc[kPv];                                  // Q4: (row quad, dword column) of the lane's pieces; 100 pieces, 112 lane slots: the last ones repeat piece 99
    if constexpr (Q4) {
        const int64_t qp = (int64_t)g.pitch * 4;
        const uint8_t* gp = bl + (int64_t)(ys >> 2) * qp + (xs >> 2) * 16;
#pragma unroll
        for (int k = 0; k < kPv; k++) {
            const int p = min(16 * k + l16, kDescQ * kDescQ - 1);
            pq[k] = (p * 205) >> 11;                        // p / 10 for p < 1024
            pc[k] = p - kDescQ * pq[k];
            pv[k] = *reinterpret_cast<const DwordQuad*>(gp + pq[k] * qp + pc[k] * 16);
        }
    } else {
        const int r4 = l16 >> 2, c4 = l16 & 3;             // lane -> (row mod 4, piece): four whole rows per instruction
        const int64_t p64 = g.pitch;
        const uint8_t* gp = bl + (int64_t)(y - kDescR) * p64 + xs + 16 * c4;
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const int row = (4 * k + 4 <= kDescRows) ? 4 * k + r4 : min(4 * k + r4, kDescRows - 1);
            pv[k] = *reinterpret_cast<const DwordQuad*>(gp + row * p64);
        }
        (void)pq; (void)pc;
    }
    const int rr = l16 / 3, cc = l16 - 3 * rr;             // raw window: lane -> (row mod 5, piece), five rows per instruction
    int pitch;
    const uint8_t* img = raw_level_ptr(P, S, raw, frame, l, pitch);
    // a caller's level-0 image may be only 4- or 1-byte aligned (the level is the same for the wave's four keypoints)
    const bool raw_dword = __builtin_amdgcn_readfirstlane((l > 0) || S.aligned4) != 0;
    const bool raw_a16 = __builtin_amdgcn_readfirstlane((l > 0) || S.aligned16) != 0;
    const int xr = raw_a16 ? ((x - kHalfPatch) & ~15) : ((x - kHalfPatch) & ~3);   // x - 15 - xr in 0..15: 31 + 15 < 48
    {
        uint8_t* sr = s_raw[wv * 4 + grp] + 16 * cc;
        const int64_t p64 = pitch;
        const uint8_t* gp = img + (int64_t)(y - kHalfPatch) * p64 + xr + 16 * cc;
        DwordQuad rv[7];
        if (raw_dword) {
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const int row = (5 * k + 5 < 31) ? 5 * k + rr : min(5 * k + rr, 30);
                rv[k] = *reinterpret_cast<const DwordQuad*>(gp + row * p64);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const int row = (5 * k + 5 < 31) ? 5 * k + rr : min(5 * k + rr, 30);
                const uint8_t* q = gp + row * p64;
                rv[k].a = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
                rv[k].b = (uint32_t)q[4] | ((uint32_t)q[5] << 8) | ((uint32_t)q[6] << 16) | ((uint32_t)q[7] << 24);
                rv[k].c = (uint32_t)q[8] | ((uint32_t)q[9] << 8) | ((uint32_t)q[10] << 16) | ((uint32_t)q[11] << 24);
                rv[k].d = (uint32_t)q[12] | ((uint32_t)q[13] << 8) | ((uint32_t)q[14] << 16) | ((uint32_t)q[15] << 24);
            }
        }
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const int row = (5 * k + 5 < 31) ? 5 * k + rr : min(5 * k + rr, 30);
            *reinterpret_cast<uint4*>(sr + row * kIcPitch) = make_uint4(rv[k].a, rv[k].b, rv[k].c, rv[k].d);
        }
    }
    if (stamps) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
    DSTAMP(2);
    __builtin_amdgcn_wave_barrier();

    // ---- orb.cpp ICAngles: m10 = sum u*I, m01 = sum v*I over the radius-15 disc (integer, order-free) ----
    // Dword-wise: lane (hp = l16 >> 3, j = l16 & 7) owns patch columns 4j-15 .. 4j-12 of the rows hp, hp+2, ...; a
    // step realigns the window dword (64-bit shift by the keypoint's sub-dword offset), masks it to the disc and feeds
    // three accumulators: sum (u+16) I and sum (v+16-hp) I by v_dot4_u32_u8 (unsigned weights), sum I by v_sad_u8.
    const int sh = (x - xr) - kHalfPatch;                    // 0..15: window byte of patch column -15
    const int hp = l16 >> 3, jj = l16 & 7;
    const uint8_t* rbase = s_raw[wv * 4 + grp] + 4 * jj + (sh & ~3);
    const uint32_t wu = 0x04030201u + 0x04040404u * (uint32_t)jj;      // (u + 16) for the lane's four columns
    uint32_t acc_u = 0, acc_v = 0, acc_s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int r = min(2 * k + hp, 30);                   // k == 15, hp == 1: padding step, its mask is 0
        const uint32_t lo = *reinterpret_cast<const uint32_t*>(rbase + r * kIcPitch);
        const uint32_t hi = *reinterpret_cast<const uint32_t*>(rbase + r * kIcPitch + 4);
        const uint32_t pix = __builtin_amdgcn_alignbyte(hi, lo, sh) & icm[k];
        acc_u = __builtin_amdgcn_udot4(pix, wu, acc_u, false);
        acc_v = __builtin_amdgcn_udot4(pix, 0x01010101u * (uint32_t)(2 * k + 1), acc_v, false);   // v + 16 - hp = 2k + 1
        acc_s = __builtin_amdgcn_sad_u8(pix, 0u, acc_s);
    }
    const int sum_u = row16_sum((int)acc_u), sum_v = row16_sum((int)acc_v);
    // rows of one parity live in one half of the DPP row: sum I per half for the hp term, total for the -16 terms
    const int s_half = row8_sum((int)acc_s);                  // sum I over this lane's 8-lane half (one row parity)
    const int s_othr = __builtin_amdgcn_update_dpp(0, s_half, 0x140, 0xF, 0xF, true);   // row_mirror: the other half's sum
    const int s_all = s_half + s_othr;
    const int s_hp1 = hp ? s_half : s_othr;                   // sum I over the odd rows (hp == 1)
    const int m10 = sum_u - 16 * s_all;
    const int m01 = sum_v + s_hp1 - 16 * s_all;
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    DSTAMP(3);

    // ---- orb.cpp computeOrbDescriptors (WTA_K 2) on the blurred level, samples served from the LDS window ----
    float ang = angle;
    ang *= (float)(3.1415926535897932384626433832795 / 180.f);
    double sd, cd;
    det_sincos((double)ang, sd, cd);
    const float a = (float)cd, b = (float)sd;
    __builtin_amdgcn_wave_barrier();                       // every lane of the wave is done reading the raw window
    constexpr int kPitchL = Q4 ? kDescPitchQ : kDescPitch;
    if constexpr (Q4) {
        uint8_t* sp = s_patch[wv * 4 + grp];
#pragma unroll
        for (int k = 0; k < kPv; k++) {
            uint32_t* d = reinterpret_cast<uint32_t*>(sp + (4 * pq[k]) * kPitchL + 4 * pc[k]);    // piece = 4 rows of one dword column
            d[0] = pv[k].a; d[kPitchL / 4] = pv[k].b; d[2 * kPitchL / 4] = pv[k].c; d[3 * kPitchL / 4] = pv[k].d;
        }
    } else {
        const int r4 = l16 >> 2, c4 = l16 & 3;
        uint8_t* sp = s_patch[wv * 4 + grp] + 16 * c4;
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const int row = (4 * k + 4 <= kDescRows) ? 4 * k + r4 : min(4 * k + r4, kDescRows - 1);
            *reinterpret_cast<uint4*>(sp + row * kDescPitch) = make_uint4(pv[k].a, pv[k].b, pv[k].c, pv[k].d);
        }
    }
    const uint8_t* bc = s_patch[wv * 4 + grp] + (y - ys) * kPitchL + (x - xs);     // window address of the keypoint centre
    __builtin_amdgcn_wave_barrier();
    // Both samples of a test ride in one register pair: (fx0, fx1) = (px0, px1) * a + (py0, py1) * (-b) and
    // (fy0, fy1) = (px0, px1) * b + (py0, py1) * a -- packed multiplies and adds, each rounded once like the
    // reference's px*a - py*b / px*b + py*a (no FMA). cvRound is round-half-even = adding 1.5 * 2^23 in
    // the default rounding mode: the integer lands in the low mantissa bits, so the window address is one 24-bit mad
    // of the raw float bits plus a constant.
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 aa = {a, a}, bb = {b, b}, nbb = {-b, -b};
    const f2 magic = {12582912.0f, 12582912.0f};
    constexpr uint32_t kMagicBits = 0x4B400000u;          // bits of 1.5 * 2^23
    // mad_u24(by, pitch, bx) = ((by & 0xFFFFFF) * pitch) + bx = (0x400000 + iy) * pitch + kMagicBits + ix
    const uint8_t* bc0 = bc - (size_t)(0x400000u * (uint32_t)kPitchL + kMagicBits);
    DSTAMP(4);
    uint32_t mine = 0;
#pragma unroll
    for (int it = 0; it < 16; it++) {
        const float px0 = (float)(signed char)(pat[it] & 0xFF), py0 = (float)(signed char)((pat[it] >> 8) & 0xFF);
        const float px1 = (float)(signed char)((pat[it] >> 16) & 0xFF), py1 = (float)(signed char)((pat[it] >> 24) & 0xFF);
        const f2 PX = {px0, px1}, PY = {py0, py1};
        const f2 fx = PX * aa + PY * nbb + magic;
        const f2 fy = PX * bb + PY * aa + magic;
        const uint32_t o0 = __umul24(__float_as_uint(fy.x), kPitchL) + __float_as_uint(fx.x);
        const uint32_t o1 = __umul24(__float_as_uint(fy.y), kPitchL) + __float_as_uint(fx.y);
        const int t0 = bc0[o0];
        const int t1 = bc0[o1];
        const unsigned long long m = __ballot(t0 < t1);
        // lane l16 == it of every group keeps its group's 16 bits; the lane mask is a compile-time constant handed to
        // v_cndmask as it is (no per-iteration v_cmp)
        if (__builtin_amdgcn_inverse_ballot_w64(0x0001000100010001ull << it)) mine = (uint32_t)(m >> (16 * grp)) & 0xFFFFu;
    }
    DSTAMP(5);
    if (valid && fits) {
        const int64_t orow = (int64_t)frame * kp_cap + oidx;
        *reinterpret_cast<uint16_t*>(desc + orow * 32 + 2 * l16) = (uint16_t)mine;
        if (l16 == 0) {
            aria_keypoint k;
            k.x = (float)x * g.scale;          // orb.cpp computeKeyPoints: pt *= layerScale[octave]
            k.y = (float)y * g.scale;
            k.size = kPatchSize * g.scale;     // size = patchSize * sf
            k.angle = angle;
            k.response = __uint_as_float(sv.y);
            k.octave = l;
            kps[orow] = k;
        }
    }
    };
    if constexpr (MODE == 0) {
        describe_block((int)blockIdx.x, false);
    } else if constexpr (MODE == 1) {
        const int n_blk = (min(ovf[1], osel_cap) + kDescKp - 1) / kDescKp;
        for (int blk = blockIdx.x; blk < n_blk; blk += gridDim.x) describe_block(blk, true);
    } else {
        if ((int)blockIdx.x < n_regular) {
            describe_block((int)blockIdx.x, false);
        } else {
            const int n_blk = (min(ovf[1], osel_cap) + kDescKp - 1) / kDescKp;
            for (int blk = (int)blockIdx.x - n_regular; blk < n_blk; blk += kDescArenaBlocks) describe_block(blk, true);
        }
    }
    DSTAMP(6);
#undef DSTAMP
}

// ------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------
static int env_int(const char* name, int dflt) { const char* e = aria_getenv(name); return e ? atoi(e) : dflt; }
static bool env_is(const char* name, char c) { const char* e = aria_getenv(name); return e && e[0] == c; }

// The environment is read once per process (thread-safe static initialisation) and never written again.
const EnvConfig& env_config() {
    static const EnvConfig cfg = [] {
        EnvConfig c{};
        c.ablate = env_int("ARIA_ABLATE", 0);
        c.level_streams = env_is("ARIA_LEVEL_STREAMS", '1') ? 1 : 0;
        c.stamp_level = env_int("ARIA_STAMPS", -1);
        c.sel_stamps = env_is("ARIA_SEL_STAMPS", '1') ? 1 : 0;
        c.select_bitonic = env_is("ARIA_SELECT_SORT", 'b') ? 1 : 0;
        c.band_xcd_map = env_is("ARIA_BAND_XCD_MAP", '0') ? 0 : 1;
        c.band_trim = env_is("ARIA_BAND_TRIM", '0') ? 0 : 1;
        c.desc_stamps = env_is("ARIA_DESC_STAMPS", '1') ? 1 : 0;
        c.fast_blur_impl = env_is("ARIA_FAST_BLUR_IMPL", 't') ? 0 : env_is("ARIA_FAST_BLUR_IMPL", 'm') ? 1 : 2;
        c.pyr_impl = env_is("ARIA_PYRAMID_IMPL", 'f') ? 1 : 0;
        c.rs_impl = env_is("ARIA_RESIZE_IMPL", 'd') ? 0 : env_is("ARIA_RESIZE_IMPL", 'l') ? 1 : 2;
        // A separate resize pass runs only when asked for (ARIA_RESIZE_FUSE=0, ARIA_RESIZE_IMPL, ARIA_PYRAMID_IMPL) or when
        // the band kernel is not the one in use (tile kernel, per-level side streams).
        c.fuse_resize = (c.fast_blur_impl != 0 && !env_is("ARIA_RESIZE_FUSE", '0') && !aria_getenv("ARIA_RESIZE_IMPL") &&
                         !aria_getenv("ARIA_PYRAMID_IMPL") && !c.level_streams) ? 1 : 0;
        c.band_budget_kb = env_int("ARIA_BAND_BUDGET_KB", 0);
        c.band_qpct0 = env_int("ARIA_BAND_QPCT0", -1);
        c.band_qstep = env_int("ARIA_BAND_QPCT_STEP", -1);
        c.batch_stream = (c.fast_blur_impl == 2 && !env_is("ARIA_FAST_BLUR_IMPL", 'b') && c.fuse_resize && c.stamp_level < 0) ? 1 : 0;
        return c;
    }();
    return cfg;
}

hipEvent_t Profiler::get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
void Profiler::begin(int stage, hipStream_t st) {
    open = enabled && ((stage_mask >> stage) & 1u);
    if (!open) return;
    cur = LaunchEvents{get(), get(), stage, 0};
    hipStreamSynchronize(st);
    hipEventRecord(cur.start, st);
}
void Profiler::end(hipStream_t st) {
    if (!open) return;
    open = false;
    hipEventRecord(cur.stop, st);
    hipStreamSynchronize(st);     // pin the stop marker to the end of this stage, not to whatever is queued next
    pending.push_back(cur);
}
void Profiler::collect() {
    for (LaunchEvents& le : pending) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, le.start, le.stop) == hipSuccess) ms[le.stage] += t;
        launches[le.stage] += le.launches;
        pool.push_back(le.start);
        pool.push_back(le.stop);
    }
    pending.clear();
}
void Profiler::release() {
    collect();
    for (hipEvent_t e : pool) hipEventDestroy(e);
    pool.clear();
}

int LaunchCtx::init(int dev) {
    device = dev;
    const EnvConfig& E = env_config();
    // kernels that may need more than the default 64 KB of dynamic LDS: the attribute belongs to (function, device),
    // so every handle sets it for its own device
    ARIA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_select<false, 256>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(sizeof(unsigned long long) * kSortCapMax)));
    ARIA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_select<true, 256>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(sizeof(unsigned long long) * kSortCapMax)));
    if (int prc = pyramid_set_attributes(); prc != ARIA_OK) return prc;
    int rc = band_set_attributes();
    if (rc != ARIA_OK) return rc;
#ifdef ARIA_VARIANTS
    rc = band2_set_attributes();
    if (rc != ARIA_OK) return rc;
#endif
    rc = stream_set_attributes();
    if (rc != ARIA_OK) return rc;
    rc = band_init_ctx(*this);
    if (rc != ARIA_OK) return rc;
    if (E.sel_stamps) ARIA_HIP(hipMalloc(&d_sel_stamps, sizeof(unsigned long long) * 8 * kLevels * 4096));
    if (E.desc_stamps) ARIA_HIP(hipMalloc(&d_desc_stamps, sizeof(unsigned long long) * 8 * (1u << 22)));
    return ARIA_OK;
}

void LaunchCtx::release() {
    for (int l = 0; l < kLevels; l++) {
        if (side[l]) hipStreamDestroy(side[l]);
        if (ev_join[l]) hipEventDestroy(ev_join[l]);
        if (ev_lvl[l]) hipEventDestroy(ev_lvl[l]);
        side[l] = nullptr; ev_join[l] = nullptr; ev_lvl[l] = nullptr;
    }
    if (ev_fork) hipEventDestroy(ev_fork);
    ev_fork = nullptr;
    hipFree(d_band_stamps); hipFree(d_sel_stamps); hipFree(d_desc_stamps);
    d_band_stamps = d_sel_stamps = d_desc_stamps = nullptr;
}

static bool latency_schedule(const LaunchCtx& ctx, const Profiler* prof) {
    return ctx.schedule == 1 && ctx.ev_fork != nullptr && env_config().fast_blur_impl == 2 && !(prof && prof->enabled);
}

bool latency_zero_copy(const Plan& P, const LaunchCtx& ctx, const Profiler* prof) {
    // opt-in (ARIA_ZERO_COPY=1): measured slower than the upload node, 128 vs 125 us per frame -- the band staging of the
    // pyramid kernel turns into host-link round trips
    static const bool off = [] { const char* e = aria_getenv("ARIA_ZERO_COPY"); return !(e && e[0] == '1'); }();
    const char* f = aria_getenv("ARIA_LATENCY_FORK");        // (the forked level-0 launch reads the device copy at once)
    return !off && !(f && f[0] == '1') && ctx.host_img != nullptr && latency_schedule(ctx, prof) && pyramid_fused_available(P);
}

void launch_extract_chunk(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames,
                          aria_keypoint* d_kps, uint8_t* d_desc, int* d_counts, int kp_cap, hipStream_t st,
                          Profiler* prof, LaunchCtx& ctx) {
    const EnvConfig& E = env_config();
    if (prof && prof->enabled) prof->frames += n_frames;

    // ---- a6.1 pyramid ----
    // Default: no pass of its own -- the FAST/blur launch of level l writes the raw rows of level l+1 from the rows it
    // has staged (fast_blur_band.hip). A separate pass runs only when asked for (ARIA_RESIZE_FUSE=0, ARIA_RESIZE_IMPL,
    // ARIA_PYRAMID_IMPL) or when the band kernel is not the one in use (tile kernel, per-level side streams).
    const bool latency = latency_schedule(ctx, prof);
    const bool fuse_resize = E.fuse_resize != 0 && !latency;
    ctx.last_blur_q4 = false;
    if (!latency) {
        hipMemsetAsync(D.ovf, 0, sizeof(int) * (4 + kLevels * (size_t)n_frames), st);    // arena counters + candidate counters (adjacent)
        if (ctx.hdr) hipMemsetAsync(ctx.hdr, 0, 64, st);                                  // single-frame result header
    }
    if (latency) {
        // single-frame latency schedule: resize chain and the 8 FAST/blur launches overlap (fast_blur_band.hip)
        ctx.last_fast_blur = "k_fast_blur_band";
        launch_pyramid_and_band_latency(P, S, D, n_frames, st, prof, ctx);
    } else {
#ifdef ARIA_VARIANTS
    if (!fuse_resize) {         // stand-alone pyramid pass: variants build only (the product always fuses it)
        if (prof) prof->begin(STAGE_RESIZE, st);
        launch_pyramid_pass(P, S, D, n_frames, st, prof);
        if (prof) prof->end(st);
    }
#endif

    // ---- a6.2 + a6.3 + a6.7 FAST, NMS, blur ----
    if (prof) prof->begin(STAGE_FAST_BLUR, st);
    if (E.batch_stream && fuse_resize && stream_eligible(P, S)) {
        ctx.last_fast_blur = "k_fast_blur_stream";
        ctx.last_blur_q4 = true;
        launch_fast_blur_stream(P, S, D, n_frames, st, prof, ctx);
#ifdef ARIA_VARIANTS
    } else if (E.fast_blur_impl == 1 && !E.level_streams && P.tie_mode <= 1) {
        ctx.last_fast_blur = "k_band2";
        launch_band2(P, S, D, n_frames, st, prof, fuse_resize, ctx);
    } else if (E.fast_blur_impl == 0 && P.tie_mode <= 1) {    // (the variant kernels know blur_tie_mode 0 and 1 only)
        ctx.last_fast_blur = "k_fast_blur";
        launch_fast_blur_tile(P, S, D, n_frames, st, prof);
#endif
    } else {
        ctx.last_fast_blur = "k_fast_blur_band";
        launch_fast_blur_band(P, S, D, n_frames, st, prof, fuse_resize, ctx);
    }
    if (prof) prof->end(st);
    }

    // ---- a6.3-a6.5 selection ----
    if (ctx.stage_events_armed && ctx.stage_event[STAGE_SELECT]) hipEventRecord(ctx.stage_event[STAGE_SELECT], st);
    if (prof) prof->begin(STAGE_SELECT, st);
    // diagnostic: ARIA_SEL_STAMPS=1 prints mean phase lengths per level
    unsigned long long* sstp = (ctx.d_sel_stamps && n_frames <= 4096) ? ctx.d_sel_stamps : nullptr;
    if (sstp) hipMemsetAsync(sstp, 0, sizeof(unsigned long long) * 8 * kLevels * (size_t)n_frames, st);
    if (latency)
        ARIA_LAUNCH(prof, (k_select<true, 256>), dim3(kLevels, n_frames), dim3(256), sizeof(unsigned long long) * (size_t)P.sort_cap,
                    st, P, S, D.raw, D.cand, D.cand_cnt, D.sel, D.sel_cnt, D.err, sstp, D.ovf, D.ovf_items, D.ovf_keys,
                    D.ovf_keys_cap, D.osel, D.osel_cap, E.select_bitonic);
    else
        ARIA_LAUNCH(prof, (k_select<false, 256>), dim3(kLevels, n_frames), dim3(256), sizeof(unsigned long long) * (size_t)P.sort_cap,
                    st, P, S, D.raw, D.cand, D.cand_cnt, D.sel, D.sel_cnt, D.err, sstp, D.ovf, D.ovf_items, D.ovf_keys,
                    D.ovf_keys_cap, D.osel, D.osel_cap, E.select_bitonic);
    // tie-storm fallback: a fixed small grid that finds the work list empty on ordinary images
    if (!latency)
    ARIA_LAUNCH(prof, k_select_ovf, dim3(64), dim3(256), 0, st, P, S, D.raw, D.cand, D.cand_cnt, D.sel, D.sel_cnt, D.err,
                D.ovf, D.ovf_items, D.ovf_keys, D.ovf_keys_cap, D.osel, D.osel_cap);
    if (sstp) {
        hipStreamSynchronize(st);
        std::vector<unsigned long long> hs((size_t)n_frames * kLevels * 8);
        hipMemcpy(hs.data(), sstp, hs.size() * 8, hipMemcpyDeviceToHost);
        for (int l = 0; l < kLevels; l += 7) {
            double ph[5] = {0}; int n = 0;
            for (int f = 0; f < n_frames; f++) {
                const unsigned long long* q = &hs[((size_t)f * kLevels + l) * 8];
                if (!q[5]) continue;
                for (int k = 0; k < 5; k++) ph[k] += (double)(q[k + 1] - q[k]);
                n++;
            }
            if (n) fprintf(stderr, "[select stamps L%d] %d blocks, cycles: hist %.0f cut+compact %.0f harris %.0f sort %.0f emit %.0f\n",
                           l, n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n);
        }
    }

    if (prof) prof->end(st);

    // ---- a6.6 + a6.8 angle + descriptor ----
    if (prof) prof->begin(STAGE_DESCRIBE, st);
    {
        const int bpf = (P.sel_frame_entries + kDescKp - 1) / kDescKp;
        const int frames8 = (n_frames + 7) / 8 * 8;
        const size_t nwaves = (size_t)bpf * frames8 * kDescWaves;
        // diagnostic: ARIA_DESC_STAMPS=1 prints mean phase lengths
        unsigned long long* stp = (ctx.d_desc_stamps && nwaves <= (1u << 22)) ? ctx.d_desc_stamps : nullptr;
        if (stp) hipMemsetAsync(stp, 0, sizeof(unsigned long long) * 8 * nwaves, st);
        if (latency) {
            ARIA_LAUNCH(prof, (k_describe<2, false>), dim3((unsigned)(bpf * frames8 + kDescArenaBlocks)), dim3(64 * kDescWaves), 0, st, P, S,
                        D.raw, D.blur, D.sel, D.sel_cnt, d_kps, d_desc, d_counts, kp_cap, D.err, n_frames, bpf, stp,
                        (const uint4*)D.osel, (const int*)D.ovf, D.osel_cap);
        } else if (ctx.last_blur_q4) {
            ARIA_LAUNCH(prof, (k_describe<0, true>), dim3((unsigned)(bpf * frames8)), dim3(64 * kDescWaves), 0, st, P, S, D.raw, D.blur,
                        D.sel, D.sel_cnt, d_kps, d_desc, d_counts, kp_cap, D.err, n_frames, bpf, stp, (const uint4*)nullptr,
                        (const int*)nullptr, 0);
            // arena pass of the tie-storm fallback: every block finds the arena empty on ordinary images
            ARIA_LAUNCH(prof, (k_describe<1, true>), dim3(256), dim3(64 * kDescWaves), 0, st, P, S,
                        D.raw, D.blur, D.sel, D.sel_cnt, d_kps, d_desc, d_counts, kp_cap, D.err, n_frames, bpf,
                        (unsigned long long*)nullptr, (const uint4*)D.osel, (const int*)D.ovf, D.osel_cap);
        } else {
            ARIA_LAUNCH(prof, (k_describe<0, false>), dim3((unsigned)(bpf * frames8)), dim3(64 * kDescWaves), 0, st, P, S, D.raw, D.blur,
                        D.sel, D.sel_cnt, d_kps, d_desc, d_counts, kp_cap, D.err, n_frames, bpf, stp, (const uint4*)nullptr,
                        (const int*)nullptr, 0);
            ARIA_LAUNCH(prof, (k_describe<1, false>), dim3(256), dim3(64 * kDescWaves), 0, st, P, S,
                        D.raw, D.blur, D.sel, D.sel_cnt, d_kps, d_desc, d_counts, kp_cap, D.err, n_frames, bpf,
                        (unsigned long long*)nullptr, (const uint4*)D.osel, (const int*)D.ovf, D.osel_cap);
        }
        if (stp) {
            hipStreamSynchronize(st);
            std::vector<unsigned long long> hs(nwaves * 8);
            hipMemcpy(hs.data(), stp, hs.size() * 8, hipMemcpyDeviceToHost);
            double ph[6] = {0}; size_t n = 0;
            for (size_t w = 0; w < nwaves; w++) {
                if (!hs[w * 8 + 6] || !hs[w * 8 + 5] || !hs[w * 8 + 1]) continue;      // wave left early (empty slots: only stamps 0 and 6)
                for (int k = 0; k < 6; k++) ph[k] += (double)(hs[w * 8 + k + 1] - hs[w * 8 + k]);
                n++;
            }
            if (n) fprintf(stderr, "[describe stamps] %zu waves, 100 MHz ticks: sel %.1f windows %.1f moments+atan %.1f sincos %.1f pattern %.1f store %.1f\n",
                           n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ph[5] / n);
        }
    }
    if (prof) prof->end(st);
}

}  // namespace aria
