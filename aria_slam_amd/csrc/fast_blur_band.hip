// k_fast_blur_band: FAST-9/16 + NMS + 7x7 Gaussian + the bilinear step to the next pyramid level for one horizontal band
// (13 output rows, 21 staged, x full level width) of one pyramid level of one frame per workgroup.
// SURVEY.md rows a6.1 (pyramid), a6.2 (FAST), a6.3 (border filter), a6.7 (blur).
//
// Why this shape (MI355X):
//  * one lane owns a 4-pixel column strip and walks DOWN the band with a 7-row window in registers, so every staged
//    pixel is read from LDS once as part of an aligned dword (3 ds_read_b32 per 4 px per row) instead of ~45 byte
//    reads per pixel; rows are full level rows, so global loads/stores are fully coalesced 256-B wave rows and
//    there is no horizontal halo to recompute;
//  * the horizontal 7-tap pass is two v_dot4_u32_u8 per pixel on byte-shifted dwords (v_alignbyte_b32); the
//    vertical pass is 7 integer MADs on the register window;
//  * FAST is split: a 4-point compass reject in packed-16 arithmetic on every pixel (a 9-arc always contains one of
//    each antipodal compass pair), survivors (3-30 % of pixels, by level) are appended to an LDS queue and scored DENSELY
//    afterwards (all lanes busy) with the (d, -d) packed min-tree -- score >= threshold <=> FAST-9 corner;
//  * 3x3 strict-max NMS runs only over scored corners, candidates leave through a wave-aggregated append
//    (ballot + popcount prefix, one global atomic per wave).
// Integer/byte work only: no MFMA here (band_mfma.hip is the variant with the blur on the matrix cores). Results are
// bit-identical to the tile kernel it replaced and to that variant (tests/test_gpu_variants.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "orb_device.h"
#include "orb_kernels.h"

namespace aria {

constexpr int kBandR = 13;                 // output rows per strip: R + 8 = 21 staged rows = exactly three 7-row groups of the lane walk
                                           // (12: 2.28, 13: 2.26, 20: 2.33, 32: 3.1 us/frame)
constexpr int kStripRows = kBandR + 8;     // staged rows yb-4 .. yb+R+3 of a strip

// Lanes per row of the walk for a level of width w: one per 4-px column group that some lane HAS to walk.
//  * blur_tie_mode 1: the last, partial group (w mod 4 columns) needs no lane -- its blurred bytes are computed by the
//    integer fix-up after the walk anyway, and it lies outside the FAST range (a 257-px level: 64 lanes, not 65);
//  * blur_tie_mode 1, `trim`: the columns right of the FAST range (x > w - 31) are blur-only. When cutting at most 32 of
//    them brings the lane count down to a whole number of waves, the walk stops there and the fix-up computes those
//    columns too (a 533-px level: 128 lanes + 21 fix-up columns instead of 134 lanes in 192 threads).
static int band_lanes(const Plan& P, int w, bool trim) {
    const int wq = (w + 3) & ~3;
    if (P.tie_mode == 0) return wq >> 2;
    const int full = (w & 3) ? (w >> 2) : (wq >> 2);
    if (!trim || w <= 2 * kEdgeThreshold) return full;
    const int fast_lanes = ((w - kEdgeThreshold) >> 2) + 1;       // lanes whose columns reach into x <= w - 31
    const int down = full & ~63;                                  // whole waves
    if (down >= 64 && down >= fast_lanes && w - 4 * down <= 32 && down < full) return down;
    return full;
}

// Workgroup geometry for a level of width w: NB vertically stacked strips of R rows share one staged block of
// NB*R + 8 rows; thread = (strip, 4-px column). NB is chosen so the waves are full even on narrow levels.
struct BandCfg { int nb, lpr, nthr, qcap; size_t lds; };
static BandCfg band_cfg(const Plan& P, int w, int level) {
    // Occupancy is what this kernel is short of (2-3 waves per SIMD), and LDS is what limits it, so the LDS budget per
    // workgroup is small and the survivor queue is sized by how corner-dense a level is expected to be: the compass
    // test passes ~3 % of level-0 pixels but 25-30 % at level 7 on the benchmark frames (coarse levels pack more
    // structure per pixel). A full queue is not an error: that block takes the slow rescoring path and is counted
    // (aria_orb_slow_path_blocks); the host raises band_qpct0 when it sees such blocks.
    const EnvConfig& E = env_config();
    const int env_budget = E.band_budget_kb, env_q0 = E.band_qpct0, env_qs = E.band_qstep;
    const int budget_kb = env_budget > 0 ? env_budget : P.band_budget_kb;
    const int q0 = env_q0 >= 0 ? env_q0 : P.band_qpct0, qstep = env_qs >= 0 ? env_qs : P.band_qstep;
    const size_t budget = (size_t)budget_kb * 1024;
    const int qpct = std::min(50, q0 + qstep * level);
    const int wq = (w + 3) & ~3, lpr = band_lanes(P, w, false);
    BandCfg best{};
    double best_util = -1.0;
    {   // one strip per workgroup with the lane count trimmed to whole waves, when the level allows it
        const int lt = band_lanes(P, w, E.band_trim != 0);
        if (lt < lpr) {
            int qcap = (int)((int64_t)kBandR * wq * qpct / 100);
            qcap = std::max(512, (qcap + 63) & ~63);
            return BandCfg{1, lt, lt, qcap, (size_t)(kBandR + 8) * (wq + 8) + 4 * (size_t)qcap};
        }
    }
    for (int nb = 1; nb <= 8; nb++) {
        const int need = nb * lpr, nthr = ((need + 63) / 64) * 64;
        if (nthr > 512) break;
        const size_t pix = (size_t)(nb * kBandR + 8) * (wq + 8);
        const int px_blk = nb * kBandR * wq;
        int qcap = (int)((int64_t)px_blk * qpct / 100);
        qcap = std::max(512, (qcap + 63) & ~63);
        const size_t lds = pix + 4 * (size_t)qcap;
        if (nb > 1 && lds > budget) break;
        const double util = (double)need / nthr;
        if (util > best_util + 0.02) { best_util = util; best = BandCfg{nb, lpr, nthr, qcap, lds}; }
    }
    return best;
}

// v_mad_u32_u24: full-rate 24-bit multiply-add (operands here are < 2^24)
__device__ __forceinline__ uint32_t umad24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

typedef short short2v __attribute__((ext_vector_type(2)));
typedef unsigned short us2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
}
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, a) - __builtin_bit_cast(short2v, b));
}
__device__ __forceinline__ uint32_t pk_add_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, a) + __builtin_bit_cast(short2v, b));
}

// fast_score.cpp cornerScore<16> for the pixel at LDS address c (row pitch `pitch`):
// max over the 16 nine-arcs of min(v - ring) and of min(ring - v), minus 1; both polarities ride in one register as
// packed int16 (lo = v - p, hi = p - v). The pixel is a FAST-9 corner for threshold t iff the result is >= t.
__device__ __forceinline__ int fast_score_pk(const uint8_t* c, int pitch) {
    const uint8_t* rm3 = c - 3 * pitch; const uint8_t* rm2 = c - 2 * pitch; const uint8_t* rm1 = c - pitch;
    const uint8_t* rp1 = c + pitch; const uint8_t* rp2 = c + 2 * pitch; const uint8_t* rp3 = c + 3 * pitch;
    const uint32_t v = c[0];
    uint32_t ring[16] = {rp3[0], rp3[1], rp2[2], rp1[3], c[3], rm1[3], rm2[2], rm3[1],
                         rm3[0], rm3[-1], rm2[-2], rm1[-3], c[-3], rp1[-3], rp2[-2], rp3[-1]};
    return fast9_score_f16(v, ring);      // packed half floats, three-input minimum / maximum (orb_device.h, round 4)
}

template <int TIE_EVEN, int NB1>
__global__ __launch_bounds__(512) void k_fast_blur_band(Plan P, FrameSrc S, const uint8_t* __restrict__ raw,
                                                        uint8_t* __restrict__ blur, uint32_t* __restrict__ cand,
                                                        int* __restrict__ cand_cnt, int* __restrict__ err, int l_arg,
                                                        int nb, int qcap_arg, int ablate, unsigned long long* __restrict__ stamps,
                                                        int* __restrict__ slow_blocks, const uint32_t* __restrict__ tab,
                                                        uint8_t* __restrict__ raw_next, BandAll ba) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_qn, s_ovf;
    // l_arg >= 0: this launch is level l_arg, blockIdx.x = strip. l_arg < 0 (single-frame latency schedule): ONE launch
    // covers every level -- blockIdx.x runs over the strips of all levels (ba.first = prefix sums), one strip per workgroup
    int l = l_arg, strip = blockIdx.x, qcap = qcap_arg, frame = blockIdx.y;
    if (ba.xcd_map) {
        // Workgroups go round-robin to the 8 XCDs (each with its own L2) by linear id. Neighbouring strips of a frame
        // share 8 of their 21 staged rows, so all strips of a frame are given to ONE XCD (frame f -> XCD f mod 8): the
        // shared rows then come from that XCD's L2 instead of HBM twice. Needs a frame count that is a multiple of 8.
        const uint32_t b = blockIdx.y * gridDim.x + blockIdx.x, j = b >> 3;
        const uint32_t fq = __builtin_amdgcn_readfirstlane(j / gridDim.x);
        strip = (int)(j - fq * gridDim.x);
        frame = (int)(fq * 8u + (b & 7u));
    }
    if (l_arg < 0) {
        l = 0;
#pragma unroll
        for (int i = 1; i < kLevels; i++) if ((int)blockIdx.x >= ba.first[i]) l = i;
        strip = (int)blockIdx.x - ba.first[l];
        qcap = ba.qcap[l];
    }
    // diagnostic build only (ARIA_STAMPS=1): phase boundaries of wave 0 of every workgroup, s_memtime ticks
#define STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    STAMP(0);

    const LevelGeom g = P.lv[l];
    const int w = g.w, h = g.h;
    const int wq = (w + 3) & ~3;
    const int lpr = ba.lpr[l];                           // lanes per row of the walk (band_lanes on the host)
    const int pitchL = wq + 8;                           // LDS pitch of the staged block: column 0 <-> x = -4
    const int sp = wq + 4;                               // score-map pitch: column 0 <-> x = -1
    const int RB = nb * kBandR;                          // output rows of this workgroup
    const int rowsL = RB + 8;                            // staged rows y0-4 .. y0+RB+3
    uint8_t* s_pix = smem;
    uint8_t* s_map = smem;                               // score map, aliases s_pix once the pixels are dead
    uint32_t* s_queue = reinterpret_cast<uint32_t*>(smem + rowsL * pitchL);

    const int tid = threadIdx.x, nthr = blockDim.x;
    const int y0 = strip * RB;
    int pitch;
    const uint8_t* img = raw_level_ptr(P, S, raw, frame, l, pitch);

    if (tid == 0) { s_qn = 0; s_ovf = 0; }
    // stage rows y0-4 .. y0+RB+3, columns -4 .. wq+3, BORDER_REFLECT_101 outside the level.
    // Interior: 16-byte global loads, four in flight per lane before the first LDS write (a load-then-store loop
    // would serialise on HBM latency); edges (x < 0, x >= 16*floor(w/16)): per-byte reflected loads.
    const int dpr = pitchL >> 2;
    const int nch = ((l > 0) || S.aligned16) ? (w >> 4) : 0;     // full 16-byte chunks per row
    if (nch > 0 && !(ablate & 1)) {
        const int rpp = nthr / nch;                               // rows staged per pass (nthr >= w/4 > nch)
        const int my_r = tid / nch, my_c = tid - my_r * nch;
        if (my_r < rpp) {
            for (int r0 = my_r; r0 < rowsL; r0 += 4 * rpp) {
                uint4 v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int r = r0 + k * rpp;
                    if (r < rowsL) {
                        const int gy = reflect101(y0 - 4 + r, h);
                        v[k] = *reinterpret_cast<const uint4*>(img + (int64_t)gy * pitch + 16 * my_c);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int r = r0 + k * rpp;
                    if (r < rowsL) {
                        uint32_t* d = reinterpret_cast<uint32_t*>(s_pix + r * pitchL + 4 + 16 * my_c);
                        d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
                    }
                }
            }
        }
    }
    {
        const int xe0 = nch * 16;                                  // first column not covered by full chunks
        const int ne = 1 + ((wq + 4 - xe0) >> 2);                  // edge dwords per row: x = -4 and x >= xe0
        for (int i = tid; i < ((ablate & 1) ? 0 : rowsL * ne); i += nthr) {
            const int r = i / ne, e = i - r * ne;
            const int gx = e == 0 ? -4 : xe0 + 4 * (e - 1);
            const int gy = reflect101(y0 - 4 + r, h);
            const uint8_t* rowp = img + (int64_t)gy * pitch;
            const uint32_t v = (uint32_t)rowp[reflect101(gx, w)] | ((uint32_t)rowp[reflect101(gx + 1, w)] << 8) |
                               ((uint32_t)rowp[reflect101(gx + 2, w)] << 16) | ((uint32_t)rowp[reflect101(gx + 3, w)] << 24);
            *reinterpret_cast<uint32_t*>(s_pix + r * pitchL + (gx + 4)) = v;
        }
    }
    __syncthreads();
    STAMP(1);

    // keypoints.cpp runByImageBorder keeps x in [31, w-31), y in [31, h-31); FAST is needed on that region + 1 ring
    const bool level_has_kp = (w > 2 * kEdgeThreshold) && (h > 2 * kEdgeThreshold);
    const int fx0 = kEdgeThreshold - 1, fx1 = w - kEdgeThreshold;        // inclusive FAST ranges
    const int fy0 = kEdgeThreshold - 1, fy1 = h - kEdgeThreshold;
    const int thr = P.fast_threshold;

    // NB1: a single strip per workgroup (the common case) -- the strip index is then wave-uniform, so the per-row
    // guards below compile to scalar branches instead of exec-mask updates
    const int sb = NB1 ? 0 : tid / lpr, li = NB1 ? tid : tid - sb * lpr;       // strip, column group
    if ((NB1 ? tid < lpr : sb < nb) && !(ablate & 2)) {
        const int x = li * 4;
        const int yb = y0 + sb * kBandR;                  // first output row of this strip
        // per-lane packed masks of the pixels inside the FAST x range (bit 15: px 0/2, bit 31: px 1/3 of a pair)
        uint32_t xm0 = 0, xm1 = 0;
        if (level_has_kp) {
            if (x + 0 >= fx0 && x + 0 <= fx1) xm0 |= 0x00008000u;
            if (x + 1 >= fx0 && x + 1 <= fx1) xm0 |= 0x80000000u;
            if (x + 2 >= fx0 && x + 2 <= fx1) xm1 |= 0x00008000u;
            if (x + 3 >= fx0 && x + 3 <= fx1) xm1 |= 0x80000000u;
        }
        const uint32_t T2 = (uint32_t)thr * 0x00010001u;
        const int o_min = sb == 0 ? -1 : 0, o_max = sb == nb - 1 ? kBandR : kBandR - 1;
        // uniform base + 32-bit lane offset: the store takes the scalar-base form and the row address is one v_mad_u32_u24
        // (a 64-bit per-lane pointer cost a v_mad_i64_i32 per output row)
        uint8_t* bl = blur + (int64_t)frame * P.blur_frame_bytes + g.blur_off;

        const uint32_t KLO = 18u | (34u << 8) | (49u << 16) | (55u << 24);   // taps x-3..x
        const uint32_t KHI = 49u | (34u << 8) | (18u << 16);                 // taps x+1..x+3 (x+4 weight 0)

        // 7-row register window; slot = step % 7 (static after unrolling by 7)
        uint32_t RS[7][4], RC2[7][2], RE[7], RW[7];
        float RF[7][4];       // the row sums as floats (blur_tie_mode 1: the column pass runs in fp32, see below)
#pragma unroll
        for (int u = 0; u < 7; u++) {
            RC2[u][0] = RC2[u][1] = RE[u] = RW[u] = 0; RS[u][0] = RS[u][1] = RS[u][2] = RS[u][3] = 0;
            RF[u][0] = RF[u][1] = RF[u][2] = RF[u][3] = 0.f;
        }

        const uint32_t* lrow = reinterpret_cast<const uint32_t*>(s_pix) + (sb * kBandR) * dpr + li;
        for (int tb = 0; tb < kStripRows; tb += 7) {
            uint32_t accw = 0;     // survivors of this group of 7 steps: step u, px j -> bit (j&1 ? 31 : 15) - (j>>1) - 2u
#pragma unroll
            for (int u = 0; u < 7; u++) {
                const int t = tb + u;             // staged row of the strip; level row = yb - 4 + t
                if (t < kStripRows) {
                    const uint32_t* rp = lrow + t * dpr;
                    const uint32_t w0 = rp[0], w1 = rp[1], w2 = rp[2];
                    RC2[u][0] = __builtin_amdgcn_perm(0u, w1, 0x0c010c00u);   // px 0, 1 as int16 pair
                    RC2[u][1] = __builtin_amdgcn_perm(0u, w1, 0x0c030c02u);   // px 2, 3
                    RW[u] = __builtin_amdgcn_alignbyte(w1, w0, 1);     // x-3 .. x
                    RE[u] = __builtin_amdgcn_alignbyte(w2, w1, 3);     // x+3 .. x+6
                    RS[u][0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), KHI,
                                                      __builtin_amdgcn_udot4(RW[u], KLO, 0u, false), false);
                    RS[u][1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), KHI,
                                                      __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), KLO, 0u, false), false);
                    RS[u][2] = __builtin_amdgcn_udot4(RE[u], KHI,
                                                      __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), KLO, 0u, false), false);
                    RS[u][3] = __builtin_amdgcn_udot4(w2, KHI, __builtin_amdgcn_udot4(w1, KLO, 0u, false), false);
                    if (TIE_EVEN) {
#pragma unroll
                        for (int j = 0; j < 4; j++) RF[u][j] = (float)RS[u][j];      // exact: < 2^16
                    }

                    const int o = t - 7;              // strip row whose window [o-3, o+3] is now complete
                    const int Y = yb + o;
                    // row o+d of the window lives in slot (u + 4 + d) % 7
                    const int sC = (u + 4) % 7, sM1 = (u + 3) % 7, sP1 = (u + 5) % 7, sM2 = (u + 2) % 7, sP2 = (u + 6) % 7,
                              sM3 = (u + 1) % 7, sP3 = u;
                    if (o >= 0 && o < kBandR && Y < h) {
                        // vertical pass + rounding by 2^16 (filter.simd.hpp SymmColumnFilter / SymmColumnVec_32s8u)
                        uint32_t outw = 0;
                        if (TIE_EVEN) {
                            // blur_tie_mode 1: OpenCV's SymmColumnVec_32s8u rounds the column sums to nearest EVEN, which is
                            // what v_cvt_pk_u8_f32 does (plus saturation and the byte insert) in one instruction. All values
                            // are integers * 2^-16 below 2^9, so the fp32 column pass is exact (sums < 2^25: a partial sum
                            // can only be inexact above 256.0, which saturates either way); 7 fast-rate float ops per pixel
                            // instead of 3 adds + 4 24-bit multiplies + 4 integer rounding ops. The scalar-tail columns
                            // (x >= w & ~3, ties round UP there) are redone in integers after the walk.
                            constexpr float k0 = 55.f / 65536.f, k1 = 49.f / 65536.f, k2 = 34.f / 65536.f, k3 = 18.f / 65536.f;
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                float acc = k0 * RF[sC][j];
                                acc = __builtin_fmaf(k1, RF[sM1][j] + RF[sP1][j], acc);
                                acc = __builtin_fmaf(k2, RF[sM2][j] + RF[sP2][j], acc);
                                acc = __builtin_fmaf(k3, RF[sM3][j] + RF[sP3][j], acc);
                                outw = __builtin_amdgcn_cvt_pk_u8_f32(acc, j, outw);
                            }
                        } else {
                        constexpr uint32_t put[4] = {0x03020106u, 0x03020600u, 0x03060100u, 0x06020100u};   // byte 2 of r -> byte j
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            // operands < 2^24: 24-bit multiplies (v_mad_u32_u24) are full rate, v_mul_lo_u32 is not
                            uint32_t acc = __umul24(55u, RS[sC][j]);
                            acc = umad24(49u, RS[sM1][j] + RS[sP1][j], acc);
                            acc = umad24(34u, RS[sM2][j] + RS[sP2][j], acc);
                            acc = umad24(18u, RS[sM3][j] + RS[sP3][j], acc);
                            // ties up: + 0x8000, quotient by 2^16, saturate to 255
                            const uint32_t r = acc + 0x8000u;
                            outw = __builtin_amdgcn_perm(min(r, 0x00FFFFFFu), outw, put[j]);
                        }
                        }
                        *reinterpret_cast<uint32_t*>(bl + (uint32_t)(Y * g.pitch + x)) = outw;
                    }
                    // the ring rows (o = -1, o = R) are only this strip's job at the block's top / bottom edge;
                    // inside the block they are ordinary rows of the neighbouring strip
                    if (o >= o_min && o <= o_max && Y >= fy0 && Y <= fy1) {
                        // compass reject, two pixels per packed-int16 op: survive iff (N|S)&(E|W) are all darker than
                        // c - t or all brighter than c + t (every 9-arc holds one pixel of each antipodal compass pair)
                        // darker: both compass pairs have a member below c - t  <=>  max(min(n, s), min(e, w)) < c - t;
                        // brighter: min(max(n, s), max(e, w)) > c + t. The centre row's bytes were widened to int16
                        // pairs when the row entered the window (it serves as n, c and s of three output rows).
                        uint32_t pass[2];
#pragma unroll
                        for (int pr = 0; pr < 2; pr++) {
                            const uint32_t sel = pr ? 0x0c030c02u : 0x0c010c00u;
                            const uint32_t c2 = RC2[sC][pr], n2 = RC2[sM3][pr], s2 = RC2[sP3][pr];
                            const uint32_t e2 = __builtin_amdgcn_perm(0u, RE[sC], sel);
                            const uint32_t w2p = __builtin_amdgcn_perm(0u, RW[sC], sel);
                            const uint32_t lo = pk_sub_i16(c2, T2), hi = pk_add_i16(c2, T2);
                            // sign bit set <=> darker than c - t / brighter than c + t
                            const uint32_t dk = pk_sub_i16(pk_max_i16(pk_min_i16(n2, s2), pk_min_i16(e2, w2p)), lo);
                            const uint32_t br = pk_sub_i16(hi, pk_min_i16(pk_max_i16(n2, s2), pk_max_i16(e2, w2p)));
                            pass[pr] = (dk | br) & (pr ? xm1 : xm0);
                        }
                        accw |= (pass[0] | (pass[1] >> 1)) >> (2 * u);
                    }
                }
            }
            // append this group's survivors: one LDS atomic per lane, then one store per survivor
            if (accw) {
                int base = atomicAdd(&s_qn, __popc(accw));
                while (accw) {
                    const int b = 31 - __clz(accw);
                    accw &= ~(1u << b);
                    const int hi16 = b >> 4, bb = 15 - (b & 15);
                    const int u = bb >> 1, px = ((bb & 1) << 1) | hi16;
                    const uint32_t ro = (uint32_t)(sb * kBandR + tb + u - 6);    // block score row of strip row o = t-7: o+1
                    if (base < qcap) s_queue[base] = (uint32_t)(x + px) | (ro << 11);
                    else s_ovf = 1;    // queue full (pathological image): the block is rescored densely below
                    base++;
                }
            }
        }

        // ---- a6.1 fused: this workgroup also produces the rows of level l+1 whose upper source row it owns, from the
        //      level-l rows it has staged (the bilinear resize needs rows oy, oy+1 <= y0 + RB, inside the halo), with
        //      the arithmetic of k_resize_lds<true> (v_perm + v_dot2_u32_u16; same integers). The separate resize pass
        //      and its re-read of level l from HBM disappear. A lane keeps the x-table words of its output dword in
        //      registers and walks the band's ~RB/1.2 output rows; oy is monotone in dy, so the candidate rows are
        //      bracketed arithmetically and the y table decides ownership exactly. ----
        if (raw_next != nullptr) {
            const LevelGeom gn = P.lv[l + 1];
            const int groups_n = gn.pitch >> 2;
            uint8_t* dstl = raw_next + (int64_t)frame * P.raw_frame_bytes + gn.raw_off;
            const uint32_t* xt = tab + gn.xtab;
            const uint32_t* yt = tab + gn.ytab;
            const int y_end = min(y0 + RB, h);
            const int dlo = max((int)(((int64_t)y0 * gn.h) / h) - 2, 0);
            const int dhi = min((int)(((int64_t)y_end * gn.h + h - 1) / h) + 2, gn.h);
            constexpr uint32_t put[4] = {0x03020106u, 0x03020600u, 0x03060100u, 0x06020100u};   // byte 2 of v -> byte i
            for (int gx = li; gx < groups_n; gx += lpr) {
                const int dx0 = gx * 4;
                uint32_t xw[4], xo[4], xs[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t t = xt[min(dx0 + i, gn.w - 1)];
                    const uint32_t ox = t & 0xFFFFu, cx1 = t >> 16;
                    xw[i] = (256u - cx1) | (cx1 << 16);
                    xo[i] = (ox & ~3u) + 4u;                            // staged column 0 is x = -4
                    xs[i] = 0x0C010C00u + (ox & 3u) * 0x00010001u;
                }
                // y-table words of a batch of candidate rows are fetched together (one global round trip per 16 rows,
                // not one per row)
                // NB1 (rows in order, dy uniform in the wave): the horizontal blend of source row oy+1 is kept and serves as
                // the TOP row of the next output row whenever that one starts at oy+1 (5 rows out of 6 at scale 1.2)
                uint32_t hprev[4] = {0u, 0u, 0u, 0u};
                int hprev_row = -1;
                for (int d0 = dlo + sb; d0 < dhi; d0 += 16 * nb) {
                uint32_t tyv[16];
#pragma unroll
                for (int k = 0; k < 16; k++) tyv[k] = yt[min(d0 + k * nb, gn.h - 1)];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int dy = d0 + k * nb;
                    const uint32_t ty = tyv[k];
                    const int oy = ty & 0xFFFF;
                    if (dy >= dhi || oy < y0 || oy >= y_end) continue;  // past the bracket / another workgroup's row
                    uint32_t outw = 0;
                    if (dx0 < gn.w) {
                        const uint32_t cy1 = ty >> 16, cyp = (256u - cy1) | (cy1 << 16);
                        const int rb = min(oy + 1, h - 1);
                        const uint8_t* rowa = s_pix + (oy - y0 + 4) * pitchL;
                        const uint8_t* rowb = s_pix + (rb - y0 + 4) * pitchL;
                        const bool reuse = NB1 && oy == hprev_row;       // wave-uniform
                        uint32_t h0v[4];
                        if (reuse) {
#pragma unroll
                            for (int i = 0; i < 4; i++) h0v[i] = hprev[i];
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; i++) {
                                const uint32_t* qa = reinterpret_cast<const uint32_t*>(rowa + xo[i]);
                                const uint32_t top = __builtin_amdgcn_perm(qa[1], qa[0], xs[i]);     // p00 | p01 << 16
                                h0v[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, top), __builtin_bit_cast(us2v, xw[i]), 0u, false);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const uint32_t* qb = reinterpret_cast<const uint32_t*>(rowb + xo[i]);
                            const uint32_t bot = __builtin_amdgcn_perm(qb[1], qb[0], xs[i]);
                            const us2v wx = __builtin_bit_cast(us2v, xw[i]);
                            const uint32_t h0 = h0v[i];
                            const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, bot), wx, 0u, false);
                            const uint32_t v = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, h0 | (h1 << 16)),
                                                                      __builtin_bit_cast(us2v, cyp), 32768u, false);   // < 2^24
                            outw = __builtin_amdgcn_perm(v, outw, put[i]);
                            hprev[i] = h1;
                        }
                        hprev_row = rb;
                    }
                    *reinterpret_cast<uint32_t*>(dstl + (int64_t)dy * gn.pitch + dx0) = outw;
                }
                }
            }
        }
    }
    STAMP(2);
    __syncthreads();     // (also waits for this wave's blurred-row stores: the fix-up below overwrites bytes of them)
    STAMP(3);

    // ---- blur_tie_mode 1: OpenCV's SIMD column filter covers x < (w & ~3) (ties to even: what the fp32 column pass above
    //      computed); its scalar tail, the last w mod 4 columns, rounds ties UP. Those <= 3 columns per row are recomputed
    //      here in integers from the staged pixels (filter.simd.hpp SymmColumnFilter) and overwrite the bytes stored above. ----
    // The same loop computes every column the walk left out (x >= 4 lpr: band_lanes' trimmed blur-only columns), with the
    // SIMD path's round-half-to-even for x < (w & ~3).
    // blur_tie_mode 2 / 3 (vector body ends at w & ~7 / w & ~15): the same loop, started at the first tail column.
    const int tail_start = P.tie_mode == 2 ? (w & ~7) : P.tie_mode == 3 ? (w & ~15) : (w & ~3);
    if (TIE_EVEN && min(4 * lpr, tail_start) < w && !(ablate & 2)) {
        // thread -> (column, part of the block's rows): it walks its rows top to bottom with the seven horizontal sums of
        // the window in registers, so a staged pixel is read once per column it contributes to (7 byte reads + the
        // vertical pass per output pixel, not 49)
        const int xb = min(4 * lpr, tail_start), nt = w - xb, x_even_end = tail_start;
        const int np = max(1, min(RB, 64 / nt));                 // row parts: one wave's worth of threads
        const int rows_pp = (RB + np - 1) / np;
        uint8_t* blf = blur + (int64_t)frame * P.blur_frame_bytes + g.blur_off;
        if (nt <= 3) {
            // the plain scalar tail: one (row, column) per thread, all 49 taps at once (independent loads: shortest chain)
            for (int i = tid; i < RB * nt; i += nthr) {
                const int ro = i / nt, xx = xb + (i - ro * nt);
                const int Y = y0 + ro;
                if (Y >= h) continue;
                uint32_t acc = 0;
#pragma unroll
                for (int dy = 0; dy < 7; dy++) {
                    const uint8_t* p = s_pix + (ro + 1 + dy) * pitchL + (xx + 4);
                    const uint32_t hs = 18u * (p[-3] + p[3]) + 34u * (p[-2] + p[2]) + 49u * (p[-1] + p[1]) + 55u * p[0];
                    const uint32_t kv = dy == 0 || dy == 6 ? 18u : dy == 1 || dy == 5 ? 34u : dy == 2 || dy == 4 ? 49u : 55u;
                    acc += kv * hs;
                }
                const uint32_t bias = xx < x_even_end ? 0x7FFFu + ((acc >> 16) & 1u) : 0x8000u;
                blf[(int64_t)Y * g.pitch + xx] = (uint8_t)min((acc + bias) >> 16, 255u);
            }
        } else if (tid < nt * np) {
            const int part = tid / nt, xx = xb + (tid - part * nt);
            const int r0 = part * rows_pp, r1 = min(RB, r0 + rows_pp);
            uint32_t hsw[7] = {0, 0, 0, 0, 0, 0, 0};
            const uint8_t* pc = s_pix + (xx + 4);
            for (int k = r0 + 1; k < r1 + 7; k++) {              // staged rows; output row ro = k - 7 has its window complete
                const uint8_t* p = pc + k * pitchL;
                const uint32_t hs = 18u * (p[-3] + p[3]) + 34u * (p[-2] + p[2]) + 49u * (p[-1] + p[1]) + 55u * p[0];
#pragma unroll
                for (int u = 0; u < 6; u++) hsw[u] = hsw[u + 1];
                hsw[6] = hs;
                const int ro = k - 7, Y = y0 + ro;
                if (ro >= r0 && Y < h) {
                    const uint32_t acc = 18u * (hsw[0] + hsw[6]) + 34u * (hsw[1] + hsw[5]) + 49u * (hsw[2] + hsw[4]) + 55u * hsw[3];
                    // ties up: + 0x8000; ties to even: + 0x7FFF + the quotient's low bit
                    const uint32_t bias = xx < x_even_end ? 0x7FFFu + ((acc >> 16) & 1u) : 0x8000u;
                    blf[(int64_t)Y * g.pitch + xx] = (uint8_t)min((acc + bias) >> 16, 255u);
                }
            }
        }
    }

    uint32_t* clist = cand + (int64_t)frame * P.cand_frame_entries + g.cand_off;
    int* ccnt = cand_cnt + frame * kLevels + l;
    const int lane = tid & 63;
    const bool ovf = s_ovf != 0;
    if (ovf && tid == 0) atomicAdd(slow_blocks, 1);
    const int qn = (ovf || (ablate & 4)) ? 0 : s_qn;

    // ---- dense scoring of the queued survivors (all lanes busy); score >= t <=> FAST-9 corner. Two entries per lane
    //      and iteration: the 2 x 17 LDS gathers overlap, which matters at 2-3 waves per workgroup. Corners are
    //      compacted to the front part of the queue memory (s_corner) so the later passes only visit corners. ----
    __shared__ int s_cn;
    if (tid == 0) s_cn = 0;
    __syncthreads();
    uint32_t* s_corner = s_queue;          // in-place compaction is safe: a corner's slot index is <= entries consumed
    // (each iteration reads its entries into registers before any thread of the block writes: see the barrier below)
    for (int i0 = 0; i0 < qn; i0 += 2 * nthr) {
        const int ia = i0 + tid, ib = i0 + nthr + tid;
        const uint32_t ea = ia < qn ? s_queue[ia] : 0u, eb = ib < qn ? s_queue[ib] : 0u;
        int sa = 0, sb2 = 0;
        if (ia < qn) sa = fast_score_pk(s_pix + (((ea >> 11) & 0x1FFF) + 3) * pitchL + ((ea & 0x7FF) + 4), pitchL);
        if (ib < qn) sb2 = fast_score_pk(s_pix + (((eb >> 11) & 0x1FFF) + 3) * pitchL + ((eb & 0x7FF) + 4), pitchL);
        const bool ca = ia < qn && sa >= thr, cb = ib < qn && sb2 >= thr;
        __syncthreads();                   // every thread holds its entries of this round in registers
        const unsigned long long ma = __ballot(ca), mb = __ballot(cb);
        int base = 0;
        if (lane == 0 && (ma | mb)) base = atomicAdd(&s_cn, __popcll(ma) + __popcll(mb));
        base = __shfl(base, 0);
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (ca) s_corner[base + __popcll(ma & lt)] = ea | ((uint32_t)sa << 24);
        if (cb) s_corner[base + __popcll(ma) + __popcll(mb & lt)] = eb | ((uint32_t)sb2 << 24);
        __syncthreads();
    }
    STAMP(4);
    const int cn = ovf ? 0 : s_cn;
    if (!ovf) {
        // the staged pixels are dead: reuse their LDS as the score map (rows ro = 0 .. RB+1, columns x+1)
        for (int i = tid; i < ((RB + 2) * sp) >> 2; i += nthr) reinterpret_cast<uint32_t*>(s_map)[i] = 0u;
        __syncthreads();
        for (int i = tid; i < cn; i += nthr) {
            const uint32_t ent = s_corner[i];
            s_map[((ent >> 11) & 0x1FFF) * sp + (ent & 0x7FF) + 1] = (uint8_t)(ent >> 24);
        }
        __syncthreads();
    }

    // ---- 3x3 strict-max NMS + border filter. Survivors are compacted into an LDS list (ballot + popcount prefix,
    //      one LDS atomic per wave), then the workgroup reserves its slice of the global candidate list with ONE
    //      global atomic and copies the list out: no wave ever waits on an L2 atomic round trip in the loop. ----
    __shared__ int s_kn, s_gbase;
    const int map_bytes = (((RB + 2) * sp) + 3) & ~3;
    uint32_t* s_keep = reinterpret_cast<uint32_t*>(smem + map_bytes);            // spare tail of the staging area
    const int keep_cap = ovf ? 0 : (rowsL * pitchL - map_bytes) >> 2;
    if (tid == 0) s_kn = 0;
    __syncthreads();
    STAMP(5);
    const int n_items = (ablate & 8) ? 0 : (ovf ? RB * wq : cn);
    for (int i0 = 0; i0 < n_items; i0 += nthr) {
        const int i = i0 + tid;
        bool keep = false;
        int X = 0, Y = 0, sc = 0;
        if (i < n_items) {
            if (!ovf) {
                const uint32_t ent = s_corner[i];
                X = ent & 0x7FF;
                const int ro = (ent >> 11) & 0x1FFF;
                sc = ent >> 24;
                Y = y0 + ro - 1;
                if (sc > 0 && ro >= 1 && ro <= RB && X >= kEdgeThreshold && X < w - kEdgeThreshold &&
                    Y >= kEdgeThreshold && Y < h - kEdgeThreshold) {
                    const uint8_t* s = &s_map[ro * sp + X + 1];
                    keep = sc > s[-1] && sc > s[1] && sc > s[-sp - 1] && sc > s[-sp] && sc > s[-sp + 1] &&
                           sc > s[sp - 1] && sc > s[sp] && sc > s[sp + 1];
                }
            } else {
                // slow path for pathological images: score the pixel and, for corners, its 8 neighbours in place
                const int ro = i / wq + 1;
                X = i - (ro - 1) * wq;
                Y = y0 + ro - 1;
                if (level_has_kp && X >= kEdgeThreshold && X < w - kEdgeThreshold && Y >= kEdgeThreshold &&
                    Y < h - kEdgeThreshold) {
                    const uint8_t* c = s_pix + (ro + 3) * pitchL + (X + 4);
                    sc = fast_score_pk(c, pitchL);
                    if (sc >= thr) {
                        keep = true;
                        for (int dy = -1; dy <= 1 && keep; dy++)
                            for (int dx = -1; dx <= 1; dx++) {
                                if (dx == 0 && dy == 0) continue;
                                int ns = fast_score_pk(c + dy * pitchL + dx, pitchL);
                                if (ns < thr) ns = 0;
                                if (!(sc > ns)) { keep = false; break; }
                            }
                    }
                }
            }
        }
        const unsigned long long mask = __ballot(keep);
        if (mask) {
            const int leader = __ffsll((long long)mask) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&s_kn, __popcll(mask));
            base = __shfl(base, leader);
            const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
            if (keep) {
                const uint32_t rec = (uint32_t)X | ((uint32_t)Y << 11) | ((uint32_t)sc << 22);
                if (pos < keep_cap) {
                    s_keep[pos] = rec;
                } else {   // LDS list full (or overflow path): straight to the global list
                    const int gp = atomicAdd(ccnt, 1);
                    if (gp < g.cand_cap) clist[gp] = rec;
                    else atomicOr(err, ERRBIT_CAND_OVERFLOW);
                }
            }
        }
    }
    STAMP(6);
    __syncthreads();
    const int kn = min(s_kn, keep_cap);
    if (tid == 0 && kn > 0) s_gbase = atomicAdd(ccnt, kn);
    __syncthreads();
    if (kn > 0) {
        const int gb = s_gbase;
        for (int i = tid; i < kn; i += nthr) {
            if (gb + i < g.cand_cap) clist[gb + i] = s_keep[i];
            else atomicOr(err, ERRBIT_CAND_OVERFLOW);
        }
    }
    STAMP(7);
#undef STAMP
}

int band_set_attributes() {
    const int lds_max = 160 * 1024 - 64;
    ARIA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fast_blur_band<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    ARIA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fast_blur_band<1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    ARIA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fast_blur_band<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    ARIA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fast_blur_band<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    return ARIA_OK;
}

int band_side_streams(LaunchCtx& ctx) {
    if (ctx.ev_fork) return ARIA_OK;
    ARIA_HIP(hipEventCreateWithFlags(&ctx.ev_fork, hipEventDisableTiming));
    for (int l = 0; l < kLevels; l++) {
        ARIA_HIP(hipStreamCreateWithFlags(&ctx.side[l], hipStreamNonBlocking));
        ARIA_HIP(hipEventCreateWithFlags(&ctx.ev_join[l], hipEventDisableTiming));
        ARIA_HIP(hipEventCreateWithFlags(&ctx.ev_lvl[l], hipEventDisableTiming));
    }
    return ARIA_OK;
}

int band_init_ctx(LaunchCtx& ctx) {
    const EnvConfig& E = env_config();
    // The 8 levels are independent; ARIA_LEVEL_STREAMS=1 forks them onto side streams and joins back on the caller's
    // stream (measured: no gain over back-to-back launches at chunk >= 256; off by default for batches, used by the
    // single-frame latency schedule).
    if (E.level_streams) {
        int rc = band_side_streams(ctx);
        if (rc != ARIA_OK) return rc;
    }
    if (E.stamp_level >= 0) ARIA_HIP(hipMalloc(&ctx.d_band_stamps, sizeof(unsigned long long) * 16 * 32768));
    return ARIA_OK;
}

// One level's launch on stream s (no stream orchestration here).
static void band_launch_level(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t s, Profiler* prof,
                              bool fuse_resize, LaunchCtx& ctx, int l) {
    const EnvConfig& E = env_config();
#ifdef ARIA_DIAG
    const int ablate = E.ablate;      // timing experiments of diagnostic builds only: results are invalid when non-zero
#else
    const int ablate = 0;
#endif
    unsigned long long* d_stamps = ctx.d_band_stamps;
    const int stamp_level = d_stamps ? E.stamp_level : -1;
    const LevelGeom& g = P.lv[l];
    const BandCfg c = band_cfg(P, g.w, l);
    const int RB = c.nb * kBandR;
    const dim3 grid((g.h + RB - 1) / RB, n_frames);
    unsigned long long* stp = (l == stamp_level && (size_t)grid.x * grid.y <= 65536) ? d_stamps : nullptr;
    BandAll ball{};
    ball.xcd_map = (E.band_xcd_map && (n_frames & 7) == 0) ? 1 : 0;
    ball.lpr[l] = c.lpr;
#define ARIA_FB_LAUNCH(T, N) ARIA_LAUNCH(prof, (k_fast_blur_band<T, N>), grid, dim3(c.nthr), c.lds, s, P, S, D.raw, D.blur, D.cand, \
                                         D.cand_cnt, D.err, l, c.nb, c.qcap, ablate, stp, D.err + 1, D.tab, \
                                         (fuse_resize && l + 1 < kLevels) ? D.raw : (uint8_t*)nullptr, ball)
    if (P.tie_mode != 0) { if (c.nb == 1) ARIA_FB_LAUNCH(1, 1); else ARIA_FB_LAUNCH(1, 0); }
    else { if (c.nb == 1) ARIA_FB_LAUNCH(0, 1); else ARIA_FB_LAUNCH(0, 0); }
#undef ARIA_FB_LAUNCH
    if (stp) {   // diagnostic: print mean phase lengths of this launch
        hipStreamSynchronize(s);
        const size_t nblk = (size_t)grid.x * grid.y;
        std::vector<unsigned long long> hs(nblk * 8);
        hipMemcpy(hs.data(), d_stamps, hs.size() * 8, hipMemcpyDeviceToHost);
        double ph[7] = {0};
        for (size_t b = 0; b < nblk; b++) for (int k = 0; k < 7; k++) ph[k] += (double)(hs[b * 8 + k + 1] - hs[b * 8 + k]);
        fprintf(stderr, "[stamps L%d] blocks %zu, cycles: stage %.0f main+pyramid %.0f barrier %.0f score %.0f map %.0f nms %.0f out %.0f\n",
                l, nblk, ph[0] / nblk, ph[1] / nblk, ph[2] / nblk, ph[3] / nblk, ph[4] / nblk, ph[5] / nblk, ph[6] / nblk);
    }
}

// Batch schedule: 8 launches in level order on one stream (or, ARIA_LEVEL_STREAMS=1, forked onto side streams).
void launch_fast_blur_band(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st,
                           Profiler* prof, bool fuse_resize, LaunchCtx& ctx) {
    const EnvConfig& E = env_config();
    const bool use_side = E.level_streams && ctx.ev_fork != nullptr;
    if (use_side) hipEventRecord(ctx.ev_fork, st);
    for (int l = 0; l < kLevels; l++) {
        hipStream_t s = (use_side && l > 0) ? ctx.side[l] : st;
        if (use_side && l > 0) hipStreamWaitEvent(s, ctx.ev_fork, 0);
        band_launch_level(P, S, D, n_frames, s, prof, fuse_resize, ctx, l);
        if (use_side && l > 0) { hipEventRecord(ctx.ev_join[l], s); hipStreamWaitEvent(st, ctx.ev_join[l], 0); }
    }
}

// Latency schedule of the single-frame path (captured into a hipGraph by orb_api.hip): with one frame a level's launch is a
// few workgroups of ~20 us each, and what the caller waits for is the CHAIN of dependent launches (8 fused ones: ~200 us).
// So the pyramid is built first -- one launch of the all-levels-in-LDS kernel k_pyramid when its bands fit, else the
// 7-launch resize chain -- and then ONE launch runs the FAST/blur strips of all eight levels side by side.
void launch_pyramid_and_band_latency(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st,
                                     Profiler* prof, LaunchCtx& ctx) {
    // Default: pyramid kernel, then ONE launch over the strips of all levels, in order on st.
    // ARIA_LATENCY_FORK=1 (experiment, measured slower: 188 vs 135 us per frame, a second branch in the graph costs more
    // in cross-queue signalling than the overlap gains): level 0 needs nothing but the source frame, so its strips run
    // on a side stream beside the pyramid kernel and levels 1..7 follow the pyramid.
    static const bool fork = [] { const char* e = aria_getenv("ARIA_LATENCY_FORK"); return e && e[0] == '1'; }();
    auto band = [&](int l_lo, int l_hi, hipStream_t s) {
        BandAll ba{};
        size_t lds = 0;
        int nthr = 64, total = 0;
        for (int l = 0; l < kLevels; l++) {
            ba.first[l] = total;
            if (l < l_lo || l >= l_hi) continue;        // no strips: blockIdx never maps to this level
            const LevelGeom& g = P.lv[l];
            // one strip per workgroup at every level (nb = 1): same code path, uniform block size = the widest level's
            const int wq = (g.w + 3) & ~3, lpr = band_lanes(P, g.w, env_config().band_trim != 0);
            ba.lpr[l] = lpr;
            const int qpct = std::min(50, P.band_qpct0 + P.band_qstep * l);
            int qcap = (int)((int64_t)kBandR * wq * qpct / 100);
            qcap = std::max(512, (qcap + 63) & ~63);
            ba.qcap[l] = qcap;
            total += (g.h + kBandR - 1) / kBandR;
            lds = std::max(lds, (size_t)(kBandR + 8) * (wq + 8) + 4 * (size_t)qcap);
            nthr = std::max(nthr, ((lpr + 63) / 64) * 64);
        }
        ba.first[kLevels] = total;
        if (total == 0) return;
        const dim3 grid(total, n_frames);
        if (P.tie_mode != 0)
            ARIA_LAUNCH(prof, (k_fast_blur_band<1, 1>), grid, dim3(nthr), lds, s, P, S, D.raw, D.blur, D.cand, D.cand_cnt, D.err, -1, 1, 0,
                        0, (unsigned long long*)nullptr, D.err + 1, D.tab, (uint8_t*)nullptr, ba);
        else
            ARIA_LAUNCH(prof, (k_fast_blur_band<0, 1>), grid, dim3(nthr), lds, s, P, S, D.raw, D.blur, D.cand, D.cand_cnt, D.err, -1, 1, 0,
                        0, (unsigned long long*)nullptr, D.err + 1, D.tab, (uint8_t*)nullptr, ba);
    };
    const bool do_fork = fork && ctx.ev_fork && ctx.side[0];
    if (do_fork) {
        // (the level-0 strips would race with the pyramid kernel's clearing: zero the counters before the fork as well)
        hipMemsetAsync(D.ovf, 0, sizeof(int) * (size_t)(4 + kLevels * n_frames), st);
        if (ctx.hdr) hipMemsetAsync(ctx.hdr, 0, 64, st);
        hipEventRecord(ctx.ev_fork, st);
        hipStreamWaitEvent(ctx.side[0], ctx.ev_fork, 0);
        band(0, 1, ctx.side[0]);
        hipEventRecord(ctx.ev_join[0], ctx.side[0]);
    }
    // the pass counters (arena + candidate counters, adjacent) and the single-frame result header are zeroed by the
    // pyramid kernel's first workgroup; by fill nodes when the fused pyramid is not available for this plan
    const int n_cnt = 4 + kLevels * n_frames;
    const bool zc = latency_zero_copy(P, ctx, prof) && n_frames == 1;
    if (!launch_pyramid_fused(P, S, D, n_frames, st, prof, D.ovf, n_cnt, ctx.hdr, ctx.hdr ? 16 : 0,
                              zc ? ctx.host_img : nullptr, zc ? const_cast<uint8_t*>(S.img) : nullptr)) {
        hipMemsetAsync(D.ovf, 0, sizeof(int) * (size_t)n_cnt, st);
        if (ctx.hdr) hipMemsetAsync(ctx.hdr, 0, 64, st);
        for (int l = 1; l < kLevels; l++) launch_pyramid_level(P, S, D, n_frames, st, prof, l);
    }
    band(do_fork ? 1 : 0, kLevels, st);
    if (do_fork) hipStreamWaitEvent(st, ctx.ev_join[0], 0);
}

}  // namespace aria
