// k_band2: FAST-9/16 + NMS + 7x7 Gaussian + bilinear step to the next pyramid level for one horizontal band (26 output
// rows x full level width) of one level of one frame per workgroup.  SURVEY.md rows a6.1, a6.2, a6.3, a6.7.
// Round-2 alternative to k_fast_blur_band (fast_blur_band.hip), selected with ARIA_FAST_BLUR_IMPL=mfma: same stages, same
// bits, but the blur -- 17 of that kernel's ~47 VALU instructions per pixel -- runs on the matrix cores:
//
//   pass 1  H = P x T_h       v_mfma_i32_32x32x32_i8. A = 32 staged rows x 16-byte pixel chunks straight from LDS (ds_read_b128,
//                             bytes XOR 0x80 so they are signed), B = the banded 7-tap Toeplitz matrix (a constant), C-in =
//                             128*257 undoes the XOR. Two instructions cover the 64 input columns a 32-column block needs.
//                             H is exact, 0..65535, column on the lane, 16 rows in the registers.
//   pass 2  out = H^T x T_v   v_mfma_f32_32x32x16_bf16. The accumulator layout of pass 1 IS the A-operand layout of pass 2
//                             (column on the lane, rows as k), so nothing moves between lanes: one v_perm per value
//                             zero-extends the low / high BYTE of two H values into a bf16 pair. A byte 0x00XX read as
//                             bf16 is XX * 2^-133 for every XX in 0..255 (subnormals continue into the first normal binade;
//                             the matrix pipe does not flush them: tools/microbench/mfma_blur_probe.hip), the Toeplitz
//                             weights are k * 2^109 (low plane) and k * 2^117 (high plane), so the f32 accumulator holds
//                             sum * 2^-24 exactly (sum < 2^25, every partial sum is a multiple of 2^-24).
//   round   v_mul_f32 by 256 and v_cvt_pk_u8_f32: round-to-nearest-even + saturate + byte insert in one instruction
//           = OpenCV's SymmColumnVec_32s8u rounding. The scalar-tail columns (x >= w & ~3, ties round UP there) are
//           recomputed in integers by a few lanes afterwards (blur_tie_mode 1).
//   Rows land on the lanes and 4 consecutive columns in the registers of pass 2's result, i.e. a packed dword is a piece of a
//   row; v_permlane32_swap pairs the two lane halves up so each lane stores 8 contiguous bytes.
//
// FAST keeps the lane walk of round 1 (a lane owns a 4-pixel column and walks down; compass reject in packed int16 on
// every pixel, survivors queued in LDS and scored densely, NMS over corners only), now without the blur's register
// window, with the east/west pairs taken by one v_perm each straight from the row's dwords, and on 26-row strips
// (34 staged rows: 31 % halo instead of 62 %).
// Integer/byte stencil work stays on the VALU; the only contraction on this path -- the separable 7x7 filter, a banded
// matrix product -- is what moved to the matrix pipe, which the extractor otherwise leaves idle.
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

constexpr int kB2R = 26;                   // output rows per strip: 26 + 6 = 32 input rows = one MFMA M-tile
constexpr int kB2Rows = kB2R + 8;          // staged rows yb-4 .. yb+R+3 of a strip (FAST ring rows need +-4)
constexpr int kB2X0 = 16;                  // LDS column of x = 0 (column c <-> x = c - 16): pixel chunks stay 16-byte aligned

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef short short2v __attribute__((ext_vector_type(2)));
typedef unsigned short us2v __attribute__((ext_vector_type(2)));

// ---- constant operand tables of the two matrix passes, one row of 24 dwords per lane ----
//   [0..3]   B1: pass-1 band matrix for input columns x0-16 .. x0+15   (lane (n, g): k = 16g .. 16g+15 as bytes)
//   [4..7]   B2: ... for input columns x0+16 .. x0+47
//   [8..15]  pass-2 weights of the LOW byte plane, k-step 0 and 1 (lane (n, g): 8 bf16 each)
//   [16..23] ... of the HIGH byte plane
struct BlurTab { uint32_t w[64][24]; };
constexpr int blur_k7(int t) { return (t < 0 || t > 6) ? 0 : (t == 0 || t == 6) ? 18 : (t == 1 || t == 5) ? 34 : (t == 2 || t == 4) ? 49 : 55; }
constexpr uint32_t bf16_bits_pow2(int k, int e) {   // bf16 bit pattern of k * 2^e, k in {0, 18, 34, 49, 55}
    if (k == 0) return 0;
    int p = 0;
    while ((k >> (p + 1)) != 0) p++;
    return (uint32_t)(((127 + e + p) << 7) | (((k << 7) >> p) & 0x7F));
}
constexpr BlurTab make_blur_tab() {
    BlurTab t{};
    for (int lane = 0; lane < 64; lane++) {
        const int n = lane & 31, g = lane >> 5;
        for (int d = 0; d < 4; d++) {
            uint32_t w1 = 0, w2 = 0;
            for (int b = 0; b < 4; b++) {
                const int k = 16 * g + 4 * d + b;
                w1 |= (uint32_t)blur_k7(k - n - 13) << (8 * b);     // tap of input column x0-16+k for output column x0+n
                w2 |= (uint32_t)blur_k7(k - n + 19) << (8 * b);     // ... of input column x0+16+k
            }
            t.w[lane][d] = w1;
            t.w[lane][4 + d] = w2;
        }
        for (int s = 0; s < 2; s++)
            for (int d = 0; d < 4; d++) {
                uint32_t lo = 0, hi = 0;
                for (int e = 0; e < 2; e++) {
                    const int j = 2 * d + e;
                    const int i = (j & 3) + 8 * (2 * s + (j >> 2)) + 4 * g;   // input row held in accumulator register 8s + j
                    const int k = blur_k7(i - n);                             // output row n <-> input rows n .. n+6
                    lo |= bf16_bits_pow2(k, 109) << (16 * e);
                    hi |= bf16_bits_pow2(k, 117) << (16 * e);
                }
                t.w[lane][8 + 4 * s + d] = lo;
                t.w[lane][16 + 4 * s + d] = hi;
            }
    }
    return t;
}
__device__ __attribute__((aligned(16))) const BlurTab kBlurTab = make_blur_tab();

struct Band2Cfg { int lpr, nthr, qcap, nblk; size_t lds; };
static Band2Cfg band2_cfg(const Plan& P, int w, int level, int w_next) {
    // Workgroup = one strip. Threads = 2 x ceil64(w / 4): the first half walks the columns (FAST compass + queue), the
    // second half runs the blur tiles on the matrix cores and the bilinear step to the next level AT THE SAME TIME on the
    // same staged pixels -- twice the waves per byte of LDS, which is what bounds this kernel's occupancy, and the two
    // halves need about the same time.
    // LDS per workgroup = staged block + survivor queue. The queue is sized by how corner-dense a level is expected to be
    // (the compass test passes ~3 % of level-0 pixels but 25-30 % at level 7 on the benchmark frames); a full queue is
    // not an error: that block takes the dense rescoring path and is counted (aria_orb_slow_path_blocks).
    const EnvConfig& E = env_config();
    const int q0 = E.band_qpct0 >= 0 ? E.band_qpct0 : P.band_qpct0, qstep = E.band_qstep >= 0 ? E.band_qstep : P.band_qstep;
    const int qpct = std::min(50, q0 + qstep * level);
    const int wq = (w + 3) & ~3, lpr = wq >> 2, nblk = (w + 31) >> 5;
    const int pitchL = 32 * nblk + 32;
    const int nthr = 2 * (((lpr + 63) / 64) * 64);
    const size_t pix = (size_t)(kB2R + 8) * pitchL;
    int qcap = (int)((int64_t)kB2R * wq * qpct / 100);
    qcap = std::max(512, (qcap + 63) & ~63);
    // + the pyramid step's tables: next level's x table and <= 48 y-table words
    return Band2Cfg{lpr, nthr, qcap, nblk, pix + 4 * (size_t)qcap + 4 * (size_t)(w_next + 48)};
}

__device__ __forceinline__ uint32_t b2_pk_min(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
}
__device__ __forceinline__ uint32_t b2_pk_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
}
__device__ __forceinline__ uint32_t b2_pk_sub(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, a) - __builtin_bit_cast(short2v, b));
}
__device__ __forceinline__ uint32_t b2_pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, a) + __builtin_bit_cast(short2v, b));
}
__device__ __forceinline__ uint32_t b2_umad24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// fast_score.cpp cornerScore<16> for the pixel at LDS address c (row pitch `pitch`): max over the 16 nine-arcs of
// min(v - ring) and of min(ring - v), minus 1; both polarities ride in one register as packed int16 (lo = v - p,
// hi = p - v). The pixel is a FAST-9 corner for threshold t iff the result is >= t.
__device__ __forceinline__ int b2_fast_score(const uint8_t* c, int pitch) {
    const uint8_t* rm3 = c - 3 * pitch; const uint8_t* rm2 = c - 2 * pitch; const uint8_t* rm1 = c - pitch;
    const uint8_t* rp1 = c + pitch; const uint8_t* rp2 = c + 2 * pitch; const uint8_t* rp3 = c + 3 * pitch;
    const uint32_t v = c[0];
    const uint32_t vhi = v << 16;
    uint32_t ring[16] = {rp3[0], rp3[1], rp2[2], rp1[3], c[3], rm1[3], rm2[2], rm3[1],
                         rm3[0], rm3[-1], rm2[-2], rm1[-3], c[-3], rp1[-3], rp2[-2], rp3[-1]};
    uint32_t P[16];
#pragma unroll
    for (int k = 0; k < 16; k++) P[k] = b2_pk_sub(v | (ring[k] << 16), ring[k] | vhi);
    uint32_t M2[16], M4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) M2[k] = b2_pk_min(P[k], P[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) M4[k] = b2_pk_min(M2[k], M2[(k + 2) & 15]);
    uint32_t Q = 0x80008000u;   // (-32768, -32768)
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t m9 = b2_pk_min(b2_pk_min(M4[k], M4[(k + 4) & 15]), P[(k + 8) & 15]);
        Q = b2_pk_max(Q, m9);
    }
    const int q0 = (int)(short)(Q & 0xFFFFu), q1 = (int)(short)(Q >> 16);
    return max(q0, q1) - 1;
}

template <int TIE_EVEN>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6))) void k_band2(Plan P, FrameSrc S, const uint8_t* __restrict__ raw,
                                               uint8_t* __restrict__ blur, uint32_t* __restrict__ cand,
                                               int* __restrict__ cand_cnt, int* __restrict__ err, int l,
                                               int qcap, int* __restrict__ slow_blocks,
                                               const uint32_t* __restrict__ tab, uint8_t* __restrict__ raw_next,
                                               unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_qn, s_ovf;
    // diagnostic only (ARIA_STAMPS=<level>): phase boundaries of wave 0 of every workgroup, s_memtime ticks
#define STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x)) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP_H(k) do { if (stamps && threadIdx.x == (blockDim.x >> 1)) stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x)) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    STAMP(0);
    if (stamps && threadIdx.x == 0) stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x)) * 16 + 14] = __builtin_amdgcn_s_memrealtime();

    const LevelGeom g = P.lv[l];
    const int w = g.w, h = g.h;
    const int wq = (w + 3) & ~3;
    const int lpr = wq >> 2;                             // lanes per row of the FAST walk
    const int nblk = (w + 31) >> 5;                      // 32-column blocks of the blur
    const int pitchL = 32 * nblk + 32;                   // LDS pitch of the staged block: column c <-> x = c - 16
    const int sp = wq + 4;                               // score-map pitch: column 0 <-> x = -1
    constexpr int RB = kB2R;                             // output rows of this workgroup
    const int rowsL = RB + 8;                            // staged rows y0-4 .. y0+RB+3
    uint8_t* s_pix = smem;
    uint8_t* s_map = smem;                               // score map, aliases s_pix once the pixels are dead
    uint32_t* s_queue = reinterpret_cast<uint32_t*>(smem + rowsL * pitchL);

    const int tid = threadIdx.x, nthr = blockDim.x;
    const int nthrW = nthr >> 1;                         // walk threads 0 .. nthrW-1; helpers nthrW .. nthr-1
    const bool helper = tid >= nthrW;
    const int htid = tid - nthrW;
    const int frame = blockIdx.y;
    const int y0 = blockIdx.x * RB;
    int pitch;
    const uint8_t* img = raw_level_ptr(P, S, raw, frame, l, pitch);

    if (tid == 0) { s_qn = 0; s_ovf = 0; }
    // resize tables of the fused pyramid step (next level's x table, the y-table words of the rows this band can own):
    // fetched now, so their global round trips hide behind the staging loads instead of stalling the pyramid step
    uint32_t* s_tab = s_queue + qcap;
    int pyr_dlo = 0, pyr_dhi = 0;
    if (raw_next != nullptr) {
        const LevelGeom gn = P.lv[l + 1];
        const int y_end = min(y0 + RB, h);
        pyr_dlo = max((int)(((int64_t)y0 * gn.h) / h) - 2, 0);
        pyr_dhi = min((int)(((int64_t)y_end * gn.h + h - 1) / h) + 2, gn.h);
        const uint32_t* xt = tab + gn.xtab;
        const uint32_t* yt = tab + gn.ytab;
        for (int i = tid; i < gn.w; i += nthr) s_tab[i] = xt[i];
        if (tid < pyr_dhi - pyr_dlo) s_tab[gn.w + tid] = yt[pyr_dlo + tid];
    }
    // ---- stage rows y0-4 .. y0+RB+3, columns -4 .. wq+3, BORDER_REFLECT_101 outside the level. Interior: 16-byte global
    //      loads, four in flight per lane before the first LDS write, 16-byte LDS stores (x = 0 sits on a 16-byte
    //      boundary); edges (x < 0, x >= 16*floor(w/16)): per-byte reflected loads. Columns further out are never
    //      initialised: the blur's band matrices are zero there, the walk never looks. ----
    const int nch = ((l > 0) || S.aligned16) ? (w >> 4) : 0;     // full 16-byte chunks per row
    if (nch > 0) {
        const int rpp = nthr / nch;                               // rows staged per pass (nthr >= w/4 > nch)
        const int my_r = tid / nch, my_c = tid - my_r * nch;
        if (my_r < rpp) {
            // rows past the end are clamped to the last row (it is then loaded and stored twice: same bytes, same place),
            // so the four loads and stores are unconditional and stay in registers
            for (int r0 = my_r; r0 < rowsL; r0 += 4 * rpp) {
                const int ra = r0, rb = min(r0 + rpp, rowsL - 1), rc = min(r0 + 2 * rpp, rowsL - 1), rd = min(r0 + 3 * rpp, rowsL - 1);
                const uint4 va = *reinterpret_cast<const uint4*>(img + (int64_t)reflect101(y0 - 4 + ra, h) * pitch + 16 * my_c);
                const uint4 vb = *reinterpret_cast<const uint4*>(img + (int64_t)reflect101(y0 - 4 + rb, h) * pitch + 16 * my_c);
                const uint4 vc = *reinterpret_cast<const uint4*>(img + (int64_t)reflect101(y0 - 4 + rc, h) * pitch + 16 * my_c);
                const uint4 vd = *reinterpret_cast<const uint4*>(img + (int64_t)reflect101(y0 - 4 + rd, h) * pitch + 16 * my_c);
                *reinterpret_cast<uint4*>(s_pix + ra * pitchL + kB2X0 + 16 * my_c) = va;
                *reinterpret_cast<uint4*>(s_pix + rb * pitchL + kB2X0 + 16 * my_c) = vb;
                *reinterpret_cast<uint4*>(s_pix + rc * pitchL + kB2X0 + 16 * my_c) = vc;
                *reinterpret_cast<uint4*>(s_pix + rd * pitchL + kB2X0 + 16 * my_c) = vd;
            }
        }
    }
    {
        const int xe0 = nch * 16;                                  // first column not covered by full chunks
        const int ne = 1 + ((wq + 4 - xe0) >> 2);                  // edge dwords per row: x = -4 and x >= xe0
        for (int i = tid; i < rowsL * ne; i += nthr) {
            const int r = i / ne, e = i - r * ne;
            const int gx = e == 0 ? -4 : xe0 + 4 * (e - 1);
            const int gy = reflect101(y0 - 4 + r, h);
            const uint8_t* rowp = img + (int64_t)gy * pitch;
            const uint32_t v = (uint32_t)rowp[reflect101(gx, w)] | ((uint32_t)rowp[reflect101(gx + 1, w)] << 8) |
                               ((uint32_t)rowp[reflect101(gx + 2, w)] << 16) | ((uint32_t)rowp[reflect101(gx + 3, w)] << 24);
            *reinterpret_cast<uint32_t*>(s_pix + r * pitchL + (gx + kB2X0)) = v;
        }
    }
    __syncthreads();
    STAMP(1);

    // keypoints.cpp runByImageBorder keeps x in [31, w-31), y in [31, h-31); FAST is needed on that region + 1 ring
    const bool level_has_kp = (w > 2 * kEdgeThreshold) && (h > 2 * kEdgeThreshold);
    const int fx0 = kEdgeThreshold - 1, fx1 = w - kEdgeThreshold;        // inclusive FAST ranges
    const int fy0 = kEdgeThreshold - 1, fy1 = h - kEdgeThreshold;
    const int thr = P.fast_threshold;
    const int dpr = pitchL >> 2;

    if (helper) {
    // =========================== helper half: the blur on the matrix cores, one wave per 32-column tile ==================
    {
        const int lane = htid & 63, wv = htid >> 6, nwv = nthrW >> 6;
        const int r = lane & 31, gh = lane >> 5;
        const uint4* tp = reinterpret_cast<const uint4*>(&kBlurTab.w[lane][0]);
        const uint4 t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3], t4 = tp[4], t5 = tp[5];
        const v4i B1 = {(int)t0.x, (int)t0.y, (int)t0.z, (int)t0.w}, B2 = {(int)t1.x, (int)t1.y, (int)t1.z, (int)t1.w};
        const v4i Wl0 = {(int)t2.x, (int)t2.y, (int)t2.z, (int)t2.w}, Wl1 = {(int)t3.x, (int)t3.y, (int)t3.z, (int)t3.w};
        const v4i Wh0 = {(int)t4.x, (int)t4.y, (int)t4.z, (int)t4.w}, Wh1 = {(int)t5.x, (int)t5.y, (int)t5.z, (int)t5.w};
        // The XOR of the pixels is undone by one more instruction on constant operands instead of a constant C-in tuple
        // (which would have to be re-materialised per tile, 16 moves): sum over k of 64 * KC[k] = 64 * 514 = 128 * 257.
        const v4i AC = {0x40404040, 0x40404040, 0x40404040, 0x40404040};
        const v4i KC = {0x10101010, 0x10101010, 0x10101010, gh ? 0x12101010 : 0x10101010};
        v16i Z;
#pragma unroll
        for (int i = 0; i < 16; i++) Z[i] = 0;
        uint8_t* bl = blur + (int64_t)frame * P.blur_frame_bytes + g.blur_off;
        const int Y = y0 + r;
        const bool row_ok = r < kB2R && Y < h;
        for (int cb = wv; cb < nblk; cb += nwv) {
            // input row i = 0..31 of the M-tile <-> level row y0 - 3 + i <-> staged row 1 + i
            const uint8_t* rowp = s_pix + (1 + r) * pitchL + 32 * cb + 16 * gh;
            v4i a0 = *reinterpret_cast<const v4i*>(rowp);
            v4i a1 = *reinterpret_cast<const v4i*>(rowp + 32);
#pragma unroll
            for (int i = 0; i < 4; i++) { a0[i] ^= (int)0x80808080; a1[i] ^= (int)0x80808080; }
            v16i H = __builtin_amdgcn_mfma_i32_32x32x32_i8(AC, KC, Z, 0, 0, 0);
            H = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, B1, H, 0, 0, 0);
            H = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, B2, H, 0, 0, 0);
            v4i Al0, Al1, Ah0, Ah1;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                Al0[d] = (int)__builtin_amdgcn_perm((uint32_t)H[2 * d + 1], (uint32_t)H[2 * d], 0x0c040c00u);
                Ah0[d] = (int)__builtin_amdgcn_perm((uint32_t)H[2 * d + 1], (uint32_t)H[2 * d], 0x0c050c01u);
                Al1[d] = (int)__builtin_amdgcn_perm((uint32_t)H[8 + 2 * d + 1], (uint32_t)H[8 + 2 * d], 0x0c040c00u);
                Ah1[d] = (int)__builtin_amdgcn_perm((uint32_t)H[8 + 2 * d + 1], (uint32_t)H[8 + 2 * d], 0x0c050c01u);
            }
            v16f O;
#pragma unroll
            for (int i = 0; i < 16; i++) O[i] = 0.f;
            O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, Ah0), __builtin_bit_cast(v8bf, Wh0), O, 0, 0, 0);
            O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, Ah1), __builtin_bit_cast(v8bf, Wh1), O, 0, 0, 0);
            O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, Al0), __builtin_bit_cast(v8bf, Wl0), O, 0, 0, 0);
            O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, Al1), __builtin_bit_cast(v8bf, Wl1), O, 0, 0, 0);
            // lane (output row r, half gh): registers 4q .. 4q+3 = columns 8q + 4gh + (0..3)
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t v = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) v = __builtin_amdgcn_cvt_pk_u8_f32(O[4 * q + e] * 256.0f, e, v);
                o[q] = v;
            }
            // pair the halves up: afterwards lane (r, gh) holds columns 16p + 8gh .. +7 in (s[p][0], s[p][1])
            const auto s0 = __builtin_amdgcn_permlane32_swap(o[0], o[1], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(o[2], o[3], false, false);
            if (row_ok) {
                uint8_t* dst = bl + (int64_t)Y * g.pitch + 32 * cb + 8 * gh;
                if (32 * cb + 8 * gh < g.pitch) *reinterpret_cast<uint2*>(dst) = make_uint2(s0[0], s0[1]);
                if (32 * cb + 16 + 8 * gh < g.pitch) *reinterpret_cast<uint2*>(dst + 16) = make_uint2(s1[0], s1[1]);
            }
        }
    }
    STAMP_H(12);
    // ---- a6.1 fused: this workgroup also produces the rows of level l+1 whose upper source row it owns, from the
    //      level-l rows it has staged (the bilinear resize needs rows oy, oy+1 <= y0 + RB, inside the halo):
    //      resize.cpp INTER_LINEAR_EXACT in v_perm + v_dot2_u32_u16. A lane keeps the x-table words of its output
    //      dword in registers and walks the band's ~RB/1.2 output rows; oy is monotone in dy, so the candidate rows
    //      are bracketed arithmetically and the y table decides ownership exactly. Helper thread -> (column group of 4
    //      output pixels, row interleave). ----
    if (raw_next != nullptr) {
        const LevelGeom gn = P.lv[l + 1];
        const int groups_n = gn.pitch >> 2;
        const int lprH = min(groups_n, nthrW), nbH = nthrW / lprH;      // column groups x row-interleave ways
        const int sbH = htid / lprH, liH = htid - sbH * lprH;
        if (sbH < nbH) {
            uint8_t* dstl = raw_next + (int64_t)frame * P.raw_frame_bytes + gn.raw_off;
            const uint32_t* xt = s_tab;                       // the next level's x table, copied to LDS while staging
            const uint32_t* yt = s_tab + gn.w - pyr_dlo;      // ... and the y-table words of rows pyr_dlo .. pyr_dhi-1
            const int y_end = min(y0 + RB, h);
            const int dlo = pyr_dlo, dhi = pyr_dhi;
            constexpr uint32_t put[4] = {0x03020106u, 0x03020600u, 0x03060100u, 0x06020100u};   // byte 2 of v -> byte i
            for (int gx = liH; gx < groups_n; gx += lprH) {
                const int dx0 = gx * 4;
                uint32_t xw[4], xo[4], xs[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t t = xt[min(dx0 + i, gn.w - 1)];
                    const uint32_t ox = t & 0xFFFFu, cx1 = t >> 16;
                    xw[i] = (256u - cx1) | (cx1 << 16);
                    xo[i] = (ox & ~3u) + kB2X0;
                    xs[i] = 0x0C010C00u + (ox & 3u) * 0x00010001u;
                }
                // y-table words of a batch of candidate rows are fetched together (one global round trip per 16 rows)
                for (int d0 = dlo + sbH; d0 < dhi; d0 += 16 * nbH) {
                uint32_t tyv[16];
#pragma unroll
                for (int k = 0; k < 16; k++) tyv[k] = yt[min(d0 + k * nbH, dhi - 1)];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int dy = d0 + k * nbH;
                    const uint32_t ty = tyv[k];
                    const int oy = ty & 0xFFFF;
                    if (dy >= dhi || oy < y0 || oy >= y_end) continue;  // past the bracket / another workgroup's row
                    uint32_t outw = 0;
                    if (dx0 < gn.w) {
                        const uint32_t cy1 = ty >> 16, cyp = (256u - cy1) | (cy1 << 16);
                        const uint8_t* rowa = s_pix + (oy - y0 + 4) * pitchL;
                        const uint8_t* rowb = s_pix + (min(oy + 1, h - 1) - y0 + 4) * pitchL;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const uint32_t* qa = reinterpret_cast<const uint32_t*>(rowa + xo[i]);
                            const uint32_t* qb = reinterpret_cast<const uint32_t*>(rowb + xo[i]);
                            const uint32_t top = __builtin_amdgcn_perm(qa[1], qa[0], xs[i]);     // p00 | p01 << 16
                            const uint32_t bot = __builtin_amdgcn_perm(qb[1], qb[0], xs[i]);
                            const us2v wx = __builtin_bit_cast(us2v, xw[i]);
                            const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, top), wx, 0u, false);
                            const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, bot), wx, 0u, false);
                            const uint32_t v = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, h0 | (h1 << 16)),
                                                                      __builtin_bit_cast(us2v, cyp), 32768u, false);   // < 2^24
                            outw = __builtin_amdgcn_perm(v, outw, put[i]);
                        }
                    }
                    *reinterpret_cast<uint32_t*>(dstl + (int64_t)dy * gn.pitch + dx0) = outw;
                }
                }
            }
        }
    }
    STAMP_H(13);
    } else {
    // =========================== walk half: compass reject on every pixel, survivors into the LDS queue ==================
    const int li = tid;                                   // column group
    if (tid < lpr) {
        const int x = li * 4;
        const int yb = y0;                                // first output row of this strip
        // per-lane packed masks of the pixels inside the FAST x range (bit 15: px 0/2, bit 31: px 1/3 of a pair)
        uint32_t xm0 = 0, xm1 = 0;
        if (level_has_kp) {
            if (x + 0 >= fx0 && x + 0 <= fx1) xm0 |= 0x00008000u;
            if (x + 1 >= fx0 && x + 1 <= fx1) xm0 |= 0x80000000u;
            if (x + 2 >= fx0 && x + 2 <= fx1) xm1 |= 0x00008000u;
            if (x + 3 >= fx0 && x + 3 <= fx1) xm1 |= 0x80000000u;
        }
        const uint32_t T2 = (uint32_t)thr * 0x00010001u;

        // window of the walk, slot = step % 7 (static after unrolling by 7): centre pairs of 7 rows, east/west pairs
        uint32_t RC2[7][2], EW[7][4];
#pragma unroll
        for (int u = 0; u < 7; u++) { RC2[u][0] = RC2[u][1] = 0; EW[u][0] = EW[u][1] = EW[u][2] = EW[u][3] = 0; }

        const uint32_t* lrow = reinterpret_cast<const uint32_t*>(s_pix) + li + (kB2X0 >> 2);
        if (xm0 | xm1) {     // lanes entirely outside the FAST columns have nothing to do here
        for (int tb = 0; tb < kB2Rows; tb += 7) {
            uint32_t accw = 0;     // survivors of this group of 7 steps: step u, px j -> bit (j&1 ? 31 : 15) - (j>>1) - 2u
#pragma unroll
            for (int u = 0; u < 7; u++) {
                const int t = tb + u;             // staged row of the strip; level row = yb - 4 + t
                if (t < kB2Rows) {
                    const uint32_t* rp = lrow + t * dpr;
                    const uint32_t w0 = rp[-1], w1 = rp[0], w2 = rp[1];
                    RC2[u][0] = __builtin_amdgcn_perm(0u, w1, 0x0c010c00u);   // px 0, 1 as int16 pair
                    RC2[u][1] = __builtin_amdgcn_perm(0u, w1, 0x0c030c02u);   // px 2, 3
                    EW[u][0] = __builtin_amdgcn_perm(w1, w0, 0x0c020c01u);    // west of (0, 1): px -3, -2
                    EW[u][1] = __builtin_amdgcn_perm(w1, w0, 0x0c040c03u);    // west of (2, 3): px -1, 0
                    EW[u][2] = __builtin_amdgcn_perm(w2, w1, 0x0c040c03u);    // east of (0, 1): px 3, 4
                    EW[u][3] = __builtin_amdgcn_perm(w2, w1, 0x0c060c05u);    // east of (2, 3): px 5, 6

                    const int o = t - 7;              // strip row whose rows o-3, o, o+3 are now in the window
                    const int Y = yb + o;
                    // row o+d of the window lives in slot (u + 4 + d) % 7
                    const int sC = (u + 4) % 7, sM3 = (u + 1) % 7, sP3 = u;
                    // rows o = -1 and o = R are the ring rows the NMS of the strip's first / last row looks at
                    if (o >= -1 && o <= kB2R && Y >= fy0 && Y <= fy1) {
                        // compass reject, two pixels per packed-int16 op: survive iff (N|S)&(E|W) are all darker than
                        // c - t or all brighter than c + t (every 9-arc holds one pixel of each antipodal compass pair)
                        // darker: max(min(n, s), min(e, w)) < c - t; brighter: min(max(n, s), max(e, w)) > c + t
                        uint32_t pass[2];
#pragma unroll
                        for (int pr = 0; pr < 2; pr++) {
                            const uint32_t c2 = RC2[sC][pr], n2 = RC2[sM3][pr], s2 = RC2[sP3][pr];
                            const uint32_t w2p = EW[sC][pr], e2 = EW[sC][2 + pr];
                            const uint32_t lo = b2_pk_sub(c2, T2), hi = b2_pk_add(c2, T2);
                            // sign bit set <=> darker than c - t / brighter than c + t
                            const uint32_t dk = b2_pk_sub(b2_pk_max(b2_pk_min(n2, s2), b2_pk_min(e2, w2p)), lo);
                            const uint32_t br = b2_pk_sub(hi, b2_pk_min(b2_pk_max(n2, s2), b2_pk_max(e2, w2p)));
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
                    const uint32_t ro = (uint32_t)(tb + u - 6);    // block score row of strip row o = t-7: o+1
                    if (base < qcap) s_queue[base] = (uint32_t)(x + px) | (ro << 11);
                    else s_ovf = 1;    // queue full (pathological image): the block is rescored densely below
                    base++;
                }
            }
        }
        }
    }
    STAMP(2);
    }
    __syncthreads();     // (waits for this wave's global stores too: the tail fix-up below overwrites blurred bytes)

    // ---- blur_tie_mode 1: OpenCV's SIMD column filter covers x < (w & ~3) (ties to even, what the matrix path computed);
    //      its scalar tail, the last w mod 4 columns, rounds ties UP. Those <= 3 columns per row are recomputed here in
    //      integers from the staged pixels (filter.simd.hpp SymmColumnFilter) and overwrite the bytes stored above. ----
    STAMP(5);
    if (TIE_EVEN && (w & 3)) {
        const int nt = w & 3, xb = w & ~3;
        uint8_t* bl = blur + (int64_t)frame * P.blur_frame_bytes + g.blur_off;
        for (int i = tid; i < RB * nt; i += nthr) {
            const int ro = i / nt, x = xb + (i - ro * nt);
            const int Y = y0 + ro;
            if (Y >= h) continue;
            uint32_t acc = 0;
#pragma unroll
            for (int dy = 0; dy < 7; dy++) {
                const uint8_t* p = s_pix + (ro + 1 + dy) * pitchL + (x + kB2X0);
                const uint32_t hs = 18u * (p[-3] + p[3]) + 34u * (p[-2] + p[2]) + 49u * (p[-1] + p[1]) + 55u * p[0];
                acc += (uint32_t)blur_k7(dy) * hs;
            }
            bl[(int64_t)Y * g.pitch + x] = (uint8_t)min((acc + 0x8000u) >> 16, 255u);
        }
    }

    STAMP(6);
    uint32_t* clist = cand + (int64_t)frame * P.cand_frame_entries + g.cand_off;
    int* ccnt = cand_cnt + frame * kLevels + l;
    const int lane = tid & 63;
    const bool ovf = s_ovf != 0;
    if (ovf && tid == 0) atomicAdd(slow_blocks, 1);
    const int qn = ovf ? 0 : s_qn;

    // ---- dense scoring of the queued survivors (all lanes busy); score >= t <=> FAST-9 corner. Two entries per lane
    //      and iteration: the 2 x 17 LDS gathers overlap. Corners are compacted to the front part of the queue memory
    //      (s_corner) so the later passes only visit corners. ----
    __shared__ int s_cn;
    if (tid == 0) s_cn = 0;
    __syncthreads();
    uint32_t* s_corner = s_queue;          // in-place compaction is safe: a corner's slot index is <= entries consumed
    // (each iteration reads its entries into registers before any thread of the block writes: see the barrier below)
    for (int i0 = 0; i0 < qn; i0 += 2 * nthr) {
        const int ia = i0 + tid, ib = i0 + nthr + tid;
        const uint32_t ea = ia < qn ? s_queue[ia] : 0u, eb = ib < qn ? s_queue[ib] : 0u;
        int sa = 0, sb2 = 0;
        if (ia < qn) sa = b2_fast_score(s_pix + (((ea >> 11) & 0x1FFF) + 3) * pitchL + ((ea & 0x7FF) + kB2X0), pitchL);
        if (ib < qn) sb2 = b2_fast_score(s_pix + (((eb >> 11) & 0x1FFF) + 3) * pitchL + ((eb & 0x7FF) + kB2X0), pitchL);
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
    STAMP(7);
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
    STAMP(8);
    const int n_items = ovf ? RB * wq : cn;
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
                    const uint8_t* c = s_pix + (ro + 3) * pitchL + (X + kB2X0);
                    sc = b2_fast_score(c, pitchL);
                    if (sc >= thr) {
                        keep = true;
                        for (int dy = -1; dy <= 1 && keep; dy++)
                            for (int dx = -1; dx <= 1; dx++) {
                                if (dx == 0 && dy == 0) continue;
                                int ns = b2_fast_score(c + dy * pitchL + dx, pitchL);
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
    STAMP(9);
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
    STAMP(10);
    if (stamps && threadIdx.x == 0) stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x)) * 16 + 15] = __builtin_amdgcn_s_memrealtime();
#undef STAMP
#undef STAMP_H
}

int band2_set_attributes() {
    const int lds_max = 160 * 1024 - 64;
    ARIA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_band2<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    ARIA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_band2<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    return ARIA_OK;
}

// 8 launches, one per level, in level order on one stream (launch l writes the raw rows of level l+1 that launch l+1 reads)
void launch_band2(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st, Profiler* prof,
                  bool fuse_resize, LaunchCtx& ctx) {
    const int stamp_level = ctx.d_band_stamps ? env_config().stamp_level : -1;
    for (int l = 0; l < kLevels; l++) {
        const LevelGeom& g = P.lv[l];
        const Band2Cfg c = band2_cfg(P, g.w, l, l + 1 < kLevels ? P.lv[l + 1].w : 0);
        const dim3 grid((g.h + kB2R - 1) / kB2R, n_frames);
        unsigned long long* stp = (l == stamp_level && (size_t)grid.x * grid.y <= 32768) ? ctx.d_band_stamps : nullptr;
#define ARIA_B2_LAUNCH(T) ARIA_LAUNCH(prof, (k_band2<T>), grid, dim3(c.nthr), c.lds, st, P, S, D.raw, D.blur, D.cand, \
                                      D.cand_cnt, D.err, l, c.qcap, D.err + 1, D.tab, \
                                      (fuse_resize && l + 1 < kLevels) ? D.raw : (uint8_t*)nullptr, stp)
        if (P.tie_mode == 1) ARIA_B2_LAUNCH(1); else ARIA_B2_LAUNCH(0);
#undef ARIA_B2_LAUNCH
        if (stp) {   // diagnostic: print mean phase lengths of this launch
            hipStreamSynchronize(st);
            const size_t nblk = (size_t)grid.x * grid.y;
            std::vector<unsigned long long> hs(nblk * 16);
            hipMemcpy(hs.data(), stp, hs.size() * 8, hipMemcpyDeviceToHost);
            auto mean = [&](int a, int b) { double t = 0; for (size_t i = 0; i < nblk; i++) t += (double)(hs[i * 16 + b] - hs[i * 16 + a]); return t / nblk; };
            fprintf(stderr, "[band2 stamps L%d] blocks %zu (%d thr, lds %zu), cycles: stage %.0f | walk %.0f || blur %.0f pyramid %.0f | "
                    "barrier(after walk) %.0f tail %.0f score %.0f map %.0f nms %.0f out %.0f | total %.0f\n", l, nblk, c.nthr, c.lds,
                    mean(0, 1), mean(1, 2), mean(1, 12), mean(12, 13), mean(2, 5), mean(5, 6), mean(6, 7), mean(7, 8), mean(8, 9),
                    mean(9, 10), mean(0, 10));
            {   // effective shader clock while this launch ran: s_memtime ticks per 100 MHz s_memrealtime tick; and how many
                // workgroups were in flight at once (sum of lifetimes / span of the launch / CUs)
                unsigned long long t_lo = ~0ull, t_hi = 0; double life = 0;
                for (size_t i = 0; i < nblk; i++) { t_lo = std::min(t_lo, hs[i * 16 + 14]); t_hi = std::max(t_hi, hs[i * 16 + 15]); life += (double)(hs[i * 16 + 15] - hs[i * 16 + 14]); }
                fprintf(stderr, "[band2 stamps L%d] shader clock %.0f MHz, launch span %.1f us, mean workgroups in flight per CU %.2f\n", l,
                        100.0 * mean(0, 10) / mean(14, 15), (double)(t_hi - t_lo) / 100.0, life / (double)(t_hi - t_lo) / 256.0);
            }
        }
    }
}

}  // namespace aria
