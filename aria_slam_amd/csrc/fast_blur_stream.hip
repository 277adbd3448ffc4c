// k_fast_blur_stream: FAST-9/16 + NMS + 7x7 Gaussian + the bilinear step to the next pyramid level for the batch path,
// as a STREAMING kernel: one wave walks a 256-pixel-wide column panel of one pyramid level from the first row to the last.
// SURVEY.md rows a6.1 (pyramid), a6.2 (FAST), a6.3 (border filter), a6.7 (blur). Same bits as k_fast_blur_band.
//
// Why this shape (MI355X), and what it removes from k_fast_blur_band (fast_blur_band.hip, still the kernel of the
// single-frame latency schedule):
//  * a band workgroup staged 21 rows to produce 13 (1.6x halo recompute of the row pass and the FAST reject) and ran a
//    chain of dependent phases separated by workgroup barriers (stage -> walk -> score -> map -> NMS -> out) at 2-3
//    workgroups per CU: the VALU sat idle a quarter of the time. Here a wave owns its panel for the whole height, every
//    row is loaded once (one dword per lane, requested a group of seven rows ahead), passes through the 7-row register
//    window once, and there is NO workgroup barrier at all: the waves of a workgroup share nothing but the LDS allocation.
//  * lanes are not tied to a frame: the rows of all frames of the launch are laid side by side as one virtual row
//    [halo | w/4 dwords | halo] x frames, and wave k takes virtual dwords 62 k - 1 .. 62 k + 62 (one dword of overlap on
//    each side provides the neighbours' pixels). A 179-px level fills its waves as well as a 1408-px one: 93-96 % of the
//    lanes carry pixels at every level (band kernel: 73 % on average, 52 % at the 257-px level).
//  * the wave keeps the last 16 raw rows of its panel in a private LDS ring (4 KB): neighbour dwords of a new row,
//    the 16-pixel rings of FAST survivors and the source pixels of the pyramid step are read from there. FAST survivors
//    (4-point compass reject on every pixel, as before) are queued in LDS and scored densely once per group of seven
//    rows; corner scores go to a 16-row score ring for the 3x3 non-maximum suppression, which visits corners only.
//  * REFLECT_101 needs no edge code: a halo lane, and the lane of a partial last dword, load ONE (unaligned) dword from
//    inside the row and a per-lane v_perm selector puts the mirrored bytes in place; everybody else's selector is the
//    identity.
//  * blur_tie_mode (OpenCV's SIMD column filter rounds ties to even, its scalar tail rounds them up) is a per-pixel
//    flag of the lane, applied in the walk (tail pixels: floor(acc + 0.5) by v_fract + v_fma before the same
//    v_cvt_pk_u8_f32); waves without a tail pixel take the plain instruction stream.
// Integer/byte work only: no MFMA. HBM traffic is the algorithmic minimum by construction: every level byte is read
// once, the blurred level and the next raw level are written once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "orb_device.h"
#include "orb_kernels.h"

namespace aria {

namespace {

constexpr int kRing = 16;                         // rows of the raw ring and of the score ring (power of two)
// ARIA_STREAM_PEND_LDS (compact lists only): the records of the pending candidate batch wait for their slice where they are, at
// the front of the out-list, instead of in a register: nothing is appended between the reservation (end of the NMS phase) and
// the store (behind the next rows' take-out), so the batch is read again and dropped there.
#ifndef ARIA_STREAM_PEND_LDS
#define ARIA_STREAM_PEND_LDS 1
#endif
constexpr int kQ1 = 256;                          // survivor queue entries (appended in rounds of at most this many)
constexpr int kQ2 = 512;                          // corner list entries (beyond: dense scan of the score ring)
// Compact lists (round 4): 16-bit survivor and corner entries, 4-byte candidate records, no pads -- 8192 + 2048 = 10 240 B per
// wave: what a fourth wave per SIMD needs from the LDS side (ARIA_STREAM_WAVES4 below is the register side).
#ifndef ARIA_STREAM_COMPACT_LDS
#define ARIA_STREAM_COMPACT_LDS 1
#endif
#if ARIA_STREAM_COMPACT_LDS
constexpr int kHdrBytes = 0;                      // (no read of the rings goes below byte 0 of a row: scored pixels start at local px 3)
#else
constexpr int kHdrBytes = 64;                     // in front of the raw ring: lane 0's left-neighbour read lands here
#endif
constexpr int kRawBytes = kRing * 256;
constexpr int kMapBytes = kRing * 256;
constexpr int kOut = 128;                         // candidate out-list entries (record + frame), flushed 64 at a time
#if ARIA_STREAM_COMPACT_LDS
constexpr int kMapPad = 0;
constexpr int kWaveLds = kHdrBytes + kRawBytes + kMapBytes + kMapPad + 2 * kQ1 + 2 * kQ2 + 4 * kOut;      // 10 240 B per wave
static_assert(kWaveLds == 10240, "16 waves per CU");
typedef uint16_t qent_t;
#else
constexpr int kMapPad = 64;                       // behind the score ring (its last row's right-neighbour read)
constexpr int kWaveLds = kHdrBytes + kRawBytes + kMapBytes + kMapPad + 4 * kQ1 + 4 * kQ2 + 8 * kOut;      // 12.3 KB per wave
typedef uint32_t qent_t;
#endif
static_assert(kOut >= 128 && kQ1 >= 64 && kQ2 >= 64, "emit() appends up to 64 entries behind a 64-entry batch; the lists are filled 64 lanes at a time");
constexpr int kOwned = 62;                        // productive lanes per wave (lanes 1..62)

typedef short short2v __attribute__((ext_vector_type(2)));
typedef unsigned short us2v __attribute__((ext_vector_type(2)));
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// (gload_sv / gstore_sv: wave-uniform 64-bit base in SGPRs + the lane's 32-bit offset -- orb_device.h)
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

// inclusive prefix sum over the 64 lanes of the wave (DPP: Hillis-Steele inside the 16-lane rows, then the row carries)
__device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);     // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);     // row_bcast:31 -> rows 2, 3
    return v;
}

// Lanes of a wave hand data to each other through LDS without a workgroup barrier: the hardware executes a wave's LDS
// instructions in order, this keeps the COMPILER from moving a lane's reads above another lane's writes.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// byte offset of level row y inside a ring of kRing rows of 256 bytes (row -3 is slot 0)
__device__ __forceinline__ int ring_off(int y) { return ((y + 3) & (kRing - 1)) << 8; }

// Tuning switches of this file (A/B builds only; the defaults are the product): see DESIGN.md section 4, round 4.
#ifndef ARIA_SCORE_F16
#define ARIA_SCORE_F16 1
#endif
#ifndef ARIA_COMPASS2
#define ARIA_COMPASS2 1
#endif
// Rows per group of the wave's walk (the unroll factor of its loop). 7 = the window's height (round 3). 8 (round 4, measured
// and NOT adopted): a group is then two whole row quads of the Q4-ordered blurred level, a lane collects the four dwords of
// a quad in registers under STATIC indices and stores them with one 16-byte instruction (64 lanes x 16 contiguous bytes
// instead of four instructions of 64 x 4 bytes at a 16-byte stride), and the per-group phases run once per 8 rows instead
// of once per 7 -- 159 VGPRs, bit-identical (the 56 stream / parity tests), and no faster: 0.1992 / 0.2007 of the roofline
// against 0.1995 / 0.2025 for 7 (A B A B, one box). Neither the store shape nor the per-group overhead is what bounds the
// kernel; its vector-instruction count is (profiles/r4_mfma_blur_arithmetic.md).
#ifndef ARIA_STREAM_G
#define ARIA_STREAM_G 7
#endif
// Flat walk: the three per-row range tests of the walk (row ingested / row in the segment / row in the FAST range) select
// a predicate instead of branching, so that a group's seven rows are one basic block.
#ifndef ARIA_STREAM_FLAT
#define ARIA_STREAM_FLAT 1
#endif
// Pyramid step IN the walk (round 4): a source row's horizontal blend is taken from the row's dword and its two neighbours while
// they are in registers (the walk has them for the blur), and blended with the previous row's into the output row whose lower
// source row it is (table by source row, orb_plan.cpp) -- no second pass over the LDS ring, no y-table window, no row loop.
#ifndef ARIA_STREAM_PYR_INWALK
#define ARIA_STREAM_PYR_INWALK 1
#endif
#ifndef ARIA_STREAM_SPLIT
#define ARIA_STREAM_SPLIT 0
#endif
constexpr int kG = ARIA_STREAM_G;
static_assert(!ARIA_STREAM_FLAT || ARIA_COMPASS2, "the flat walk masks rows through the one-comparison compass test");
static_assert(kG == 7 || kG == 8, "the window holds 7 rows; 8 makes groups whole row quads");

// fast_score.cpp cornerScore<16> for local pixel px of row `row` of the wave's raw ring: max over the 16 nine-arcs of
// min(v - ring) and of min(ring - v), minus 1; both polarities ride in one register as a packed pair. The pixel is a
// FAST-9 corner for threshold t iff the result is >= t. (k_fast_blur_band's fast_score_pk on ring addresses.)
//
// Round 4: the pair is packed HALF floats and the trees use three-input minimum / maximum: fast9_score_f16 (orb_device.h).
__device__ __forceinline__ int fast_score_ring(const uint8_t* ring, int px, int row) {
    const uint8_t* rm3 = ring + ring_off(row - 3) + px; const uint8_t* rm2 = ring + ring_off(row - 2) + px;
    const uint8_t* rm1 = ring + ring_off(row - 1) + px; const uint8_t* c = ring + ring_off(row) + px;
    const uint8_t* rp1 = ring + ring_off(row + 1) + px; const uint8_t* rp2 = ring + ring_off(row + 2) + px;
    const uint8_t* rp3 = ring + ring_off(row + 3) + px;
    const uint32_t v = c[0];
    uint32_t rg[16] = {rp3[0], rp3[1], rp2[2], rp1[3], c[3], rm1[3], rm2[2], rm3[1],
                       rm3[0], rm3[-1], rm2[-2], rm1[-3], c[-3], rp1[-3], rp2[-2], rp3[-1]};
#if ARIA_SCORE_F16
    return fast9_score_f16(v, rg);
#else
    uint32_t Pk[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        // (v - ring, ring - v) from ONE packed register (v | ring << 16) and itself with the halves swapped (op_sel of the
        // packed subtract): one operand to build instead of two
        // (the compiler does not fold the swizzle into op_sel: it builds the swapped operand with a v_perm)
        const uint32_t x = v | (rg[k] << 16);
        asm("v_pk_sub_i16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(Pk[k]) : "v"(x));
    }
    uint32_t M2[16], M4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) M2[k] = pk_min_i16(Pk[k], Pk[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) M4[k] = pk_min_i16(M2[k], M2[(k + 2) & 15]);
    uint32_t Q = 0x80008000u;   // (-32768, -32768)
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t m9 = pk_min_i16(pk_min_i16(M4[k], M4[(k + 4) & 15]), Pk[(k + 8) & 15]);
        Q = pk_max_i16(Q, m9);
    }
    const int q0 = (int)(short)(Q & 0xFFFFu), q1 = (int)(short)(Q >> 16);
    return max(q0, q1) - 1;
#endif
}

// What a lane is, fixed for the life of the wave.
struct LaneRole {
    int frame;            // frame of the launch this lane's dword belongs to
    int gdw;              // dword index inside the level row (-1: left halo, D: right halo)
    uint32_t sel;         // v_perm selector that turns the loaded dword into the lane's pixels (identity / mirrored forms)
    uint32_t xm0, xm1;    // FAST x-range masks of px 0,1 / 2,3 (bit 15, bit 31), zero where the lane must not report
    uint32_t tailbits;    // bit i: pixel i rounds ties UP (scalar tail of OpenCV's column filter)
    bool owner;           // stores blurred pixels and reports candidates
};

// Everything the kernel needs about ONE level launch, as plain scalars (a Plan indexed by a run-time level lives in kernarg
// memory and every use of a field in the walk became a scalar load + wait: the kernel is short of SGPRs).
struct StreamArgs {
    const uint8_t* src;       // level l, frame 0, row 0
    uint8_t* blur;            // blurred level l of frame 0
    uint8_t* next;            // raw level l+1 of frame 0, or nullptr
    uint32_t* cand;           // candidate list of (frame 0, level l)
    int* cand_cnt;            // candidate counter of (frame 0, level l); frame stride kLevels
    int* err;
    const uint32_t* xt;       // tables of level l+1
    const uint32_t* yt;
    const uint32_t* xinv;
    const uint32_t* ytr;      // pyramid step by source row of level l (always a readable table, also without a next level)
    int64_t src_fstride, blur_fstride, next_fstride;     // bytes between frames
    int cand_fstride;         // entries between frames
    int cand_cap;
    int src_pitch, blur_pitch, next_pitch;
    int w, h, next_w, next_h;
    int n_frames, thr, tail_start;
    // Rows are cut into segments of seg_rows (the last one shorter): a wave walks ONE segment of its panel, with the rows
    // around it that its first / last output rows need (4 above, 3 below). One segment per level would give the fewest
    // recomputed rows, but few, long-lived waves: a 1408-row level of 1024 frames is less than one round of waves on 256 CUs
    // and the last ones run alone. The host picks the segment so that a launch has several rounds (launch_fast_blur_stream).
    int seg_rows, n_seg, panels, panels8;
};

}  // namespace

// FB: the wave does FAST + NMS + blur; PYR: it does the pyramid step. Both = the fused wave (round 3). ARIA_STREAM_SPLIT (round 4)
// gives the two jobs to two waves of the same launch: the FAST/blur wave then fits 128 VGPRs and, with the compact lists,
// 10 KB of LDS -- four waves per SIMD instead of three --, and the pyramid wave (a few dozen registers, memory-bound) runs
// beside it on the same CUs.
template <int WPB, bool TAIL, bool FB, bool PYR>
__device__ __forceinline__ void stream_wave(const StreamArgs& A, uint8_t* __restrict__ wl, const int lane, const LaneRole R,
                                            const int frame0, const int seg, unsigned long long* __restrict__ stamps) {
    // diagnostic builds only (-DARIA_DIAG, ARIA_STREAM_STAMPS=1): s_memtime ticks per phase, summed over the waves of the
    // launch. Not in the product build: the accumulators alone are 16 SGPRs of a kernel that is short of them.
#ifdef ARIA_DIAG
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
#define PHASE(k) do { if (stamps) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[k] += tn - tlast; tlast = tn; } } while (0)
#else
    (void)stamps;
#define PHASE(k) do { } while (0)
#endif
    const int w = A.w, h = A.h;
    // this wave's rows: blurred rows, candidates and pyramid source rows [r0, r1); ingest indices t_first .. t_last
    const int r0 = seg * A.seg_rows, r1 = min(h, r0 + A.seg_rows);
    // (kG == 8: the first group's first OUTPUT row t_first - 3 is a multiple of 4 -- r0 is -- so that every group is two whole
    // row quads; the one or two extra rows above the segment cost a step each and feed no output)
    // (a pyramid-only wave needs the rows its output rows blend: upper source rows r0 .. r1 - 1 and the row below the last one)
    const int t_first = !FB ? r0 : (kG == 8 ? r0 - 5 : (r0 == 0 ? -3 : r0 - 4));
    const int t_last = !FB ? min(r1, h - 1) : (r1 == h ? h + 2 : r1 + 3);
    uint32_t* s_raw = reinterpret_cast<uint32_t*>(wl + kHdrBytes);                 // [kRing][64] dwords
    uint8_t* s_rawb = wl + kHdrBytes;
    uint8_t* s_map = wl + kHdrBytes + kRawBytes;                                   // [kRing][256] scores
    uint32_t* s_mapw = reinterpret_cast<uint32_t*>(s_map);
    qent_t* s_q1 = reinterpret_cast<qent_t*>(wl + kHdrBytes + kRawBytes + kMapBytes + kMapPad);
    qent_t* s_q2 = s_q1 + kQ1;
    uint32_t* s_out = reinterpret_cast<uint32_t*>(s_q2 + kQ2);      // [kOut] candidate records (+ [kOut] their frames in the wide form)
#if !ARIA_STREAM_COMPACT_LDS
    uint32_t* s_outf = s_out + kOut;
    static_assert(!ARIA_STREAM_PEND_LDS, "ARIA_STREAM_PEND_LDS goes with the compact lists");
#endif

    // ---- addresses: wave-uniform 64-bit bases + 32-bit lane offsets ----
    const int pitch_in = A.src_pitch;
    const uint8_t* src0 = A.src + (int64_t)frame0 * A.src_fstride;                 // level l of frame0
    const int64_t fstride_in = A.src_fstride;
    const int r4 = w & 3, D = (w + 3) >> 2;
    int xload;
    if (R.gdw < 0) xload = 1;
    else if (R.gdw >= D) xload = r4 ? w - 9 + r4 : w - 5;
    else if (r4 && R.gdw == D - 1) xload = w - 4;
    else xload = 4 * R.gdw;
    const uint32_t in_off = (uint32_t)((int64_t)(R.frame - frame0) * fstride_in + xload);
    uint8_t* blur0 = A.blur + (int64_t)frame0 * A.blur_fstride;
    // the blurred level is stored in Q4 order (orb_device.h): dword column gdw of row o at (o >> 2) * 4 * pitch + 16 gdw + 4 (o & 3)
    const uint32_t out_off = (uint32_t)((int64_t)(R.frame - frame0) * A.blur_fstride + 16 * max(R.gdw, 0));

    // keypoints.cpp runByImageBorder keeps x in [31, w-31), y in [31, h-31); FAST is needed on that region + 1 ring
    // ... restricted to the segment + the row above and below it (their scores are the NMS neighbours of its edge rows)
    const int fy0 = max(kEdgeThreshold - 1, r0 - 1), fy1 = min(h - kEdgeThreshold, r1);
    const int thr = A.thr;
    const uint32_t T2 = (uint32_t)thr * 0x00010001u;
    const uint32_t KLO = 18u | (34u << 8) | (49u << 16) | (55u << 24);   // taps x-3..x
    const uint32_t KHI = 49u | (34u << 8) | (18u << 16);                 // taps x+1..x+3 (x+4 weight 0)

    // per-pixel rounding flags of the column pass (TAIL variant): bias 0.5 and "take the floor" for tail pixels
    float tb[4] = {0.f, 0.f, 0.f, 0.f}, tm[4] = {0.f, 0.f, 0.f, 0.f};
    if (TAIL) {
#pragma unroll
        for (int j = 0; j < 4; j++) { tb[j] = (R.tailbits >> j) & 1u ? 0.5f : 0.f; tm[j] = (R.tailbits >> j) & 1u ? 1.f : 0.f; }
    }

    // ---- the pyramid step: the lane hosts the output dword of level l+1 whose anchor source column lies in its 4 px ----
#if defined(ARIA_STREAM_NOPYR)
    const bool pyr = false;
#else
    const bool pyr = PYR && A.next != nullptr;
#endif
    const int gn_w = A.next_w, gn_h = A.next_h;
    int host_gx = -1;
    uint32_t xw[4] = {0, 0, 0, 0}, xo[4] = {0, 0, 0, 0}, xs[4] = {0, 0, 0, 0};
    uint32_t nout_off = 0;
    uint8_t* next0 = nullptr;
    const uint32_t* yt = nullptr;
    if (pyr) {
        yt = A.yt;
        next0 = A.next + (int64_t)frame0 * A.next_fstride;
        if (R.owner) {
            const uint32_t hv = A.xinv[R.gdw];
            if (hv != 0xFFFFFFFFu) host_gx = (int)hv;
        }
        if (host_gx >= 0) {
            const uint32_t* xt = A.xt;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t t = xt[min(4 * host_gx + i, gn_w - 1)];
                const int ox = (int)(t & 0xFFFFu);
                const uint32_t cx1 = t >> 16;
                const int lpx = 4 * lane + (ox - 4 * R.gdw);            // local pixel of source column ox in the wave's ring row
                xw[i] = (256u - cx1) | (cx1 << 16);
#if ARIA_STREAM_PYR_INWALK
                // byte index of source column ox in the register pair the pixel blends from: pixel 0 in (left neighbour, own
                // dword), pixels 1..3 in (own dword, right neighbour) -- build_plan checks that every pair lies inside
                (void)lpx;
                xs[i] = 0x0C010C00u + (uint32_t)(ox - 4 * R.gdw + (i == 0 ? 4 : 0)) * 0x00010001u;
                if (4 * host_gx + i >= gn_w) xs[i] = 0x0C0C0C0Cu;        // padding pixel of a partial last dword: zeros
#else
                xo[i] = (uint32_t)(lpx & ~3);
                xs[i] = 0x0C010C00u + (uint32_t)(lpx & 3) * 0x00010001u;
#endif
            }
            nout_off = (uint32_t)((int64_t)(R.frame - frame0) * A.next_fstride + 4 * host_gx);
        }
    }
#if ARIA_STREAM_PYR_INWALK
    static_assert(!ARIA_STREAM_SPLIT && ARIA_STREAM_FLAT && kG == 7, "the in-walk pyramid step belongs to the fused, flat walk");
    // the table by source row goes through the scalar cache (wave-uniform index): seven words per group, loaded at its top
    typedef const uint32_t __attribute__((address_space(4))) * const_u32p;
    const const_u32p ytrs = reinterpret_cast<const_u32p>(reinterpret_cast<uint64_t>(A.ytr));
    uint32_t hprev[4] = {0u, 0u, 0u, 0u};          // horizontal blend of the previous source row (the upper row of the next output row)
    (void)yt; (void)xo; (void)gn_h;
#else
    // next output row of level l+1 (wave-uniform): the first one whose upper source row lies in this segment
    int dy_next = 0;
    if (pyr && r0 > 0) {
        dy_next = min(max((int)(((int64_t)(2 * r0 + 1) * gn_h) / (2 * h)) - 1, 0), gn_h - 1);      // ~ (r0 + 0.5) gn_h / h - 0.5, then exact:
        while (dy_next > 0 && (int)(yt[dy_next - 1] & 0xFFFFu) >= r0) dy_next--;
        while (dy_next < gn_h && (int)(yt[dy_next] & 0xFFFFu) < r0) dy_next++;
        dy_next = __builtin_amdgcn_readfirstlane(dy_next);
    }
    // the y-table words of 64 output rows at a time, one per lane (a load per row would put an L2 round trip in front of
    // every output row)
    int yt_base = dy_next;
    uint32_t ytv = pyr ? gload_sv<uint32_t>(reinterpret_cast<const uint8_t*>(A.yt), 4u * (uint32_t)min(yt_base + lane, gn_h - 1)) : 0u;
    uint32_t hprev[4] = {0u, 0u, 0u, 0u};
    int hprev_row = -1;

#endif
    // ---- the 7-row register window in kG slots; slot = position of the row in its group (static after unrolling by kG) ----
    uint32_t RC2[kG][2], RE[kG], RW[kG];
    float RF[kG][4];
#pragma unroll
    for (int u = 0; u < kG; u++) {
        RC2[u][0] = RC2[u][1] = RE[u] = RW[u] = 0;
        RF[u][0] = RF[u][1] = RF[u][2] = RF[u][3] = 0.f;
    }

    auto row_ptr = [&](int t) -> const uint8_t* {                  // level row of ingest index t (REFLECT_101 above / below)
        const int ry = reflect101(min(max(t, -3), t_last), h);      // (the loads issued past the segment's end repeat its last row: cache hits)
        return src0 + (int64_t)ry * pitch_in;
    };
    // Row loads run TWO groups ahead of the walk and are taken out of their registers in the middle of a group (after the
    // scoring phase): vmcnt counts loads and stores together, in order, and the compiler cannot know how many stores follow
    // a load, so taking a row out of its register waits for EVERY vector-memory operation issued so far. At the top of a
    // group that meant waiting for the stores the pyramid step had issued a moment earlier (a full write round trip per
    // group); behind the scoring phase the youngest stores are the blurred rows of the walk, thousands of cycles old.
    uint32_t pre[kG], cur[kG];
#pragma unroll
    for (int u = 0; u < kG; u++) pre[u] = gload_sv<u32_unaligned>(row_ptr(t_first + u), in_off);
#pragma unroll
    for (int u = 0; u < kG; u++) cur[u] = __builtin_amdgcn_perm(0u, pre[u], R.sel);

    // per-lane info for whoever processes a queue entry of this lane's pixels: x of px 0 | frame - frame0 | owner
    const uint32_t linfo = (uint32_t)(4 * max(R.gdw, 0)) | ((uint32_t)(R.frame - frame0) << 11) | (R.owner ? 1u << 31 : 0u);

#if ARIA_STREAM_COMPACT_LDS
    const uint32_t lane_hi = (uint32_t)lane << 5;      // queue entry of a survivor: lane << 5 | bit of accw (11 bits)
#else
    const uint32_t lane_hi = (uint32_t)lane << 8;      // queue entry of a survivor: lane << 8 | bit of accw  (>> 6 = 4 * lane)
#endif
    int q2n = 0;                 // corners waiting for their lower neighbours' scores (wave-uniform)
    bool dense = false;          // the corner list overflowed once: NMS scans the score ring from here on
    int nms_done = fy0 - 1;      // rows <= this have been through NMS (diagnostic-free: the dense path starts after it)

    // ---- candidates leave through an out-list in LDS, 64 at a time, and the global append is split in two: the slice of
    //      the frame's list is reserved in the NMS phase (one returning atomic per frame in the batch, NOT waited for), the
    //      records are stored after the next rows have been taken out of their registers, a phase later. A wave never sits
    //      out an L2 atomic round trip (1-2 us each, once per group of rows, was a sixth of this kernel's time). ----
    int outn = 0;                        // entries in the out-list (wave-uniform)
    // the batch whose slice has been requested, one entry per lane: the record, and in one register the frame (relative to
    // frame0, 20 bits), the leader lane of the frame's slice (6 bits) and the rank inside it (6 bits)
    uint32_t p_rec = 0, p_meta = 0;
    int p_base = 0;
    bool p_valid = false;
    bool pend_any = false;               // a batch is waiting for its slice (wave-uniform)
    // the first min(outn, 64) entries of the out-list, one per lane: record, frame, and per frame of the batch the leader
    // lane, the rank inside the frame's slice and (in the leader) the size of the slice
#if ARIA_STREAM_COMPACT_LDS
    // entry `lane` of the out-list as the final record: compact record = local px | row << 8 | score << 19; x and the frame come
    // from the lane that holds the pixel
    auto front_record = [&](bool valid, uint32_t& rec, int& frame) {
        const uint32_t rc = valid ? s_out[lane] : 0u;
        const uint32_t lpx = rc & 0xFFu;
        const uint32_t li = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lpx >> 2) << 2), (int)linfo);
        rec = ((li & 0x7FFu) + (lpx & 3u)) | (((rc >> 8) & 0x7FFu) << 11) | ((rc >> 19) << 22);
        frame = frame0 + (int)(valid ? ((li >> 11) & 0xFFFFFu) : 0u);
    };
#endif
    auto take_batch = [&](uint32_t& rec, int& frame, bool& valid, uint32_t& leader, uint32_t& rank, int& my_count) {
        const int nb = min(outn, 64);
        valid = lane < nb;
#if ARIA_STREAM_COMPACT_LDS
        front_record(valid, rec, frame);
#else
        rec = valid ? s_out[lane] : 0u;
        frame = frame0 + (int)(valid ? s_outf[lane] : 0u);
#endif
        unsigned long long pend = __builtin_amdgcn_ballot_w64(valid);
        my_count = 0; leader = 0; rank = 0;
        while (pend) {
            const int ld = __ffsll((long long)pend) - 1;
            const int f = __builtin_amdgcn_readlane(frame, ld);
            const bool mine = valid && frame == f;
            const unsigned long long mm = __builtin_amdgcn_ballot_w64(mine);
            if (mine) {
                leader = (uint32_t)ld;
                rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
            }
            if (lane == ld) my_count = __popcll(mm);
            pend &= ~mm;
        }
    };
    auto drop_batch = [&]() {            // entries 64.. move to the front
        const int nb = min(outn, 64);
#if ARIA_STREAM_COMPACT_LDS
        const uint32_t mr = lane + 64 < outn ? s_out[lane + 64] : 0u;
        wave_sync();
        if (lane + 64 < outn) s_out[lane] = mr;
#else
        const uint32_t mr = lane + 64 < outn ? s_out[lane + 64] : 0u, mf = lane + 64 < outn ? s_outf[lane + 64] : 0u;
        wave_sync();
        if (lane + 64 < outn) { s_out[lane] = mr; s_outf[lane] = mf; }
#endif
        outn -= nb;
        wave_sync();
    };
    auto store_batch = [&](uint32_t rec, int frame, bool valid, uint32_t leader, uint32_t rank, int base_of_leader) {
        const int base = __builtin_amdgcn_ds_bpermute((int)(leader << 2), base_of_leader);
        if (valid) {
            const int at = base + (int)rank;
            if (at < A.cand_cap) A.cand[(int64_t)frame * A.cand_fstride + at] = rec;
            else atomicOr(A.err, ERRBIT_CAND_OVERFLOW);
        }
    };
    // The ONE place a slice is reserved without waiting (end of the NMS phase) and the ONE place its result is read (after
    // the next rows have left their registers): inlined more than once, the compiler merges the copies' result registers
    // with moves, and a move of a pending atomic result is an s_waitcnt vmcnt(0).
    auto flush_async = [&]() {
        wave_sync();
        int my_count;
        uint32_t leader, rank;
        int frame;
        take_batch(p_rec, frame, p_valid, leader, rank, my_count);
        if (my_count > 0) p_base = atomicAdd(A.cand_cnt + frame * kLevels, my_count);
        p_meta = (uint32_t)(frame - frame0) | (leader << 20) | (rank << 26);
        pend_any = true;
#if !ARIA_STREAM_PEND_LDS
        drop_batch();
#endif
    };
    auto complete_pending = [&]() {
#if ARIA_STREAM_PEND_LDS
        {   // the batch is still the first 64 entries of the out-list (nothing has been appended since): read it again, drop it now
            int frame_again;
            front_record(p_valid, p_rec, frame_again);
            (void)frame_again;
        }
#endif
        store_batch(p_rec, frame0 + (int)(p_meta & 0xFFFFFu), p_valid, (p_meta >> 20) & 63u, p_meta >> 26, p_base);
#if ARIA_STREAM_PEND_LDS
        drop_batch();
#endif
        p_valid = false;
        pend_any = false;
    };
    // the out-list is full in the middle of an NMS phase (more than ~64 candidates from one group of rows: corner-dense
    // images), or the wave is done: reserve, wait, store
    auto flush_sync = [&](bool all) {
        wave_sync();
        while (outn >= (all ? 1 : 64)) {
            uint32_t rec, leader, rank; int frame, my_count; bool valid;
            take_batch(rec, frame, valid, leader, rank, my_count);
            int base = 0;
            if (my_count > 0) base = atomicAdd(A.cand_cnt + frame * kLevels, my_count);
            store_batch(rec, frame, valid, leader, rank, base);
            drop_batch();
        }
    };

    const int G = (t_last - t_first + 1 + kG - 1) / kG;
    // (loads past the last group are issued all the same, clamped to the last row: a condition around them would make the
    // compiler load into temporaries and copy -- i.e. wait -- at once)
#pragma unroll
    for (int u = 0; u < kG; u++) pre[u] = gload_sv<u32_unaligned>(row_ptr(t_first + kG + u), in_off);
    for (int gi = 0; gi < G; gi++) {
        const int t0 = t_first + kG * gi;
        // ---- rows of this group into the ring; clear the score rows this group will fill ----
#pragma unroll
        for (int u = 0; u < kG; u++) {
            s_raw[(((t0 + u + 3) & (kRing - 1)) << 6) + lane] = cur[u];
            if constexpr (FB) s_mapw[(((t0 + u) & (kRing - 1)) << 6) + lane] = 0u;             // score row o = t - 3 (slot (o + 3) & 15)
        }
        wave_sync();
        // y-table window of the pyramid step: refilled here, a whole walk ahead of its first use
#if !ARIA_STREAM_PYR_INWALK
        if (pyr && dy_next + kG + 2 > yt_base + 64) { yt_base = dy_next; ytv = gload_sv<uint32_t>(reinterpret_cast<const uint8_t*>(A.yt), 4u * (uint32_t)min(yt_base + lane, gn_h - 1)); }
#else
        uint32_t yr[kG];
#pragma unroll
        for (int u = 0; u < kG; u++) yr[u] = ytrs[min(max(t0 + u, 0), h - 1)];
#endif
        PHASE(0);

        if constexpr (FB) {
        // ---- the walk: row pass, column pass + store, compass reject ----
        uint32_t accw = 0;     // survivors of this group: step u, px j -> bit (j&1 ? 31 : 15) - (j>>1) - 2u
        u32x4 qv = {0u, 0u, 0u, 0u};      // kG == 8: the four blurred dwords of the lane's row quad
#pragma unroll
        for (int u = 0; u < kG; u++) {
            const int t = t0 + u;
            // (flat walk: rows past t_last -- only in a segment's last group -- are copies of the last row, walked like any other;
            // their outputs fail the two range tests below)
            if (ARIA_STREAM_FLAT || t <= t_last) {
                // neighbours' dwords straight from their registers (DPP wave shifts): no LDS round trip in the walk. Lane 0's
                // left and lane 63's right neighbour do not exist (they read 0): those two lanes only provide pixels.
                const uint32_t w1 = cur[u];
                const uint32_t w0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w1, 0x138, 0xF, 0xF, true);    // wave_shr:1
                const uint32_t w2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w1, 0x130, 0xF, 0xF, true);    // wave_shl:1
                RC2[u][0] = __builtin_amdgcn_perm(0u, w1, 0x0c010c00u);   // px 0, 1 as int16 pair
                RC2[u][1] = __builtin_amdgcn_perm(0u, w1, 0x0c030c02u);   // px 2, 3
                RW[u] = __builtin_amdgcn_alignbyte(w1, w0, 1);     // x-3 .. x
                RE[u] = __builtin_amdgcn_alignbyte(w2, w1, 3);     // x+3 .. x+6
                const uint32_t rs0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), KHI,
                                                            __builtin_amdgcn_udot4(RW[u], KLO, 0u, false), false);
                const uint32_t rs1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), KHI,
                                                            __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), KLO, 0u, false), false);
                const uint32_t rs2 = __builtin_amdgcn_udot4(RE[u], KHI,
                                                            __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), KLO, 0u, false), false);
                const uint32_t rs3 = __builtin_amdgcn_udot4(w2, KHI, __builtin_amdgcn_udot4(w1, KLO, 0u, false), false);
                RF[u][0] = (float)rs0; RF[u][1] = (float)rs1; RF[u][2] = (float)rs2; RF[u][3] = (float)rs3;   // exact: < 2^16
#if ARIA_STREAM_PYR_INWALK
                if constexpr (PYR) {
                    // ---- a6.1 in the walk: this row's horizontal blend for the hosted output dword (v_perm picks the two source
                    //      bytes of a pixel out of the registers, v_dot2 weighs them: the integers of k_resize_lds), blended with
                    //      the previous row's into the output row whose LOWER source row this is. Rows that are nobody's lower
                    //      row (one in six), rows of another segment's output rows and lanes that host nothing store nothing. ----
                    constexpr uint32_t put[4] = {0x03020106u, 0x03020600u, 0x03060100u, 0x06020100u};   // byte 2 of v -> byte i
                    const uint32_t ye = yr[u];
                    const uint32_t cy1 = (ye >> 16) & 0x1FFu, cyp = (256u - cy1) | (cy1 << 16);
                    uint32_t poutw = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t pair = i == 0 ? __builtin_amdgcn_perm(w1, w0, xs[0]) : __builtin_amdgcn_perm(w2, w1, xs[i]);
                        const uint32_t hc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, pair), __builtin_bit_cast(us2v, xw[i]), 0u, false);
                        const uint32_t v = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, hprev[i] | (hc << 16)),
                                                                  __builtin_bit_cast(us2v, cyp), 32768u, false);   // < 2^24
                        poutw = __builtin_amdgcn_perm(v, poutw, put[i]);
                        hprev[i] = hc;
                    }
                    // (output rows belong to the segment that holds their UPPER source row t - 1: r0 <= t - 1 < r1)
                    const bool out_row = (ye >> 31) != 0u && t > r0 && t <= min(r1, h - 1);
                    if (out_row && host_gx >= 0) gstore_sv<uint32_t>(next0 + (int64_t)(ye & 0xFFFFu) * A.next_pitch, nout_off, poutw);
                }
#endif

                const int o = t - 3;              // level row whose window [o-3, o+3] is now complete
                // row o+d of the window lives in slot (u - 3 + d) mod kG
                const int sC = (u + kG - 3) % kG, sM1 = (u + kG - 4) % kG, sP1 = (u + kG - 2) % kG, sM2 = (u + kG - 5) % kG,
                          sP2 = (u + kG - 1) % kG, sM3 = (u + kG - 6) % kG, sP3 = u;
                const bool blur_row = o >= r0 && o < r1;
                if (ARIA_STREAM_FLAT || blur_row) {
                    // vertical pass + rounding by 2^16 (filter.simd.hpp SymmColumnFilter / SymmColumnVec_32s8u). All values are
                    // integers * 2^-16 below 2^9, so the fp32 column pass is exact (a partial sum can only be inexact above
                    // 256.0, which saturates either way); v_cvt_pk_u8_f32 rounds to nearest EVEN (the SIMD path), saturates
                    // and inserts the byte. Tail pixels (ties UP): x = acc + 0.5 (the bias rides in the first multiply-add),
                    // floor(x) = x - fract(x), an integer, which the same conversion leaves alone.
                    constexpr float k0 = 55.f / 65536.f, k1 = 49.f / 65536.f, k2 = 34.f / 65536.f, k3 = 18.f / 65536.f;
                    uint32_t outw = 0;
                    // (Written pixel by pixel on purpose. The same arithmetic on float2 values -- which the compiler turns into
                    // v_pk_add_f32 with op_sel-swizzled register pairs -- gave a few wrong blurred pixels per ~1000 frames
                    // whenever the matcher's MFMA waves shared the SIMDs, and never alone: tools/race_probe.py,
                    // tests/test_gpu_stream.py::test_extraction_beside_the_matcher_equals_extraction_alone.)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float acc = TAIL ? __builtin_fmaf(k0, RF[sC][j], tb[j]) : k0 * RF[sC][j];
                        acc = __builtin_fmaf(k1, RF[sM1][j] + RF[sP1][j], acc);
                        acc = __builtin_fmaf(k2, RF[sM2][j] + RF[sP2][j], acc);
                        acc = __builtin_fmaf(k3, RF[sM3][j] + RF[sP3][j], acc);
                        if (TAIL) acc = __builtin_fmaf(-__builtin_amdgcn_fractf(acc), tm[j], acc);
                        outw = __builtin_amdgcn_cvt_pk_u8_f32(acc, j, outw);
                    }
                    // Q4 order (orb_device.h): dword column gdw of row o at (o >> 2) * 4 * pitch + 16 gdw + 4 (o & 3). What the
                    // layout costs HERE (640x480, 8192 frames, rocprofv3, one box): row-major 1.626 us per frame, Q4 1.703
                    // (levels 0-2 +9 / +6 / +6 %, levels 5-7 < 2 %; HBM bytes written +4.7 %, fetched the same) -- against
                    // 0.127 us less in k_describe. Collecting the four rows of a quad in registers for one 16-byte store
                    // needs a wave-uniform but run-time register index (groups are 7 rows, quads 4), which the compiler
                    // turns into four v_cndmask per row: 1.777 us; parking them in LDS under that index and storing 16 bytes
                    // per quad: +8 %; non-temporal stores: +35 % (the L2 no longer merges the rows of a line).
                    if constexpr (kG == 8) qv[u & 3] = outw;
                    else if (R.owner && blur_row) gstore_sv<uint32_t>(blur0 + (int64_t)(o >> 2) * (4 * A.blur_pitch) + ((o & 3) << 2), out_off, outw);
                } else if constexpr (kG == 8) {
                    qv[u & 3] = 0u;           // rows outside the segment (or below the image, in the level's padded last quad)
                }
                const bool fast_row = o >= fy0 && o <= fy1;
                // (flat walk: a row outside the FAST range gets a threshold no difference reaches -- a scalar select -- instead of
                // a branch around the test)
                const uint32_t T2r = ARIA_STREAM_FLAT ? (fast_row ? T2 : 0x40004000u) : T2;
                if (ARIA_STREAM_FLAT || fast_row) {
                    // compass reject, two pixels per packed-int16 op (see k_fast_blur_band): survive iff one of N, S AND one
                    // of E, W are darker than c - t, or the same with brighter than c + t
                    uint32_t pass[2];
#pragma unroll
                    for (int pr = 0; pr < 2; pr++) {
                        const uint32_t sl = pr ? 0x0c030c02u : 0x0c010c00u;
                        const uint32_t c2 = RC2[sC][pr], n2 = RC2[sM3][pr], s2 = RC2[sP3][pr];
                        const uint32_t e2 = __builtin_amdgcn_perm(0u, RE[sC], sl);
                        const uint32_t w2p = __builtin_amdgcn_perm(0u, RW[sC], sl);
#if ARIA_COMPASS2
                        // darker: X = max(min(n,s), min(e,w)) < c - t; brighter: Y = min(max(n,s), max(e,w)) > c + t; both in
                        // one comparison: t - max(c - X, Y - c) < 0 (sign bits 15 / 31)
                        const uint32_t X = pk_max_i16(pk_min_i16(n2, s2), pk_min_i16(e2, w2p));
                        const uint32_t Y = pk_min_i16(pk_max_i16(n2, s2), pk_max_i16(e2, w2p));
                        const uint32_t m = pk_max_i16(pk_sub_i16(c2, X), pk_sub_i16(Y, c2));
                        pass[pr] = pk_sub_i16(T2r, m) & (pr ? R.xm1 : R.xm0);
#else
                        const uint32_t lo = pk_sub_i16(c2, T2), hi = pk_add_i16(c2, T2);
                        const uint32_t dk = pk_sub_i16(pk_max_i16(pk_min_i16(n2, s2), pk_min_i16(e2, w2p)), lo);
                        const uint32_t br = pk_sub_i16(hi, pk_min_i16(pk_max_i16(n2, s2), pk_max_i16(e2, w2p)));
                        pass[pr] = (dk | br) & (pr ? R.xm1 : R.xm0);
#endif
                    }
                    accw |= (pass[0] | (pass[1] >> 1)) >> (2 * u);
                }
            } else if constexpr (kG == 8) {
                qv[u & 3] = 0u;
            }
            if constexpr (kG == 8) {
                // the quad's four rows are in: one 16-byte store per lane (Q4 order: 16 gdw inside row quad qo >> 2), as long as
                // the quad belongs to this segment (segments are whole quads; the last quad of a level may reach below it)
                if ((u & 3) == 3) {
                    const int qo = t0 + u - 6;                       // first row of the quad: o - 3 of this step
                    if (qo >= r0 && qo < r1 && R.owner) gstore_sv<u32x4>(blur0 + (int64_t)(qo >> 2) * (4 * A.blur_pitch), out_off, qv);
                }
            }
        }

        PHASE(1);
        // ---- survivors -> queue -> dense scoring. The queue holds kQ1 entries; the wave's survivors are numbered by a
        //      prefix sum and go through it in rounds (one round unless the panel is very corner-dense). Corners: score
        //      into the score ring, (px, row, score) onto the corner list. ----
        const int o_lo = t0 - 3;                                       // survivor rows of this group: o_lo .. o_lo + 6
        if (__builtin_amdgcn_ballot_w64(accw != 0)) {
            const int cnt = __popc(accw);
            const int incl = wave_scan_incl(cnt);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            int my = incl - cnt;                                       // global number of this lane's next survivor
            for (int r0 = 0; r0 < total; r0 += kQ1) {
                // append the survivors numbered r0 .. r0 + kQ1 - 1
                // (this loop runs as many trips as the busiest lane has survivors, with few lanes active: it only records
                // (lane, bit); the dense side below turns that into pixel and row)
                while (accw && my < r0 + kQ1) {
                    const int b = 31 - __clz(accw);
                    accw &= ~(1u << b);
                    s_q1[my - r0] = (qent_t)(lane_hi | (uint32_t)b);
                    my++;
                }
                wave_sync();
                const int qn = min(kQ1, total - r0);
                for (int i0 = 0; i0 < qn; i0 += 64) {
                    const int i = i0 + lane;
                    const uint32_t eq = i < qn ? (uint32_t)s_q1[i] : 0u;
                    // bit b of accw -> step u, pixel px (see accw above)
                    const int b = (int)(eq & 31u), hi16 = b >> 4, bb = 15 - (b & 15);
#if ARIA_STREAM_COMPACT_LDS
                    const uint32_t lane4 = (eq >> 3) & 0xFCu;          // 4 * lane
#else
                    const uint32_t lane4 = eq >> 6;
#endif
                    const uint32_t e = (lane4 + (uint32_t)(((bb & 1) << 1) | hi16)) | ((uint32_t)(o_lo + (bb >> 1)) << 8);    // local px | row << 8
                    int sc = 0;
                    if (i < qn) sc = fast_score_ring(s_rawb, (int)(e & 0xFFu), (int)(e >> 8));
                    const bool corner = i < qn && sc >= thr;
                    if (corner) s_map[ring_off((int)(e >> 8)) + (e & 0xFFu)] = (uint8_t)sc;
                    const unsigned long long cm = __builtin_amdgcn_ballot_w64(corner);
                    if (cm) {
                        const int at = q2n + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
#if ARIA_STREAM_COMPACT_LDS
                        // corner entry: local px | ring slot << 8 (the score is in the score ring, the row follows from the slot)
                        if (corner && at < kQ2) s_q2[at] = (qent_t)((e & 0xFFu) | ((((e >> 8) + 3u) & (kRing - 1)) << 8));
#else
                        if (corner && at < kQ2) s_q2[at] = e | ((uint32_t)sc << 24);
#endif
                        q2n += __popcll(cm);
                    }
                }
                wave_sync();
            }
        }
        wave_sync();
        PHASE(2);
        if (q2n > kQ2) { dense = true; }
        if (dense) q2n = 0;

        // ---- 3x3 strict-max NMS + border filter for the rows whose lower neighbours are scored: rows <= o_lo + 5 (all of
        //      them in the last group). Candidates go to the frame's list through a wave-aggregated append. ----
        const int nms_hi = gi + 1 < G ? min(o_lo + kG - 2, fy1) : fy1;
        auto emit = [&](bool keep, int px, int row, int sc) {
            // (x of the lane's px 0, frame, owner) of the lane that holds the pixel
            const uint32_t li = __builtin_amdgcn_ds_bpermute((px >> 2) << 2, (int)linfo);
            const int X = (int)(li & 0x7FFu) + (px & 3);
            keep = keep && (li >> 31) && X >= kEdgeThreshold && X < w - kEdgeThreshold && row >= max(kEdgeThreshold, r0) && row < min(h - kEdgeThreshold, r1);
            const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
            if (km) {
                if (outn > kOut - 64) flush_sync(false);
                const int at = outn + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                if (keep) {
#if ARIA_STREAM_COMPACT_LDS
                    s_out[at] = (uint32_t)px | ((uint32_t)row << 8) | ((uint32_t)sc << 19);
#else
                    s_out[at] = (uint32_t)X | ((uint32_t)row << 11) | ((uint32_t)sc << 22);
                    s_outf[at] = (li >> 11) & 0xFFFFFu;
#endif
                }
                outn += __popcll(km);
            }
        };
        if (!dense) {
            int kept = 0;
            for (int i0 = 0; i0 < q2n; i0 += 64) {
                const int i = i0 + lane;
#if ARIA_STREAM_COMPACT_LDS
                // live corner rows lie in [o_lo - 1, o_lo + 6]: the row follows from the ring slot relative to ref = o_lo - 8
                const qent_t e = i < q2n ? s_q2[i] : (qent_t)0;
                const int px = (int)(e & 0xFFu);
                const int ref = o_lo - 8;
                const int row = ref + ((((int)(e >> 8)) - (ref + 3)) & (kRing - 1));
                const int sc = i < q2n ? (int)s_map[(((int)(e >> 8)) << 8) + px] : 0;
#else
                const uint32_t e = i < q2n ? s_q2[i] : 0u;
                const int px = (int)(e & 0xFFu), row = (int)((e >> 8) & 0x7FFu), sc = (int)(e >> 24);
#endif
                const bool ready = i < q2n && row <= nms_hi;
                const bool later = i < q2n && row > nms_hi;
                bool keep = false;
                if (ready) {
                    const uint8_t* a = s_map + ring_off(row - 1) + px; const uint8_t* b = s_map + ring_off(row) + px;
                    const uint8_t* c = s_map + ring_off(row + 1) + px;
                    keep = sc > a[-1] && sc > a[0] && sc > a[1] && sc > b[-1] && sc > b[1] && sc > c[-1] && sc > c[0] && sc > c[1];
                }
                emit(keep, px, row, sc);
                const unsigned long long lm = __builtin_amdgcn_ballot_w64(later);
                if (later) s_q2[kept + __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u))] = e;
                kept += __popcll(lm);
                wave_sync();
            }
            q2n = kept;
        } else {
            // corner list overflowed (pathological image): every pixel of the ready rows looks at the score ring
            for (int row = nms_done + 1; row <= nms_hi; row++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int px = 4 * lane + j;
                    const uint8_t* b = s_map + ring_off(row) + px;
                    const int sc = b[0];
                    bool keep = false;
                    if (sc > 0 && px >= 1 && px <= 254) {
                        const uint8_t* a = s_map + ring_off(row - 1) + px; const uint8_t* c = s_map + ring_off(row + 1) + px;
                        keep = sc > a[-1] && sc > a[0] && sc > a[1] && sc > b[-1] && sc > b[1] && sc > c[-1] && sc > c[0] && sc > c[1];
                    }
                    emit(keep, px, row, sc);
                }
            }
        }
        nms_done = max(nms_done, nms_hi);
        if (outn >= 64) flush_async();        // (the pending batch of the previous group was stored after its rows-in phase)
        wave_sync();
        PHASE(3);
        }      // FB
        // ---- next group's rows out of their registers (mirrored bytes in place), loads of the group after that ----
#pragma unroll
        for (int u = 0; u < kG; u++) {
            cur[u] = __builtin_amdgcn_perm(0u, pre[u], R.sel);
            asm volatile("" : "+v"(cur[u]) : : "memory");      // here, not sunk to the loop end behind this group's stores
        }
#pragma unroll
        for (int u = 0; u < kG; u++) pre[u] = gload_sv<u32_unaligned>(row_ptr(t0 + 2 * kG + u), in_off);
        PHASE(5);
        // the batch whose slice this group's NMS phase reserved: the waits of the rows above have outlasted its atomic (vmcnt
        // counts in order), so the result is read here for free. It must not stay in its register across the loop's back
        // edge: the compiler copies loop-carried values there, and the copy of a pending atomic result is an
        // s_waitcnt vmcnt(0) in EVERY group -- prefetched rows, pyramid stores, everything (3.2k of a group's 13.9k cycles by
        // the phase stamps).
        if constexpr (FB) { if (pend_any) { complete_pending(); } }

        // ---- a6.1 fused: rows of level l+1 whose two source rows are in the ring now. The lane that hosts an output dword
        //      blends its 4 pixels from the ring (v_perm + v_dot2_u32_u16, the integers of k_resize_lds); the horizontal
        //      blend of the lower source row is kept for the next output row, which starts there 5 times out of 6. ----
#if !ARIA_STREAM_PYR_INWALK
        if (pyr) {
            const int ring_last = min(t0 + kG - 1, min(t_last, h - 1));    // last level row in the ring
            constexpr uint32_t put[4] = {0x03020106u, 0x03020600u, 0x03060100u, 0x06020100u};   // byte 2 of v -> byte i
            while (dy_next < gn_h && dy_next < yt_base + 64) {
                dy_next = __builtin_amdgcn_readfirstlane(dy_next);
                const uint32_t ty = (uint32_t)__builtin_amdgcn_readlane((int)ytv, dy_next - yt_base);
                const int oy = (int)(ty & 0xFFFFu);
                const int rb = min(oy + 1, h - 1);
                if (rb > ring_last || oy >= r1) break;
                if (host_gx >= 0) {
                    const uint32_t cy1 = ty >> 16, cyp = (256u - cy1) | (cy1 << 16);
                    const uint8_t* rowa = s_rawb + ring_off(oy);
                    const uint8_t* rowb = s_rawb + ring_off(rb);
                    const bool reuse = oy == hprev_row;                 // wave-uniform
                    // every LDS read of the row is issued before the first use (one round trip, not one per pixel)
                    uint32_t qb0[4], qb1[4], qa0[4] = {0, 0, 0, 0}, qa1[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t* qb = reinterpret_cast<const uint32_t*>(rowb + xo[i]);
                        qb0[i] = qb[0]; qb1[i] = qb[1];
                    }
                    if (!reuse) {
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const uint32_t* qa = reinterpret_cast<const uint32_t*>(rowa + xo[i]);
                            qa0[i] = qa[0]; qa1[i] = qa[1];
                        }
                    }
                    uint32_t outw = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const us2v wx = __builtin_bit_cast(us2v, xw[i]);
                        uint32_t h0 = hprev[i];
                        if (!reuse) h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, __builtin_amdgcn_perm(qa1[i], qa0[i], xs[i])), wx, 0u, false);
                        const uint32_t bot = __builtin_amdgcn_perm(qb1[i], qb0[i], xs[i]);                 // p10 | p11 << 16
                        const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, bot), wx, 0u, false);
                        const uint32_t v = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, h0 | (h1 << 16)),
                                                                  __builtin_bit_cast(us2v, cyp), 32768u, false);   // < 2^24
                        outw = __builtin_amdgcn_perm(v, outw, put[i]);
                        hprev[i] = h1;
                    }
                    gstore_sv<uint32_t>(next0 + (int64_t)dy_next * A.next_pitch, nout_off, outw);
                }
                hprev_row = rb;
                dy_next++;
            }
        }
#endif
        PHASE(4);
    }
    if constexpr (FB) {
        if (pend_any) complete_pending();
        flush_sync(true);
    }
#ifdef ARIA_DIAG
    if (stamps && lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; k++) atomicAdd(stamps + k, ph[k]);
        atomicAdd(stamps + 7, 1ull);
    }
#endif
#undef PHASE
}

// The pyramid step as a wave of its own (ARIA_STREAM_SPLIT): the panel / segment decomposition, lane roles, hosting rule and
// arithmetic of stream_wave's pyramid phase, without anything else -- so that it can be shaped for what it is, a memory-latency-
// bound stream: rows go through a 32-row LDS ring in groups of 16, with TWO groups of row loads in flight (32 registers; the
// FAST/blur wave can afford 14), and a group's ~13 output rows are blended while the next rows arrive. Output rows are owned by
// the segment that holds their upper source row, exactly as in the fused wave: every row of level l + 1 is written once.
__device__ __forceinline__ void pyramid_wave(const StreamArgs& A, uint8_t* __restrict__ wl, const int lane, const LaneRole R,
                                             const int frame0, const int seg) {
    constexpr int GB = 16, RB = 32;
    const int w = A.w, h = A.h;
    const int r0 = seg * A.seg_rows, r1 = min(h, r0 + A.seg_rows);
    const int t_first = r0, t_last = min(r1, h - 1);           // upper source rows r0 .. r1 - 1 and the row below the last one
    uint32_t* s_raw = reinterpret_cast<uint32_t*>(wl);         // [RB][64] dwords (8 KB of the wave's allocation)
    const uint8_t* s_rawb = wl;
    auto roff = [](int y) { return (y & (RB - 1)) << 8; };

    const int pitch_in = A.src_pitch;
    const uint8_t* src0 = A.src + (int64_t)frame0 * A.src_fstride;
    const int r4 = w & 3, D = (w + 3) >> 2;
    int xload;
    if (R.gdw < 0) xload = 1;
    else if (R.gdw >= D) xload = r4 ? w - 9 + r4 : w - 5;
    else if (r4 && R.gdw == D - 1) xload = w - 4;
    else xload = 4 * R.gdw;
    const uint32_t in_off = (uint32_t)((int64_t)(R.frame - frame0) * A.src_fstride + xload);

    const int gn_w = A.next_w, gn_h = A.next_h;
    int host_gx = -1;
    uint32_t xw[4] = {0, 0, 0, 0}, xo[4] = {0, 0, 0, 0}, xs[4] = {0, 0, 0, 0};
    uint32_t nout_off = 0;
    const uint32_t* yt = A.yt;
    uint8_t* next0 = A.next + (int64_t)frame0 * A.next_fstride;
    if (R.owner) {
        const uint32_t hv = A.xinv[R.gdw];
        if (hv != 0xFFFFFFFFu) host_gx = (int)hv;
    }
    if (host_gx >= 0) {
        const uint32_t* xt = A.xt;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t t = xt[min(4 * host_gx + i, gn_w - 1)];
            const int ox = (int)(t & 0xFFFFu);
            const uint32_t cx1 = t >> 16;
            const int lpx = 4 * lane + (ox - 4 * R.gdw);
            xw[i] = (256u - cx1) | (cx1 << 16);
            xo[i] = (uint32_t)(lpx & ~3);
            xs[i] = 0x0C010C00u + (uint32_t)(lpx & 3) * 0x00010001u;
        }
        nout_off = (uint32_t)((int64_t)(R.frame - frame0) * A.next_fstride + 4 * host_gx);
    }
    int dy_next = 0;
    if (r0 > 0) {
        dy_next = min(max((int)(((int64_t)(2 * r0 + 1) * gn_h) / (2 * h)) - 1, 0), gn_h - 1);
        while (dy_next > 0 && (int)(yt[dy_next - 1] & 0xFFFFu) >= r0) dy_next--;
        while (dy_next < gn_h && (int)(yt[dy_next] & 0xFFFFu) < r0) dy_next++;
        dy_next = __builtin_amdgcn_readfirstlane(dy_next);
    }
    // The y-table word of an output row is wave-uniform: it comes through the SCALAR cache, one row ahead of its use (s_load:
    // counted on lgkmcnt with the LDS reads the row waits for anyway). As a vector load it sat on vmcnt between the row reloads,
    // and the compiler waited for it with vmcnt(0) -- i.e. for the reloads just issued, the whole memory latency per group.
    typedef const uint32_t __attribute__((address_space(4))) * const_u32p;       // constant address space: a uniform index is an s_load
    const const_u32p yts = reinterpret_cast<const_u32p>(reinterpret_cast<uint64_t>(yt));
    uint32_t ty_n = yts[min(dy_next, gn_h - 1)];
    uint32_t hprev[4] = {0u, 0u, 0u, 0u};
    int hprev_row = -1;

    auto row_ptr = [&](int t) -> const uint8_t* { return src0 + (int64_t)min(max(t, 0), t_last) * pitch_in; };
    uint32_t b0[GB], b1[GB];
#pragma unroll
    for (int u = 0; u < GB; u++) b0[u] = gload_sv<u32_unaligned>(row_ptr(t_first + u), in_off);
#pragma unroll
    for (int u = 0; u < GB; u++) b1[u] = gload_sv<u32_unaligned>(row_ptr(t_first + GB + u), in_off);
    const int G = (t_last - t_first + 1 + GB - 1) / GB;

    // one group: its rows (mirrored bytes in place) into the ring, the loads of the group after next into the freed
    // registers, then every output row whose two source rows are in the ring now
    auto group = [&](uint32_t (&buf)[GB], int gi) {
        const int t0 = t_first + GB * gi;
#pragma unroll
        for (int u = 0; u < GB; u++) s_raw[(((t0 + u) & (RB - 1)) << 6) + lane] = __builtin_amdgcn_perm(0u, buf[u], R.sel);
#pragma unroll
        for (int u = 0; u < GB; u++) buf[u] = gload_sv<u32_unaligned>(row_ptr(t0 + 2 * GB + u), in_off);
        wave_sync();
        const int ring_last = min(t0 + GB - 1, t_last);
        constexpr uint32_t put[4] = {0x03020106u, 0x03020600u, 0x03060100u, 0x06020100u};
        while (dy_next < gn_h) {
            dy_next = __builtin_amdgcn_readfirstlane(dy_next);
            const uint32_t ty = ty_n;
            const int oy = (int)(ty & 0xFFFFu);
            const int rb = min(oy + 1, h - 1);
            if (rb > ring_last || oy >= r1) break;
            ty_n = yts[min(dy_next + 1, gn_h - 1)];
            if (host_gx >= 0) {
                const uint32_t cy1 = ty >> 16, cyp = (256u - cy1) | (cy1 << 16);
                const uint8_t* rowa = s_rawb + roff(oy);
                const uint8_t* rowb = s_rawb + roff(rb);
                const bool reuse = oy == hprev_row;
                uint32_t qb0[4], qb1[4], qa0[4] = {0, 0, 0, 0}, qa1[4] = {0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t* qb = reinterpret_cast<const uint32_t*>(rowb + xo[i]);
                    qb0[i] = qb[0]; qb1[i] = qb[1];
                }
                if (!reuse) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t* qa = reinterpret_cast<const uint32_t*>(rowa + xo[i]);
                        qa0[i] = qa[0]; qa1[i] = qa[1];
                    }
                }
                uint32_t outw = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const us2v wx = __builtin_bit_cast(us2v, xw[i]);
                    uint32_t h0 = hprev[i];
                    if (!reuse) h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, __builtin_amdgcn_perm(qa1[i], qa0[i], xs[i])), wx, 0u, false);
                    const uint32_t bot = __builtin_amdgcn_perm(qb1[i], qb0[i], xs[i]);
                    const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, bot), wx, 0u, false);
                    const uint32_t v = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, h0 | (h1 << 16)), __builtin_bit_cast(us2v, cyp), 32768u, false);
                    outw = __builtin_amdgcn_perm(v, outw, put[i]);
                    hprev[i] = h1;
                }
                gstore_sv<uint32_t>(next0 + (int64_t)dy_next * A.next_pitch, nout_off, outw);
            }
            hprev_row = rb;
            dy_next++;
        }
        wave_sync();            // the next group's ring writes must not pass this group's ring reads
    };
    // (both calls unconditional: behind a condition the second buffer's reload is not a certain event for the compiler's
    // wait-count bookkeeping, and the first buffer's rows are then taken out with vmcnt(0) -- the whole memory latency, every
    // other group. A surplus group past the segment's end loads its last row again and finds no output row to produce.)
    for (int gi = 0; gi < G; gi += 2) {
        group(b0, gi);
        group(b1, gi + 1);
    }
}

// FOUR waves per SIMD (round 4). The kernel needs <= 128 VGPRs and <= 10 240 B of LDS per wave for that; the compact lists give
// the LDS, and with the flat walk, the pending batch left in the out-list and the y-table through a scalar base the allocator gets to 128
// registers when it is told to (left alone it stops at 138-140: it has no reason to go below the 168 of three waves). What it
// still spills are five values that live through the group loop without being used in it: stored once in front of the
// loop, loaded once behind it (24 bytes of scratch per lane, no scratch access inside the loop -- tests/test_isa_lint.py holds
// the kernel to that). Measured A B A B on one box (tools/ab_levels.sh, 640x480 at 8192 frames): 1.481 against 1.581 us per
// frame, 0.2157 / 0.2147 against 0.1995 / 0.2012 of 8 TB/s. ARIA_STREAM_WAVES4=0 restores the unconstrained allocation.
#ifndef ARIA_STREAM_WAVES4
#define ARIA_STREAM_WAVES4 1
#endif
#if ARIA_STREAM_WAVES4
static_assert(ARIA_STREAM_COMPACT_LDS, "four waves per SIMD need the 10 240-byte LDS layout");
#define ARIA_STREAM_OCC __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define ARIA_STREAM_OCC
#endif
template <int WPB>
__global__ __launch_bounds__(64 * WPB) ARIA_STREAM_OCC void k_fast_blur_stream(StreamArgs A, unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int w = A.w, h = A.h;
    const int D = (w + 3) >> 2, U = D + 2;
    int wave_id = (int)blockIdx.x * WPB + wv;
    // ARIA_STREAM_SPLIT: ids come in blocks of 16 -- eight FAST/blur waves (one per XCD: workgroups go round-robin to the XCDs by
    // id), then the eight pyramid waves of the same panels -- so both jobs are spread evenly over the XCDs and the panel map
    // below (XCD k gets a contiguous eighth of the panels) holds for each of them
    bool role_pyr = false;
    if (ARIA_STREAM_SPLIT && A.next != nullptr) {
        // (which of a block's two rows of eight is the pyramid row follows the Thue-Morse sequence of the block index: with a
        // plain alternation the XCD's k-th and (k+1)-th workgroup -- which its dispatcher deals to its CUs in turn -- were always
        // one of each kind, every FAST/blur wave landed on the same half of the CUs and the launch took twice as long)
        role_pyr = (((wave_id >> 3) ^ __popc(wave_id >> 4)) & 1) != 0;
        wave_id = ((wave_id >> 4) << 3) | (wave_id & 7);
    }
    // Workgroups go round-robin to the 8 XCDs (each with its own L2) by linear id. Neighbouring panels share the 128-byte
    // lines their 248-byte row pieces end in (rows are not cut at line boundaries), so XCD k is given a CONTIGUOUS eighth of
    // the panels: the shared lines are then fetched, and the partial lines written, by one L2 (before: 1.7x the level bytes
    // fetched and 1.2x written, FETCH_SIZE / WRITE_SIZE). A.panels8 = ceil(panels / 8).
    const int per_seg = 8 * A.panels8;
    const int seg = wave_id / per_seg, rr = wave_id - seg * per_seg;
    const int pw = (rr & 7) * A.panels8 + (rr >> 3);
    if (seg >= A.n_seg || pw >= A.panels) return;        // no barrier anywhere in this kernel: a wave may leave
    const int64_t total = (int64_t)A.n_frames * U;
    const int64_t pos0 = (int64_t)pw * kOwned - 1;
    int64_t pos = pos0 + lane;
    const bool valid = pos >= 0 && pos < total;
    pos = min(max(pos, (int64_t)0), total - 1);
    LaneRole R;
    R.frame = (int)(pos / U);
    const int j = (int)(pos - (int64_t)R.frame * U);
    R.gdw = j - 1;
    const int frame0 = __builtin_amdgcn_readfirstlane(R.frame);
    const int r4 = w & 3;
    if (R.gdw < 0 || R.gdw >= D) R.sel = 0x00010203u;                                  // halo: mirrored dword
    else if (r4 && R.gdw == D - 1) R.sel = r4 == 1 ? 0x00010203u : r4 == 2 ? 0x01020302u : 0x02030201u;
    else R.sel = 0x03020100u;
    const bool data = valid && R.gdw >= 0 && R.gdw < D;
    R.owner = data && lane >= 1 && lane <= kOwned;
    const bool level_has_kp = (w > 2 * kEdgeThreshold) && (h > 2 * kEdgeThreshold);
    const int fx0 = kEdgeThreshold - 1, fx1 = w - kEdgeThreshold;                      // inclusive FAST range
    R.xm0 = R.xm1 = 0;
    R.tailbits = 0;
    if (data) {
        const int x = 4 * R.gdw;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // lane 0 and lane 63 are neighbour providers: only their pixel next to an owned lane is scored (NMS needs it)
            const bool scored = (lane >= 1 && lane <= kOwned) || (lane == 0 && i == 3) || (lane == 63 && i == 0);
            if (level_has_kp && scored && x + i >= fx0 && x + i <= fx1) {
                if (i < 2) R.xm0 |= (i & 1) ? 0x80000000u : 0x00008000u;
                else R.xm1 |= (i & 1) ? 0x80000000u : 0x00008000u;
            }
            if (x + i >= A.tail_start) R.tailbits |= 1u << i;
        }
    }
    uint8_t* wl = smem + wv * kWaveLds;
    const bool any_tail = __builtin_amdgcn_ballot_w64(R.tailbits != 0) != 0;
#if ARIA_STREAM_SPLIT
#ifdef ARIA_PROBE_SPLIT_EMPTY_PYR
    if (role_pyr) return;        // timing probe: what do the pyramid waves cost when they do nothing?
#endif
    if (role_pyr) pyramid_wave(A, wl, lane, R, frame0, seg);
    else if (any_tail) stream_wave<WPB, true, true, false>(A, wl, lane, R, frame0, seg, stamps);
    else stream_wave<WPB, false, true, false>(A, wl, lane, R, frame0, seg, stamps);
#else
    (void)role_pyr;
#if ARIA_STREAM_PYR_INWALK
    // (the last level has no pyramid step: its walk is an instance of its own, without the blend)
    if (A.next == nullptr) {
        if (any_tail) stream_wave<WPB, true, true, false>(A, wl, lane, R, frame0, seg, stamps);
        else stream_wave<WPB, false, true, false>(A, wl, lane, R, frame0, seg, stamps);
        return;
    }
#endif
    if (any_tail) stream_wave<WPB, true, true, true>(A, wl, lane, R, frame0, seg, stamps);
    else stream_wave<WPB, false, true, true>(A, wl, lane, R, frame0, seg, stamps);
#endif
}

// The batch path takes this kernel when every level is at least 16 px wide and high (tiny images stay with the band
// kernel). Source rows need no alignment: every lane's row load is a dword load of any byte address (the mirrored halo
// loads are unaligned by construction).
bool stream_eligible(const Plan& P, const FrameSrc& S) {
    (void)S;
    for (int l = 0; l < kLevels; l++)
        if (P.lv[l].w < 16 || P.lv[l].h < 16) return false;
    return P.stream_ok != 0;
}

int stream_set_attributes() {
    ARIA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fast_blur_stream<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    ARIA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fast_blur_stream<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    ARIA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fast_blur_stream<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    return ARIA_OK;
}

// Column from which the column filter's ties round UP (the scalar tail of OpenCV's SIMD loop), by blur_tie_mode:
// 0 = everywhere, 1 = x >= w & ~3 (4-lane vectors), 2 = w & ~7, 3 = w & ~15.
int blur_tail_start(int tie_mode, int w) {
    switch (tie_mode) {
        case 0: return 0;
        case 2: return w & ~7;
        case 3: return w & ~15;
        default: return w & ~3;
    }
}

void launch_fast_blur_stream(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st,
                             Profiler* prof, LaunchCtx& ctx) {
#ifdef ARIA_DIAG
    static const bool want_stamps = [] { const char* e = aria_getenv("ARIA_STREAM_STAMPS"); return e && e[0] == '1'; }();
#else
    const bool want_stamps = false;
#endif
    static unsigned long long* d_stamps = nullptr;
    if (want_stamps && !d_stamps) hipMalloc(&d_stamps, 8 * sizeof(unsigned long long));
    (void)ctx;
    static const int seg_force = [] { const char* e = aria_getenv("ARIA_STREAM_SEG_ROWS"); return e ? atoi(e) : 0; }();    // variants build
    static const int wpb = [] { const char* e = aria_getenv("ARIA_STREAM_WPB"); const int v = e ? atoi(e) : 1; return (v == 2 || v == 4) ? v : 1; }();     // one wave per workgroup:
    // the waves share nothing, and single-wave workgroups fill the gaps the matcher's workgroups leave (293.6k frames/s
    // against 288.7k with four waves per workgroup)
    for (int l = 0; l < kLevels; l++) {
        const LevelGeom& g = P.lv[l];
        const int U = ((g.w + 3) >> 2) + 2;
        const int64_t total = (int64_t)n_frames * U;
        const int64_t panels = (total + kOwned - 1) / kOwned;        // panel k owns virtual dwords 62 k .. 62 k + 61
        // row segments: enough waves for ~6 rounds on the chip (256 CUs x 12 waves), none shorter than 64 rows (each
        // segment re-ingests 7 rows of its neighbours)
        int n_seg = (int)std::min<int64_t>(std::max<int64_t>((6 * 256 * 12 + panels - 1) / panels, 1), std::max(g.h / 64, 1));
        if (seg_force > 0) n_seg = std::min(std::max(g.h / seg_force, 1), 64);
        const int seg_rows = ((g.h + n_seg - 1) / n_seg + 3) & ~3;      // whole row quads: the lines of a quad of the blurred level are written by one wave
        n_seg = (g.h + seg_rows - 1) / seg_rows;
        const int64_t panels8 = (panels + 7) / 8;
        // (split build: as many pyramid waves again, interleaved in blocks of eight, for every level that has a next one)
        const int64_t waves = 8 * panels8 * n_seg * ((ARIA_STREAM_SPLIT && l + 1 < kLevels) ? 2 : 1);
        const dim3 grid((unsigned)((waves + wpb - 1) / wpb));
        StreamArgs A{};
        A.seg_rows = seg_rows; A.n_seg = n_seg; A.panels = (int)panels; A.panels8 = (int)panels8;
        if (l == 0) { A.src = S.img; A.src_fstride = S.frame_stride; A.src_pitch = S.row_stride; }
        else { A.src = D.raw + g.raw_off; A.src_fstride = P.raw_frame_bytes; A.src_pitch = g.pitch; }
        A.blur = D.blur + g.blur_off; A.blur_fstride = P.blur_frame_bytes; A.blur_pitch = g.pitch;
        A.cand = D.cand + g.cand_off; A.cand_fstride = P.cand_frame_entries; A.cand_cap = g.cand_cap;
        A.cand_cnt = D.cand_cnt + l; A.err = D.err;
        A.w = g.w; A.h = g.h; A.n_frames = n_frames; A.thr = P.fast_threshold; A.tail_start = blur_tail_start(P.tie_mode, g.w);
        if (l + 1 < kLevels) {
            const LevelGeom& gn = P.lv[l + 1];
            A.next = D.raw + gn.raw_off; A.next_fstride = P.raw_frame_bytes; A.next_pitch = gn.pitch; A.next_w = gn.w; A.next_h = gn.h;
            A.xt = D.tab + gn.xtab; A.yt = D.tab + gn.ytab; A.xinv = D.tab + gn.xinv; A.ytr = D.tab + gn.ytr;
        } else {
            A.ytr = D.tab + P.lv[1].ytr;          // (never used for a store: no lane hosts anything without a next level)
        }
        // (variants build: ARIA_STREAM_LDS_KB pads the workgroup's LDS to lower the occupancy -- how the kernel scales with waves per SIMD)
        static const size_t lds_pad = [] { const char* e = aria_getenv("ARIA_STREAM_LDS_KB"); return e ? (size_t)atoi(e) * 1024 : (size_t)0; }();
#define ARIA_FS_LAUNCH(N) ARIA_LAUNCH(prof, (k_fast_blur_stream<N>), grid, dim3(64 * N), std::max((size_t)N * kWaveLds, lds_pad), st, A, d_stamps)
        if (d_stamps) hipMemsetAsync(d_stamps, 0, 8 * sizeof(unsigned long long), st);
        if (wpb == 1) ARIA_FS_LAUNCH(1); else if (wpb == 2) ARIA_FS_LAUNCH(2); else ARIA_FS_LAUNCH(4);
#undef ARIA_FS_LAUNCH
        if (d_stamps) {
            hipStreamSynchronize(st);
            unsigned long long hs[8];
            hipMemcpy(hs, d_stamps, sizeof(hs), hipMemcpyDeviceToHost);
            const double n = (double)std::max(1ull, hs[7]);
            fprintf(stderr, "[stream stamps L%d] %llu waves, ticks per wave: ring %.0f walk %.0f append+score %.0f rows-in %.0f nms+emit %.0f pyramid %.0f\n",
                    l, hs[7], hs[0] / n, hs[1] / n, hs[2] / n, hs[5] / n, hs[3] / n, hs[4] / n);
        }
    }
}

}  // namespace aria
