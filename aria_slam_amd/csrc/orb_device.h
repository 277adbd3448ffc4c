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

// Global accesses as "wave-uniform 64-bit base in SGPRs + the lane's 32-bit offset" (global_load/store v_off, s[base]).
// Left alone the compiler folds the lane's constant offset into a 64-bit VGPR base and pays a v_mad_i64_i32 (and a VGPR
// pair) per row load and store; the empty asm pins the row base to the scalar unit (243 -> 69 of them in the object; the
// Q4 build 1.739 -> 1.703 us per frame), the address space keeps the access global (an integer turned pointer would be a
// FLAT access, which also counts on lgkmcnt). (Round 4: also pinning the lane offset with an empty "+v" asm keeps its
// zero-extension in the block, and every row load / store then takes the saddr form -- global_store_dword v_off, v, s[b:b+1]
// -- instead of a v_lshl_add_u64 per access: 147 VGPRs, no faster: 0.2015 / 0.1992 against 0.2025 / 0.2009 A B A B.)
template <typename T>
__device__ __forceinline__ T gload_sv(const uint8_t* base, uint32_t off) {
    uint64_t b = reinterpret_cast<uint64_t>(base);
    asm volatile("" : "+s"(b));
    typedef const T __attribute__((address_space(1))) * gp;
    return *reinterpret_cast<gp>(b + off);
}
template <typename T>
__device__ __forceinline__ void gstore_sv(uint8_t* base, uint32_t off, T v) {
    uint64_t b = reinterpret_cast<uint64_t>(base);
    asm volatile("" : "+s"(b));
    typedef T __attribute__((address_space(1))) * gp;
    *reinterpret_cast<gp>(b + off) = v;
}

__device__ __forceinline__ int reflect101(int i, int n) {
    // BORDER_REFLECT_101; inputs here never lie more than one period outside, clamp guards tiny levels
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return min(max(i, 0), n - 1);
}

// fast_score.cpp cornerScore<16> from the centre pixel v and its 16 ring pixels (in ring order): max over the 16 nine-arcs of
// min(v - ring) and of min(ring - v), minus 1; the pixel is a FAST-9 corner for threshold t iff the result is >= t.
// Both polarities ride in one register as a packed pair of HALF FLOATS (round 4): a byte b in the low bits of a half (0x00bb)
// is the subnormal b * 2^-24 -- the subnormals and the first normal binade of binary16 are one linear ramp up to 0x07ff --
// so v_pk_add_f16 of (v | ring << 16) and its half-swapped, negated self is exactly (v - ring, ring - v) * 2^-24 (|d| <= 255:
// exact; HIP kernels run with half denormals enabled), and gfx950 has THREE-input packed minimum / maximum for halves
// (v_pk_minimum3_f16 / v_pk_maximum3_f16, the issue class of v_pk_min_i16: profiles/r2_valu_issue_rates.txt): the nine-arc
// minima are min3 of min3 (32 instructions instead of 64 two-input ones) and the maximum over the 16 arcs takes 7 instead
// of 16. A non-negative result's bit pattern IS the integer; negative ones (sign-magnitude) read as negative int16, and a
// negative score is never a corner. Used by k_fast_blur_stream (batches) and k_fast_blur_band (single frames).
#ifndef ARIA_SCORE_PAIRED
#define ARIA_SCORE_PAIRED 1
#endif
#ifndef ARIA_SCORE_SEQ
#define ARIA_SCORE_SEQ 0
#endif
__device__ __forceinline__ int fast9_score_f16(uint32_t v, const uint32_t (&rg)[16]) {
#if ARIA_SCORE_PAIRED
    // One polarity, TWO ring points per register: R_j = (v - ring_j, v - ring_{j+8}), j = 0..7 -- one v_lshl_or and one
    // v_pk_add_f16 per PAIR of ring points (17 instructions to build instead of 32). Ring point j + 8 is the other half of
    // register j, so "R_{j+8}" is R_j with its halves swapped, which the packed instructions take as an operand modifier
    // (op_sel): the nine-arc minima A_k = min(D_k .. D_{k+8}) come out as M9_j = (A_j, A_{j+8}) from 8 + 8 three-input
    // minima, the nine-arc maxima likewise. darker score = max_k A_k, brighter score = -min_k (max arc): 8 more.
    const uint32_t vv = v | (v << 16);
    uint32_t R[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t x = rg[j] | (rg[j + 8] << 16);
        asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(R[j]) : "v"(vv), "v"(x));
    }
    // three-input min / max of registers i0, i1, i2 of an 8-register ring whose indices >= 8 mean "halves swapped"
#define ARIA_PK3(OP, dst, src, i0, i1, i2)                                                                                       \
    do {                                                                                                                         \
        if ((i2) < 8) asm(OP " %0, %1, %2, %3" : "=v"(dst) : "v"(src[(i0) & 7]), "v"(src[(i1) & 7]), "v"(src[(i2) & 7]));          \
        else if ((i1) < 8) asm(OP " %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(dst) : "v"(src[(i0) & 7]), "v"(src[(i1) & 7]), "v"(src[(i2) & 7])); \
        else asm(OP " %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "=v"(dst) : "v"(src[(i0) & 7]), "v"(src[(i1) & 7]), "v"(src[(i2) & 7])); \
    } while (0)
    uint32_t m3[8], m9[8], x3[8], x9[8];
    uint32_t da, db, dk, ba, bb, br;
#if ARIA_SCORE_SEQ
    // the darker tree first, then the brighter one (the empty asm keeps the compiler from interleaving them: 8 registers of
    // arcs in flight instead of 16 -- 5 VGPRs of the kernel's peak, which sits in this phase)
#pragma unroll
    for (int j = 0; j < 8; j++) ARIA_PK3("v_pk_minimum3_f16", m3[j], R, j, j + 1, j + 2);
#pragma unroll
    for (int j = 0; j < 8; j++) ARIA_PK3("v_pk_minimum3_f16", m9[j], m3, j, j + 3, j + 6);
#else
#pragma unroll
    for (int j = 0; j < 8; j++) { ARIA_PK3("v_pk_minimum3_f16", m3[j], R, j, j + 1, j + 2); ARIA_PK3("v_pk_maximum3_f16", x3[j], R, j, j + 1, j + 2); }
#pragma unroll
    for (int j = 0; j < 8; j++) { ARIA_PK3("v_pk_minimum3_f16", m9[j], m3, j, j + 3, j + 6); ARIA_PK3("v_pk_maximum3_f16", x9[j], x3, j, j + 3, j + 6); }
#endif
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(da) : "v"(m9[0]), "v"(m9[1]), "v"(m9[2]));
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(db) : "v"(m9[3]), "v"(m9[4]), "v"(m9[5]));
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(da) : "v"(da), "v"(m9[6]), "v"(m9[7]));
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(dk) : "v"(da), "v"(db));                       // (max A_j over even half, over odd half)
#if ARIA_SCORE_SEQ
#pragma unroll
    for (int j = 0; j < 8; j++) asm volatile("" : "+v"(R[j]) : "v"(dk));
#pragma unroll
    for (int j = 0; j < 8; j++) ARIA_PK3("v_pk_maximum3_f16", x3[j], R, j, j + 1, j + 2);
#pragma unroll
    for (int j = 0; j < 8; j++) ARIA_PK3("v_pk_maximum3_f16", x9[j], x3, j, j + 3, j + 6);
#endif
#undef ARIA_PK3
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(ba) : "v"(x9[0]), "v"(x9[1]), "v"(x9[2]));
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(bb) : "v"(x9[3]), "v"(x9[4]), "v"(x9[5]));
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(ba) : "v"(ba), "v"(x9[6]), "v"(x9[7]));
    asm("v_pk_min_f16 %0, %1, %2" : "=v"(br) : "v"(ba), "v"(bb));
    // Q = max(dk, -br) per half: the sign flip of a half float is its top bit
    uint32_t Q;
    asm("v_pk_max_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(Q) : "v"(dk), "v"(br));
    const int q0 = (int)(short)(Q & 0xFFFFu), q1 = (int)(short)(Q >> 16);
    return max(q0, q1) - 1;
#else
    uint32_t Pk[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t x = v | (rg[k] << 16);
        asm("v_pk_add_f16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(Pk[k]) : "v"(x));
    }
    uint32_t M3[16], M9[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
        asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(M3[k]) : "v"(Pk[k]), "v"(Pk[(k + 1) & 15]), "v"(Pk[(k + 2) & 15]));
#pragma unroll
    for (int k = 0; k < 16; k++)
        asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(M9[k]) : "v"(M3[k]), "v"(M3[(k + 3) & 15]), "v"(M3[(k + 6) & 15]));
    uint32_t A5[6];
#pragma unroll
    for (int k = 0; k < 5; k++)
        asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(A5[k]) : "v"(M9[3 * k]), "v"(M9[3 * k + 1]), "v"(M9[3 * k + 2]));
    A5[5] = M9[15];
    uint32_t B0, B1, Q;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(B0) : "v"(A5[0]), "v"(A5[1]), "v"(A5[2]));
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(B1) : "v"(A5[3]), "v"(A5[4]), "v"(A5[5]));
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(Q) : "v"(B0), "v"(B1));
    const int q0 = (int)(short)(Q & 0xFFFFu), q1 = (int)(short)(Q >> 16);
    return max(q0, q1) - 1;
#endif
}

}  // namespace aria
