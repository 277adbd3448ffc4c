// Brute-force Hamming kNN-2 on the gfx950 matrix cores.
// Same semantics as k_knn2 in match_hip.hip (reference: src/adapters/gpu/CudaMatcher.cpp:28-68 and
// src/legacy/LoopClosure.cpp:72-114; CPU cv::BFMatcher order: the two smallest (distance, train index) pairs).
//
// The 2000 x 2000 x 256-bit distance table of one frame pair is the one GEMM-shaped piece of this path -- it is
// compute-bound (160 KB in, 4M distances), not HBM-bound -- and for bit vectors
//     hamming(q, t) = popcount(q) + popcount(t) - 2 |q & t|,
// where |q & t| (the number of common set bits) is an exact int8 x int8 -> int32 matrix product once every bit is
// widened to a byte: v_mfma_i32_32x32x32_i8, one instruction = 32 trains x 32 queries x 32 bits in 32 cycles, against
// 8 x (v_xor + v_bcnt) per single pair on the vector ALU. The bytes are scaled (train bit -> -128, query bit -> 64)
// and the accumulator is preloaded with 4096 popcount(t) + train index, so that the MFMA itself delivers the sortable
// key  4096 (popcount(t) - 2 |q & t|) + index  and the vector ALU is left with exactly the running top-2 (two
// v_med3 per distance). All arithmetic is integer: bit-identical to the VALU kernel.
// Why single-bit bytes (0x80 / 0x40 / 0) and not +-1 or +-127: the matrix pipe's clock gives way under load by
// operand toggle rate (tools/microbench/mfma_i8_power.hip: 16.3 ns per MFMA per SIMD on zeros, 17.2 on single-bit
// bytes, 20.4 on +-64 x +-127, 22.1 on random bytes) -- the sparse encoding is worth 15 % of the kernel.
//
// Workgroup = 4 waves; a wave keeps the widened fragments of its 32*NC queries (NC column tiles, 32 VGPRs each)
// resident and walks the train set in tiles of 64 that the workgroup widens once into LDS (double-buffered).
// A = trains (rows), B = queries (columns): the accumulator then has the query on the lane and 16 trains in the
// registers, so the top-2 is a per-lane chain with no cross-lane traffic until the very end.
// The k order inside a fragment does not matter (a dot product), only that A and B split K the same way, which the
// symmetric A/B lane maps guarantee: lane l carries row/column l & 31 and K-half l >> 5.
//
// Round 3: train sets up to 4096 descriptors (every SLAM frame) take the FP4 matrix path instead -- k_knn2_fp4 below, one
// E2M1 value per descriptor bit, v_mfma_f32_32x32x64_f8f6f4: twice the bits per instruction at the same instruction time,
// exact in its fp32 accumulators, 0.78 -> 0.51 us per 2000 x 2000 pair. The int8 kernel stays for larger train sets (its
// 16-bit index layout) and as the reference form in the variants build.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "common.h"
#include "match_kernels.h"

namespace aria {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kRowB = 272;           // LDS bytes per widened train: 256 + 16 (bank spread for the ds_read_b128 fragments)
constexpr int kNone = 0x70000000;    // accumulator preload of a train row past the end; any key >= kNoneMin is "no train"
constexpr int kNoneMin = 0x60000000;
constexpr int kKeyBias = 0x00800000; // keys live in [kKeyBias, 2^31): as IEEE bit patterns they are positive NORMAL floats,
                                     // whose order is the integer order -- the top-2 then runs on v_med3_f32, which the
                                     // compiler knows (MFMA hazards, scheduling groups); there is no integer med3 builtin

// Key layouts (c = |q & t|, pt = popcount(t); popcount(q) is added at the very end).
// NARROW (train sets up to 4096 descriptors -- every SLAM frame): train bit -> 0x80 (-128), query bit -> 0x40 (64),
//   product sum = -8192 c; preload = 4096 (pt + 256) + index + bias  ->  key = 4096 (pt - 2c + 256) + index + bias.
// WIDE (up to 65535): query bit -> 0x01, product sum = -128 c; key = (sum << 10) + ((pt + 256) << 16 | index) + bias,
//   one extra v_lshl_add per distance.
constexpr int kNarrowMax = 4096;

// 4 descriptor bits -> 4 bytes with bit k in byte k (the partial products n << 7k never overlap, so no carries)
__device__ __forceinline__ int spread4(uint32_t n, uint32_t mul, uint32_t mask) { return (int)((n * mul) & mask); }
template <bool WIDE>
__device__ __forceinline__ v4i widen_train16(uint32_t hw) {      // set bit -> 0x80
    v4i r;
    r.x = spread4(hw & 15u, 0x10204080u, 0x80808080u);
    r.y = spread4((hw >> 4) & 15u, 0x10204080u, 0x80808080u);
    r.z = spread4((hw >> 8) & 15u, 0x10204080u, 0x80808080u);
    r.w = spread4((hw >> 12) & 15u, 0x10204080u, 0x80808080u);
    return r;
}
template <bool WIDE>
__device__ __forceinline__ v4i widen_query16(uint32_t hw) {      // set bit -> 0x40 (NARROW) / 0x01 (WIDE)
    const uint32_t mul = WIDE ? 0x00204081u : 0x08102040u, mask = WIDE ? 0x01010101u : 0x40404040u;
    v4i r;
    r.x = spread4(hw & 15u, mul, mask);
    r.y = spread4((hw >> 4) & 15u, mul, mask);
    r.z = spread4((hw >> 8) & 15u, mul, mask);
    r.w = spread4((hw >> 12) & 15u, mul, mask);
    return r;
}

// FP4 path (round 3): 8 descriptor bits -> 8 nibbles (bit k -> nibble k), `val` (an E2M1 code) where the bit is set.
// v_mfma_f32_32x32x64_f8f6f4 with E2M1 operands takes 64 bits of K per instruction in the time the int8 form takes 32
// (tools/microbench/mfma_fp4_bits.hip: 17.3 ns per instruction per SIMD on bit patterns, exact results).
__device__ __forceinline__ int spread8(uint32_t b, uint32_t val) {
    uint32_t x = b & 0xFFu;
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return (int)(x * val);
}
__device__ __forceinline__ v4i widen32_fp4(uint32_t w, uint32_t val) {
    v4i r;
    r.x = spread8(w, val); r.y = spread8(w >> 8, val); r.z = spread8(w >> 16, val); r.w = spread8(w >> 24, val);
    return r;
}

// sum over the aligned group of 8 lanes (DPP: xor 1, xor 2 inside the quad, then the half-row mirror)
__device__ __forceinline__ int sum8(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
    return v;
}

// (m1 <= m2) <- the two smallest of {m1, m2, key}; all three are positive normal floats by bit pattern, so
// min(a, b) = med3(a, b, +0) and no NaN/denormal rule is ever involved (fminf would add a canonicalising v_max)
__device__ __forceinline__ void top2_update(float& m1, float& m2, int key) {
    const float k = __int_as_float(key);
    const float n2 = __builtin_amdgcn_fmed3f(m1, m2, k);
    m1 = __builtin_amdgcn_fmed3f(m1, k, 0.0f);
    m2 = n2;
}

// key -> (distance << 16 | train index), the form the rest of the matcher uses
template <bool WIDE>
__device__ __forceinline__ uint32_t final_key(int k, int pq) {
    if (k >= kNoneMin) return 0xFFFFFFFFu;
    const uint32_t u = (uint32_t)(k - kKeyBias);
    if (WIDE) return u + ((uint32_t)(pq - 256) << 16);
    return (((u >> 12) + (uint32_t)(pq - 256)) << 16) | (u & 4095u);
}

template <int MODE, bool WIDE, int NC, int TT>   // NC column tiles (of 32 queries) per wave: workgroup = 128 NC queries;
                                                 // TT trains per staged tile (= per barrier)
__device__ __forceinline__ void knn2_body(
    const uint8_t* __restrict__ q, const int* __restrict__ nq_arr, int nq_fixed, const uint8_t* __restrict__ t,
    const int* __restrict__ nt_arr, int nt_fixed, int64_t q_stride, int64_t t_stride, uint2* __restrict__ keys, int maxq,
    double ratio, int* __restrict__ good, int tsplit) {
    __shared__ __attribute__((aligned(16))) uint8_t s_a[2][TT * kRowB];
    __shared__ __attribute__((aligned(16))) int s_base[2][TT];    // [buf][train of the tile]: accumulator preload
    __shared__ int s_cnt;
    constexpr int QB = 128 * NC;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = lane & 31, hh = lane >> 5;
    // tsplit > 0 (one pair, latency schedule): blockIdx.y is a slice of tsplit train tiles of pair 0 and writes its own
    // row of keys (global train indices); the consumer merges the rows (k_ratio_compact)
    const int pair = tsplit ? 0 : blockIdx.y;
    const int nq = nq_arr ? nq_arr[pair] : nq_fixed;
    const int nt = nt_arr ? nt_arr[pair] : nt_fixed;
    if ((int)blockIdx.x * QB >= nq) return;
    const uint4* qp = reinterpret_cast<const uint4*>(q + (int64_t)pair * q_stride);
    const uint32_t* tw = reinterpret_cast<const uint32_t*>(t + (int64_t)pair * t_stride);

    // resident B fragments: column tile c, k-step s = bits [32 s + 16 hh, +16) of query q0 + 32 c + col
    const int q0 = blockIdx.x * QB + wv * (32 * NC);
    v4i B[NC][8];
    int pq[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int qi = min(q0 + 32 * c + col, nq - 1);      // rows past the end repeat the last query, never stored
        const uint4 lo = qp[2 * qi], hi = qp[2 * qi + 1];
        const uint32_t dw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        int pc = 0;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            pc += __popc(dw[s]);
            B[c][s] = widen_query16<WIDE>((dw[s] >> (16 * hh)) & 0xFFFFu);
        }
        pq[c] = pc;
    }

    // staging of one train tile of TT trains: thread -> (train r + 32 k, dword s), k < TT/32; rows past the end repeat
    // the last train (their accumulator preload kNone keeps them out of every top-2). The 8 lanes of a train add up
    // its popcount (DPP) and all store the same preload word. No branches: the tile after the last is staged too,
    // clamped.
    constexpr int NR = TT / 32;
    const int sr = tid >> 3, ss = tid & 7;
    uint32_t dreg[NR];
    auto load_tile = [&](int it) {
#pragma unroll
        for (int k = 0; k < NR; k++) dreg[k] = tw[(int64_t)min(it * TT + sr + 32 * k, nt - 1) * 8 + ss];
    };
    auto store_tile = [&](int it, int buf) {
#pragma unroll
        for (int k = 0; k < NR; k++) {
            v4i* p = reinterpret_cast<v4i*>(&s_a[buf][(sr + 32 * k) * kRowB + ss * 32]);
            p[0] = widen_train16<WIDE>(dreg[k] & 0xFFFFu);
            p[1] = widen_train16<WIDE>(dreg[k] >> 16);
            const int pt = sum8(__popc(dreg[k])), tt = it * TT + sr + 32 * k;
            s_base[buf][sr + 32 * k] = tt < nt ? (WIDE ? (((pt + 256) << 16) | tt) : ((pt + 256) << 12) + tt) + kKeyBias : kNone;
        }
    };

    // running (best, runner-up) per column tile, two independent chains (even / odd accumulator registers)
    float m1[NC][2], m2[NC][2];
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int p = 0; p < 2; p++) { m1[c][p] = __int_as_float(0x7F7FFFFF); m2[c][p] = __int_as_float(0x7F7FFFFF); }

    const int it0 = tsplit ? (int)blockIdx.y * tsplit : 0;
    const int ntiles = tsplit ? min((nt + TT - 1) / TT, it0 + tsplit) : (nt + TT - 1) / TT;
    if (ntiles > it0) { load_tile(it0); store_tile(it0, 0); }
    __syncthreads();
    for (int it = it0; it < ntiles; it++) {
        const int buf = (it - it0) & 1;
        load_tile(it + 1);
#pragma unroll
        for (int rt = 0; rt < NR; rt++) {
            const uint8_t* arow = &s_a[buf][(rt * 32 + col) * kRowB + hh * 16];
            v4i A[8];
#pragma unroll
            for (int s = 0; s < 8; s++) A[s] = *reinterpret_cast<const v4i*>(arow + s * 32);
            // preload of accumulator register j of this lane = train row (j & 3) + 8 (j >> 2) + 4 hh of the row tile
            v16i bs;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const v4i b4 = *reinterpret_cast<const v4i*>(&s_base[buf][rt * 32 + 8 * g + 4 * hh]);
                bs[4 * g] = b4.x; bs[4 * g + 1] = b4.y; bs[4 * g + 2] = b4.z; bs[4 * g + 3] = b4.w;
            }
#pragma unroll
            for (int cp = 0; cp < NC; cp += 2) {
                v16i acc0 = {}, acc1 = {};
                if (!WIDE) { acc0 = bs; acc1 = bs; }
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[s], B[cp][s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[s], B[cp + 1][s], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    top2_update(m1[cp][j & 1], m2[cp][j & 1], WIDE ? (acc0[j] << 10) + bs[j] : acc0[j]);
                    top2_update(m1[cp + 1][j & 1], m2[cp + 1][j & 1], WIDE ? (acc1[j] << 10) + bs[j] : acc1[j]);
                }
            }
        }
        store_tile(it + 1, buf ^ 1);
        __syncthreads();
    }

    // merge the two chains of a column tile, then lanes l and l ^ 32 (same query, disjoint trains); plain integer
    // compares from here on
    uint32_t k0[NC], k1[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int x1 = __float_as_int(m1[c][0]), x2 = __float_as_int(m2[c][0]);
        const int y1 = __float_as_int(m1[c][1]), y2 = __float_as_int(m2[c][1]);
        const int a1 = min(x1, y1);
        const int a2 = min(max(x1, y1), min(x2, y2));
        const int p1 = __shfl_xor(a1, 32), p2 = __shfl_xor(a2, 32);
        k0[c] = final_key<WIDE>(min(a1, p1), pq[c]);
        k1[c] = final_key<WIDE>(min(max(a1, p1), min(a2, p2)), pq[c]);
    }
    if (MODE == 0) {
        if (hh == 0) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const int qi = q0 + 32 * c + col;
                if (qi < nq) keys[(int64_t)(tsplit ? blockIdx.y : pair) * maxq + qi] = make_uint2(k0[c], k1[c]);
            }
        }
    } else {
        // LoopClosure.cpp:92  m[0].distance < 0.7 * m[1].distance  (float distances, double arithmetic)
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        int n_ok = 0;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int qi = q0 + 32 * c + col;
            bool ok = false;
            if (hh == 0 && qi < nq && k1[c] != 0xFFFFFFFFu)
                ok = (double)(float)(k0[c] >> 16) < ratio * (double)(float)(k1[c] >> 16);
            n_ok += __popcll(__ballot(ok));
        }
        if (lane == 0 && n_ok) atomicAdd(&s_cnt, n_ok);
        __syncthreads();
        if (tid == 0 && s_cnt) atomicAdd(&good[pair], s_cnt);
    }
}

// ---- the same kernel on the FP4 matrix path (train sets up to 4096 descriptors) ----------------------------------------
// Every descriptor bit is one E2M1 value: train bit -> -2.0 (1100b), query bit -> +1.0 (0010b); one
// v_mfma_f32_32x32x64_f8f6f4 (unscaled form: the scale arguments are the constant 0) covers 64 bits of K in the time the int8
// form covers 32, with fp32 accumulators in which everything here is an exact integer multiple of 2^-12:
//     key = (popcount(t) + 257 - 2 |q & t|) + index / 4096          (< 1024, 22 significant bits)
// preloaded as (popcount(t) + 257) + index / 4096, so the MFMA again delivers the sortable key and the vector ALU keeps the
// running top-2 with v_med3_f32 on the VALUES (all positive). Fragments are half the size of the int8 ones (16 bytes per lane
// and k-step of 64 bits): half the LDS traffic and registers per distance. Lane map as above: row / column = lane & 31,
// K half = lane >> 5 (32 bits -> 32 nibbles), identical for A and B.
constexpr int kRowF = 144;           // LDS bytes per widened train: 128 + 16 (ds_read_b128 of the 16-lane groups conflict-free)
constexpr float kNoneF = 1.0e30f;
template <int MODE, int NC, int TT>
__device__ __forceinline__ void knn2_body_fp4(
    const uint8_t* __restrict__ q, const int* __restrict__ nq_arr, int nq_fixed, const uint8_t* __restrict__ t,
    const int* __restrict__ nt_arr, int nt_fixed, int64_t q_stride, int64_t t_stride, uint2* __restrict__ keys, int maxq,
    double ratio, int* __restrict__ good, int tsplit) {
    __shared__ __attribute__((aligned(16))) uint8_t s_a[2][TT * kRowF];
    __shared__ __attribute__((aligned(16))) float s_base[2][TT];
    __shared__ uint32_t s_lut[256];          // byte -> eight E2M1 nibbles of -2.0 (the train side): widening a staged dword is
                                             // four LDS reads instead of 32 vector-ALU instructions, which the top-2 updates
                                             // leave no room for in the MFMAs' shadow (0.510 -> 0.500 us per pair)
    __shared__ int s_cnt;
    constexpr int QB = 128 * NC;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = lane & 31, hh = lane >> 5;
    const int pair = tsplit ? 0 : blockIdx.y;
    const int nq = nq_arr ? nq_arr[pair] : nq_fixed;
    const int nt = nt_arr ? nt_arr[pair] : nt_fixed;
    if ((int)blockIdx.x * QB >= nq) return;
    s_lut[tid] = (uint32_t)spread8((uint32_t)tid, 0xCu);
    __syncthreads();
    // (the query side reads the same table: +1.0 = 0010b is -2.0 = 1100b shifted right by two and masked; 0.4985 -> 0.494 us)
    auto widen_q = [&](uint32_t w) -> v4i {
        v4i r;
        r.x = (int)((s_lut[w & 0xFFu] >> 2) & 0x22222222u); r.y = (int)((s_lut[(w >> 8) & 0xFFu] >> 2) & 0x22222222u);
        r.z = (int)((s_lut[(w >> 16) & 0xFFu] >> 2) & 0x22222222u); r.w = (int)((s_lut[w >> 24] >> 2) & 0x22222222u);
        return r;
    };
    const uint4* qp = reinterpret_cast<const uint4*>(q + (int64_t)pair * q_stride);
    const uint32_t* tw = reinterpret_cast<const uint32_t*>(t + (int64_t)pair * t_stride);

    // resident B fragments: column tile c, k-step s = bits [64 s + 32 hh, +32) of query q0 + 32 c + col
    const int q0 = blockIdx.x * QB + wv * (32 * NC);
    v4i B[NC][4];
    int pq[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int qi = min(q0 + 32 * c + col, nq - 1);
        const uint4 lo = qp[2 * qi], hi = qp[2 * qi + 1];
        const uint32_t dw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        int pc = 0;
#pragma unroll
        for (int s = 0; s < 8; s++) pc += __popc(dw[s]);
#pragma unroll
        for (int s = 0; s < 4; s++) B[c][s] = widen_q(hh ? dw[2 * s + 1] : dw[2 * s]);
        pq[c] = pc;
    }

    // staging: thread -> (train sr + 32 k, dword ss): 32 bits -> 16 bytes at ss * 16 (= k-step ss >> 1, half ss & 1)
    constexpr int NR = TT / 32;
    const int sr = tid >> 3, ss = tid & 7;
    uint32_t dreg[NR];
    auto load_tile = [&](int it) {
#pragma unroll
        for (int k = 0; k < NR; k++) dreg[k] = tw[(int64_t)min(it * TT + sr + 32 * k, nt - 1) * 8 + ss];
    };
    auto store_tile = [&](int it, int buf) {
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const uint32_t w = dreg[k];
            v4i wv4;
            wv4.x = (int)s_lut[w & 0xFFu]; wv4.y = (int)s_lut[(w >> 8) & 0xFFu]; wv4.z = (int)s_lut[(w >> 16) & 0xFFu]; wv4.w = (int)s_lut[w >> 24];
            *reinterpret_cast<v4i*>(&s_a[buf][(sr + 32 * k) * kRowF + ss * 16]) = wv4;
            const int pt = sum8(__popc(dreg[k])), tt = it * TT + sr + 32 * k;
            s_base[buf][sr + 32 * k] = tt < nt ? (float)(pt + 257) + (float)tt * (1.0f / 4096.0f) : kNoneF;
        }
    };

    float m1[NC][2], m2[NC][2];
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int p = 0; p < 2; p++) { m1[c][p] = 3.0e38f; m2[c][p] = 3.0e38f; }

    const int it0 = tsplit ? (int)blockIdx.y * tsplit : 0;
    const int ntiles = tsplit ? min((nt + TT - 1) / TT, it0 + tsplit) : (nt + TT - 1) / TT;
    if (ntiles > it0) { load_tile(it0); store_tile(it0, 0); }
    __syncthreads();
    for (int it = it0; it < ntiles; it++) {
        const int buf = (it - it0) & 1;
        load_tile(it + 1);
#pragma unroll
        for (int rt = 0; rt < NR; rt++) {
            const uint8_t* arow = &s_a[buf][(rt * 32 + col) * kRowF + hh * 16];
            v4i A[4];
#pragma unroll
            for (int s = 0; s < 4; s++) A[s] = *reinterpret_cast<const v4i*>(arow + s * 32);
            v16f bs;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float4 b4 = *reinterpret_cast<const float4*>(&s_base[buf][rt * 32 + 8 * g + 4 * hh]);
                bs[4 * g] = b4.x; bs[4 * g + 1] = b4.y; bs[4 * g + 2] = b4.z; bs[4 * g + 3] = b4.w;
            }
#pragma unroll
            for (int cp = 0; cp < NC; cp += 2) {
                v16f acc0 = bs, acc1 = bs;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    // (the builtin takes 8 dwords per operand whatever the format; E2M1 reads the first four: leave the rest undefined)
                    const v8i a8 = __builtin_shufflevector(A[s], A[s], 0, 1, 2, 3, -1, -1, -1, -1);
                    const v8i b0 = __builtin_shufflevector(B[cp][s], B[cp][s], 0, 1, 2, 3, -1, -1, -1, -1);
                    const v8i b1 = __builtin_shufflevector(B[cp + 1][s], B[cp + 1][s], 0, 1, 2, 3, -1, -1, -1, -1);
                    acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b0, acc0, 4, 4, 0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b1, acc1, 4, 4, 0, 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    { const float k = acc0[j]; const float n2 = __builtin_amdgcn_fmed3f(m1[cp][j & 1], m2[cp][j & 1], k);
                      m1[cp][j & 1] = __builtin_amdgcn_fmed3f(m1[cp][j & 1], k, 0.0f); m2[cp][j & 1] = n2; }
                    { const float k = acc1[j]; const float n2 = __builtin_amdgcn_fmed3f(m1[cp + 1][j & 1], m2[cp + 1][j & 1], k);
                      m1[cp + 1][j & 1] = __builtin_amdgcn_fmed3f(m1[cp + 1][j & 1], k, 0.0f); m2[cp + 1][j & 1] = n2; }
                }
            }
        }
        store_tile(it + 1, buf ^ 1);
        __syncthreads();
    }

    // merge the two chains of a column tile, then lanes l and l ^ 32 (same query, disjoint trains): the keys are positive
    // floats, whose order is the order of their bit patterns
    uint32_t k0[NC], k1[NC];
    auto to_key = [&](int bits, int pqc) -> uint32_t {
        const float v = __int_as_float(bits);
        if (v > 1.0e29f) return 0xFFFFFFFFu;
        const int d = (int)v;                                         // popcount(t) + 257 - 2 |q & t|
        const int idx = (int)((v - (float)d) * 4096.0f);              // exact: multiples of 2^-12
        return ((uint32_t)(d - 257 + pqc) << 16) | (uint32_t)idx;
    };
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int x1 = __float_as_int(m1[c][0]), x2 = __float_as_int(m2[c][0]);
        const int y1 = __float_as_int(m1[c][1]), y2 = __float_as_int(m2[c][1]);
        const int a1 = min(x1, y1);
        const int a2 = min(max(x1, y1), min(x2, y2));
        const int p1 = __shfl_xor(a1, 32), p2 = __shfl_xor(a2, 32);
        k0[c] = to_key(min(a1, p1), pq[c]);
        k1[c] = to_key(min(max(a1, p1), min(a2, p2)), pq[c]);
    }
    if (MODE == 0) {
        if (hh == 0) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const int qi = q0 + 32 * c + col;
                if (qi < nq) keys[(int64_t)(tsplit ? blockIdx.y : pair) * maxq + qi] = make_uint2(k0[c], k1[c]);
            }
        }
    } else {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        int n_ok = 0;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int qi = q0 + 32 * c + col;
            bool ok = false;
            if (hh == 0 && qi < nq && k1[c] != 0xFFFFFFFFu)
                ok = (double)(float)(k0[c] >> 16) < ratio * (double)(float)(k1[c] >> 16);
            n_ok += __popcll(__ballot(ok));
        }
        if (lane == 0 && n_ok) atomicAdd(&s_cnt, n_ok);
        __syncthreads();
        if (tid == 0 && s_cnt) atomicAdd(&good[pair], s_cnt);
    }
}

// Shapes measured in bench.py's two-stream schedule (frames/s; kNN-2 us per pair): 512 queries per workgroup at 2 waves per
// SIMD 318.3 k / 0.550, at 3 waves per SIMD (<= 168 VGPRs; the spills are outside the loop) 326.7 k / 0.508, 1024 queries at 2
// waves 323.3 k / 0.513, 256 queries at 4 waves 307.9 k / 0.567 (and the FAST/blur kernel beside it at 0.16 instead of 0.19);
// 128-train tiles instead of 64: +0.4 %. The int8 kernel above: 297.3 k / 0.778. A software-pipelined inner loop (the MFMAs of
// column tile c between the top-2 updates of tile c - 1) needs 220 VGPRs = 2 waves per SIMD and measures 0.50 us like this form
// at 3: the two pipes overlap across waves, not inside one.
constexpr int kFp4Nc = 4, kFp4Tt = 64;
template <int MODE, int NC, int TT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NC == kFp4Nc ? 3 : 2))) void k_knn2_fp4(
    const uint8_t* __restrict__ q, const int* __restrict__ nq_arr, int nq_fixed, const uint8_t* __restrict__ t,
    const int* __restrict__ nt_arr, int nt_fixed, int64_t q_stride, int64_t t_stride, uint2* __restrict__ keys, int maxq,
    double ratio, int* __restrict__ good, int tsplit, const int* __restrict__ gate, int gate_want) {
    if (gate && *gate != gate_want) return;
    knn2_body_fp4<MODE, NC, TT>(q, nq_arr, nq_fixed, t, nt_arr, nt_fixed, q_stride, t_stride, keys, maxq, ratio, good, tsplit);
}

// Two waves per SIMD in both shapes (<= 256 VGPRs; the 512-query shape holds 128 VGPRs of query fragments per wave).
template <int MODE, bool WIDE, int NC, int TT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_knn2_mfma(
    const uint8_t* __restrict__ q, const int* __restrict__ nq_arr, int nq_fixed, const uint8_t* __restrict__ t,
    const int* __restrict__ nt_arr, int nt_fixed, int64_t q_stride, int64_t t_stride, uint2* __restrict__ keys, int maxq,
    double ratio, int* __restrict__ good, int tsplit, const int* __restrict__ gate, int gate_want) {
    if (gate && *gate != gate_want) return;      // the other key layout's launch does this batch (see launch_knn2_mfma)
    knn2_body<MODE, WIDE, NC, TT>(q, nq_arr, nq_fixed, t, nt_arr, nt_fixed, q_stride, t_stride, keys, maxq, ratio, good, tsplit);
}

// gate[0] = 1 when any pair's train set exceeds the narrow key layout (12-bit train index), else 0
__global__ __launch_bounds__(256) void k_nt_gate(const int* __restrict__ nt_arr, int n_pairs, int* __restrict__ gate) {
    __shared__ int s_any;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    int any = 0;
    for (int i = threadIdx.x; i < n_pairs; i += 256) any |= nt_arr[i] > kNarrowMax ? 1 : 0;
    if (any) s_any = 1;
    __syncthreads();
    if (threadIdx.x == 0) gate[0] = s_any;
}

template <int MODE, bool WIDE, int NC, int TT>
void launch_one(int nq_max, int n_pairs, hipStream_t st, const uint8_t* q, const int* nq_arr, int nq_fixed, const uint8_t* t,
                const int* nt_arr, int nt_fixed, int64_t q_stride, int64_t t_stride, uint2* keys, int maxq, double ratio,
                int* good, int tsplit = 0, int nsplit = 1, const int* gate = nullptr, int gate_want = 0) {
    const dim3 grid((unsigned)((nq_max + 128 * NC - 1) / (128 * NC)), (unsigned)(tsplit ? nsplit : n_pairs));
    hipLaunchKernelGGL((k_knn2_mfma<MODE, WIDE, NC, TT>), grid, dim3(256), 0, st, q, nq_arr, nq_fixed, t, nt_arr, nt_fixed,
                       q_stride, t_stride, keys, maxq, ratio, good, tsplit, gate, gate_want);
}

template <int MODE, int NC, int TT>
void launch_one_fp4(int nq_max, int n_pairs, hipStream_t st, const uint8_t* q, const int* nq_arr, int nq_fixed, const uint8_t* t,
                    const int* nt_arr, int nt_fixed, int64_t q_stride, int64_t t_stride, uint2* keys, int maxq, double ratio,
                    int* good, int tsplit = 0, int nsplit = 1, const int* gate = nullptr, int gate_want = 0) {
    const dim3 grid((unsigned)((nq_max + 128 * NC - 1) / (128 * NC)), (unsigned)(tsplit ? nsplit : n_pairs));
    hipLaunchKernelGGL((k_knn2_fp4<MODE, NC, TT>), grid, dim3(256), 0, st, q, nq_arr, nq_fixed, t, nt_arr, nt_fixed,
                       q_stride, t_stride, keys, maxq, ratio, good, tsplit, gate, gate_want);
}

}  // namespace

const char* launch_knn2_mfma(int mode, int nq_max, int n_pairs, hipStream_t st, const uint8_t* q, const int* nq_arr, int nq_fixed,
                             const uint8_t* t, const int* nt_arr, int nt_fixed, int64_t q_stride, int64_t t_stride,
                             uint2* keys, int maxq, double ratio, int* good, int max_train, int* gate) {
    if (nq_max <= 0 || n_pairs <= 0) return "";
    static const int force_nc = [] { const char* e = aria_getenv("ARIA_KNN_NC"); return e ? atoi(e) : 0; }();
    // variants build: ARIA_KNN_IMPL=int8 keeps the int8 kernel for every train-set size (the product takes the FP4 path up to 4096)
    static const bool force_i8 = [] { const char* e = aria_getenv("ARIA_KNN_IMPL"); return e && e[0] == 'i'; }();
    const bool wide = max_train > kNarrowMax;      // only built in the 256-query form
    // 512-query workgroups amortise the train staging and the A-fragment reads over twice the MFMAs, but need enough
    // workgroups to fill 256 CUs x 2; small jobs (a single frame pair) take the 256-query form
    const int64_t blocks4 = (int64_t)((nq_max + 511) / 512) * n_pairs;
    const bool nc4 = !wide && (force_nc ? force_nc == 4 : blocks4 >= 1024);
#define ARIA_KNN_ARGS nq_max, n_pairs, st, q, nq_arr, nq_fixed, t, nt_arr, nt_fixed, q_stride, t_stride, keys, maxq, ratio, good
    // The buffers allow more than 4096 trains per pair (e.g. 4000 features + tie slack) but the actual counts live on the
    // device and are usually within the narrow layout, whose 512-query kernel is ~25 % faster than the wide one: a tiny
    // kernel looks at the counts, and BOTH layouts are launched, each leaving at once unless the gate names it.
    if (wide && nt_arr && gate && max_train <= 65535 && (int64_t)((nq_max + 511) / 512) * n_pairs >= 1024 && !force_nc) {
        hipLaunchKernelGGL(k_nt_gate, dim3(1), dim3(256), 0, st, nt_arr, n_pairs, gate);
        if (mode == 0) {
            launch_one_fp4<0, kFp4Nc, kFp4Tt>(ARIA_KNN_ARGS, 0, 1, gate, 0);
            launch_one<0, true, 2, 64>(ARIA_KNN_ARGS, 0, 1, gate, 1);
        } else {
            launch_one_fp4<1, kFp4Nc, kFp4Tt>(ARIA_KNN_ARGS, 0, 1, gate, 0);
            launch_one<1, true, 2, 64>(ARIA_KNN_ARGS, 0, 1, gate, 1);
        }
        return "k_knn2_fp4|k_knn2_mfma";
    }
    if (mode == 0) {
        if (wide) launch_one<0, true, 2, 64>(ARIA_KNN_ARGS);
        else if (force_i8) { if (nc4) launch_one<0, false, 4, 64>(ARIA_KNN_ARGS); else launch_one<0, false, 2, 64>(ARIA_KNN_ARGS); }
        else if (nc4) launch_one_fp4<0, kFp4Nc, kFp4Tt>(ARIA_KNN_ARGS);
        else launch_one_fp4<0, 2, 64>(ARIA_KNN_ARGS);
    } else {
        if (wide) launch_one<1, true, 2, 64>(ARIA_KNN_ARGS);
        else if (force_i8) { if (nc4) launch_one<1, false, 4, 64>(ARIA_KNN_ARGS); else launch_one<1, false, 2, 64>(ARIA_KNN_ARGS); }
        else if (nc4) launch_one_fp4<1, kFp4Nc, kFp4Tt>(ARIA_KNN_ARGS);
        else launch_one_fp4<1, 2, 64>(ARIA_KNN_ARGS);
    }
#undef ARIA_KNN_ARGS
    return (wide || force_i8) ? "k_knn2_mfma" : "k_knn2_fp4";
}

int knn2_split_count(int nq, int nt) {
    const int qblocks = (nq + 255) / 256, ntiles = (nt + 63) / 64;
    int want = std::max(1, std::min(kKnnSplitMax, 192 / std::max(qblocks, 1)));
    const int per = std::max(1, (ntiles + want - 1) / want);
    return std::max(1, (ntiles + per - 1) / per);
}

void launch_knn2_mfma_split(hipStream_t st, const uint8_t* q, int nq, const uint8_t* t, int nt, uint2* keys, int maxq,
                            int nsplit, const int* nq_arr, const int* nt_arr) {
    if (nq <= 0 || nsplit <= 0) return;
    const int ntiles = (nt + 63) / 64, per = std::max(1, (ntiles + nsplit - 1) / nsplit);
    if (nt > kNarrowMax)
        launch_one<0, true, 2, 64>(nq, 1, st, q, nq_arr, nq, t, nt_arr, nt, 0, 0, keys, maxq, 0.0, nullptr, per, nsplit);
    else if (aria_getenv("ARIA_KNN_IMPL") && aria_getenv("ARIA_KNN_IMPL")[0] == 'i')      // variants build only
        launch_one<0, false, 2, 64>(nq, 1, st, q, nq_arr, nq, t, nt_arr, nt, 0, 0, keys, maxq, 0.0, nullptr, per, nsplit);
    else
        launch_one_fp4<0, 2, 64>(nq, 1, st, q, nq_arr, nq, t, nt_arr, nt, 0, 0, keys, maxq, 0.0, nullptr, per, nsplit);
}

}  // namespace aria
