// Brute-force Hamming kNN-2 on the gfx950 matrix cores.
// Same semantics as k_knn2 in match_hip.hip (reference: src/adapters/gpu/CudaMatcher.cpp:28-68 and
// src/legacy/LoopClosure.cpp:72-114; CPU cv::BFMatcher order: the two smallest (distance, train index) pairs).
//
// The 2000 x 2000 x 256-bit distance table of one frame pair is the one GEMM-shaped piece of this path -- it is
// compute-bound (160 KB in, 4M distances), not HBM-bound -- and for bit vectors
//     hamming(q, t) = popcount(q) + popcount(t) - 2 * <q, t>
// where <q, t> is the number of common set bits. With every bit widened to a 0/1 byte that inner product is an
// exact int8 x int8 -> int32 matrix product, v_mfma_i32_32x32x32_i8: one instruction = 32 trains x 32 queries x
// 32 bits (32 cycles), against 8 x (v_xor + v_bcnt) per single pair on the vector ALU. All arithmetic is integer,
// so the result is bit-identical to the VALU kernel.
//
// Workgroup = 4 waves = 256 queries; a wave keeps the widened fragments of its 64 queries (2 column tiles, 64 VGPRs)
// resident and walks the train set in tiles of 64 that the workgroup widens once into LDS (double-buffered).
// A = trains (rows), B = queries (columns): the accumulator then has the query on the lane and 16 trains in the
// registers, so the running top-2 is a per-lane min/med3 chain with no cross-lane traffic until the very end.
// The k order inside a fragment does not matter (a dot product), only that A and B split K the same way, which the
// symmetric A/B lane maps guarantee: lane l carries row/column l & 31 and K-half l >> 5.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"
#include "match_kernels.h"

namespace aria {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kQB = 256;             // queries per workgroup
constexpr int kTT = 64;              // trains per staged tile (two row tiles of 32)
constexpr int kRowB = 272;           // LDS bytes per widened train: 256 + 16 (bank spread for the ds_read_b128 fragments)
constexpr int kNone = 0x7FFFFFFF;    // key of "no train"

// 4 descriptor bits -> 4 bytes of 0/1 (bit k lands at 8k: n + n<<7 + n<<14 + n<<21, the partial products never overlap)
__device__ __forceinline__ int widen4(uint32_t n) { return (int)((n * 0x00204081u) & 0x01010101u); }
__device__ __forceinline__ v4i widen16(uint32_t hw) {
    v4i r;
    r.x = widen4(hw & 15u);
    r.y = widen4((hw >> 4) & 15u);
    r.z = widen4((hw >> 8) & 15u);
    r.w = widen4((hw >> 12) & 15u);
    return r;
}
__device__ __forceinline__ int mad_i24(int a, int b, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int med3_i32(int a, int b, int c) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

struct StageRegs { uint32_t d0, d1; uint4 a, b; };

template <int MODE>
__global__ __launch_bounds__(256) void k_knn2_mfma(const uint8_t* __restrict__ q, const int* __restrict__ nq_arr, int nq_fixed,
                                                   const uint8_t* __restrict__ t, const int* __restrict__ nt_arr, int nt_fixed,
                                                   int64_t q_stride, int64_t t_stride, uint2* __restrict__ keys, int maxq,
                                                   double ratio, int* __restrict__ good) {
    __shared__ __attribute__((aligned(16))) uint8_t s_a[2][kTT * kRowB];
    __shared__ __attribute__((aligned(16))) int s_base[2][kTT];
    __shared__ int s_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = lane & 31, hh = lane >> 5;
    const int pair = blockIdx.y;
    const int nq = nq_arr ? nq_arr[pair] : nq_fixed;
    const int nt = nt_arr ? nt_arr[pair] : nt_fixed;
    if ((int)blockIdx.x * kQB >= nq) return;
    const uint4* qp = reinterpret_cast<const uint4*>(q + (int64_t)pair * q_stride);
    const uint32_t* tw = reinterpret_cast<const uint32_t*>(t + (int64_t)pair * t_stride);
    const uint4* tp = reinterpret_cast<const uint4*>(t + (int64_t)pair * t_stride);

    // resident B fragments: column tile c, k-step s = bits [32 s + 16 hh, +16) of query q0 + 32 c + col
    const int q0 = blockIdx.x * kQB + wv * 64;
    v4i B[2][8];
    int pq[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int qi = q0 + 32 * c + col;
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if (qi < nq) { lo = qp[2 * qi]; hi = qp[2 * qi + 1]; }
        const uint32_t dw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        int pc = 0;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            pc += __popc(dw[s]);
            B[c][s] = widen16((dw[s] >> (16 * hh)) & 0xFFFFu);
        }
        pq[c] = pc;
    }

    // staging of one train tile: thread -> (train r, dword s) x 2, plus threads 0..63 -> base key of train tid
    auto load_tile = [&](int it, StageRegs& R) {
        const int t0 = it * kTT;
        const int r0 = tid >> 3, r1 = r0 + 32;
        R.d0 = (t0 + r0 < nt) ? tw[(int64_t)(t0 + r0) * 8 + (tid & 7)] : 0u;
        R.d1 = (t0 + r1 < nt) ? tw[(int64_t)(t0 + r1) * 8 + (tid & 7)] : 0u;
        if (tid < kTT) {
            R.a = make_uint4(0, 0, 0, 0); R.b = R.a;
            if (t0 + tid < nt) { R.a = tp[2 * (int64_t)(t0 + tid)]; R.b = tp[2 * (int64_t)(t0 + tid) + 1]; }
        }
    };
    auto store_tile = [&](int it, int buf, const StageRegs& R) {
        const int t0 = it * kTT;
        const int r0 = tid >> 3, r1 = r0 + 32, s = tid & 7;
        v4i* p0 = reinterpret_cast<v4i*>(&s_a[buf][r0 * kRowB + s * 32]);
        p0[0] = widen16(R.d0 & 0xFFFFu);
        p0[1] = widen16(R.d0 >> 16);
        v4i* p1 = reinterpret_cast<v4i*>(&s_a[buf][r1 * kRowB + s * 32]);
        p1[0] = widen16(R.d1 & 0xFFFFu);
        p1[1] = widen16(R.d1 >> 16);
        if (tid < kTT) {
            const int pc = __popc(R.a.x) + __popc(R.a.y) + __popc(R.a.z) + __popc(R.a.w) +
                           __popc(R.b.x) + __popc(R.b.y) + __popc(R.b.z) + __popc(R.b.w);
            s_base[buf][tid] = (t0 + tid < nt) ? ((pc << 16) | (t0 + tid)) : kNone;
        }
    };

    // running (best, runner-up) of key' = ((popcount(t) - 2 <q,t>) << 16) + train index, per column tile; popcount(q)
    // is the same for every train of a query and is added at the end
    int m1[2] = {kNone, kNone}, m2[2] = {kNone, kNone};
    const int kMinus2 = -131072;      // -2 << 16

    const int ntiles = (nt + kTT - 1) / kTT;
    StageRegs R;
    if (ntiles > 0) { load_tile(0, R); store_tile(0, 0, R); }
    __syncthreads();
    for (int it = 0; it < ntiles; it++) {
        const int buf = it & 1;
        const bool more = it + 1 < ntiles;
        if (more) load_tile(it + 1, R);
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
            const uint8_t* arow = &s_a[buf][(rt * 32 + col) * kRowB + hh * 16];
            v4i A[8];
#pragma unroll
            for (int s = 0; s < 8; s++) A[s] = *reinterpret_cast<const v4i*>(arow + s * 32);
            // accumulator register j of this lane is train row (j & 3) + 8 (j >> 2) + 4 hh of the tile
            v4i bs[4];
#pragma unroll
            for (int g = 0; g < 4; g++) bs[g] = *reinterpret_cast<const v4i*>(&s_base[buf][rt * 32 + 8 * g + 4 * hh]);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                v16i acc = {};
#pragma unroll
                for (int s = 0; s < 8; s++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[s], B[c][s], acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int key = mad_i24(acc[j], kMinus2, bs[j >> 2][j & 3]);
                    m2[c] = med3_i32(m1[c], m2[c], key);
                    m1[c] = min(m1[c], key);
                }
            }
        }
        if (more) store_tile(it + 1, buf ^ 1, R);
        __syncthreads();
    }

    // lanes l and l ^ 32 hold the same query over disjoint trains
    uint32_t k0[2], k1[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int p1 = __shfl_xor(m1[c], 32), p2 = __shfl_xor(m2[c], 32);
        const int n1 = min(m1[c], p1);
        const int n2 = min(max(m1[c], p1), min(m2[c], p2));
        k0[c] = n1 == kNone ? 0xFFFFFFFFu : (((uint32_t)((n1 >> 16) + pq[c]) << 16) | ((uint32_t)n1 & 0xFFFFu));
        k1[c] = n2 == kNone ? 0xFFFFFFFFu : (((uint32_t)((n2 >> 16) + pq[c]) << 16) | ((uint32_t)n2 & 0xFFFFu));
    }
    if (MODE == 0) {
        if (hh == 0) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int qi = q0 + 32 * c + col;
                if (qi < nq) keys[(int64_t)pair * maxq + qi] = make_uint2(k0[c], k1[c]);
            }
        }
    } else {
        // LoopClosure.cpp:92  m[0].distance < 0.7 * m[1].distance  (float distances, double arithmetic)
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        int n_ok = 0;
#pragma unroll
        for (int c = 0; c < 2; c++) {
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

}  // namespace

void launch_knn2_mfma(int mode, dim3 grid, hipStream_t st, const uint8_t* q, const int* nq_arr, int nq_fixed,
                      const uint8_t* t, const int* nt_arr, int nt_fixed, int64_t q_stride, int64_t t_stride,
                      uint2* keys, int maxq, double ratio, int* good) {
    if (mode == 0)
        hipLaunchKernelGGL(k_knn2_mfma<0>, grid, dim3(256), 0, st, q, nq_arr, nq_fixed, t, nt_arr, nt_fixed, q_stride,
                           t_stride, keys, maxq, ratio, good);
    else
        hipLaunchKernelGGL(k_knn2_mfma<1>, grid, dim3(256), 0, st, q, nq_arr, nq_fixed, t, nt_arr, nt_fixed, q_stride,
                           t_stride, keys, maxq, ratio, good);
}

}  // namespace aria
