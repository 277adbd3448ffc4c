// Brute-force Hamming kNN-2 matcher for 256-bit descriptors on gfx950 + its C-ABI.
// Replaces the reference's CudaMatcher (src/adapters/gpu/CudaMatcher.cpp:28-68, cv::cuda BFMatcher::knnMatch(k=2)
// + Lowe ratio) and the legacy loop-closure scan (src/legacy/LoopClosure.cpp:72-114, cv::BFMatcher CPU).
// Semantics follow CPU cv::BFMatcher (OpenCV 4.9.0 batch_distance.cpp): in-order scan with strict '<'
// insertion, i.e. the two smallest (distance, train index) pairs in lexicographic order.
//
// Not HBM-bound (160 KB of compulsory traffic per 2000x2000 pair): it is VALU-integer work, 8 x (v_xor +
// v_bcnt_u32_b32 accumulate) per descriptor pair. One lane owns one query (8 dwords in VGPRs); train descriptors
// are staged through LDS in tiles of 256 and read as wave-uniform (broadcast) ds_read_b128; the running top-2
// is kept as two packed keys (distance << 16 | train index) so ties resolve to the lower index for free.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <climits>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"
#include "match_kernels.h"

using namespace aria;

namespace {


// v_bcnt_u32_b32 dst, src, acc: popcount(src) + acc in one VALU instruction
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

#ifdef ARIA_VARIANTS      // the vector-ALU kNN-2 (round 1's kernel): variants build only, ARIA_KNN_IMPL=valu
constexpr int kMatchTile = 256;
#ifndef ARIA_KNN_SCALAR
#define ARIA_KNN_SCALAR 0   // measured (tools/microbench/valu_rates.hip): v_xor with an SGPR source costs 4.35 cycles per wave-instruction,
                            // VGPR-VGPR 2.55 -> the LDS-broadcast form is the faster of the two
#endif
constexpr bool kUseScalarTrain = ARIA_KNN_SCALAR != 0;
template <int MODE>   // 0: store the two keys per query; 1: count queries passing the double-precision ratio test
__global__ __launch_bounds__(256) void k_knn2(const uint8_t* __restrict__ q, const int* __restrict__ nq_arr, int nq_fixed,
                                              const uint8_t* __restrict__ t, const int* __restrict__ nt_arr, int nt_fixed,
                                              int64_t q_stride, int64_t t_stride, uint2* __restrict__ keys, int maxq,
                                              double ratio, int* __restrict__ good) {
    __shared__ uint4 s_t[kMatchTile * 2];
    __shared__ int s_cnt;
    const int tid = threadIdx.x;
    const int pair = blockIdx.y;
    const int nq = nq_arr ? nq_arr[pair] : nq_fixed;
    const int nt = nt_arr ? nt_arr[pair] : nt_fixed;
    if ((int)blockIdx.x * 256 >= nq) return;
    const int qi = blockIdx.x * 256 + tid;
    const uint4* qp = reinterpret_cast<const uint4*>(q + (int64_t)pair * q_stride);
    const uint4* tp = reinterpret_cast<const uint4*>(t + (int64_t)pair * t_stride);
    uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
    if (qi < nq) { qa = qp[2 * qi]; qb = qp[2 * qi + 1]; }
    uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
    if (kUseScalarTrain) {
        // The train descriptor is the same for every lane: read it through the scalar data cache (s_load_dwordx8 into
        // SGPRs) and feed it to the VALU as scalar operands -- no LDS staging, no barriers, no ds_read issue slots.
        // Scalar loads return out of order, so the only wait is lgkmcnt(0): fetch the NEXT four descriptors, do the
        // VALU work of the CURRENT four (~90 instructions) under that latency, then swap.
        constexpr int U = 4;
        uint4 ca[U], cb[U], na[U], nb[U];
        const int last = nt - 1;
#pragma unroll
        for (int u = 0; u < U; u++) { const int jj = min(u, last); ca[u] = tp[2 * jj]; cb[u] = tp[2 * jj + 1]; }
        for (int j = 0; j < nt; j += U) {
#pragma unroll
            for (int u = 0; u < U; u++) { const int jj = min(j + U + u, last); na[u] = tp[2 * jj]; nb[u] = tp[2 * jj + 1]; }
#pragma unroll
            for (int u = 0; u < U; u++) {
                // 8 x (v_xor, v_bcnt accumulate): the popcount adds into its own accumulator operand, no adder tree
                uint32_t d = 0;
                d = bcnt_acc(qa.x ^ ca[u].x, d);
                d = bcnt_acc(qa.y ^ ca[u].y, d);
                d = bcnt_acc(qa.z ^ ca[u].z, d);
                d = bcnt_acc(qa.w ^ ca[u].w, d);
                d = bcnt_acc(qb.x ^ cb[u].x, d);
                d = bcnt_acc(qb.y ^ cb[u].y, d);
                d = bcnt_acc(qb.z ^ cb[u].z, d);
                d = bcnt_acc(qb.w ^ cb[u].w, d);
                const uint32_t key = (j + u < nt) ? ((d << 16) | (uint32_t)(j + u)) : 0xFFFFFFFFu;
                // most candidates lose against every lane's current runner-up: one compare + a uniform branch
                if (__any(key < k1)) {
                    k1 = min(k1, max(k0, key));
                    k0 = min(k0, key);
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) { ca[u] = na[u]; cb[u] = nb[u]; }
        }
    } else
    for (int t0 = 0; t0 < nt; t0 += kMatchTile) {
        __syncthreads();
        if (t0 + tid < nt) {
            s_t[2 * tid] = tp[2 * (t0 + tid)];
            s_t[2 * tid + 1] = tp[2 * (t0 + tid) + 1];
        }
        __syncthreads();
        const int cnt = min(kMatchTile, nt - t0);
        for (int j = 0; j < cnt; j++) {
            const uint4 a = s_t[2 * j], b = s_t[2 * j + 1];
            uint32_t d = 0;
            d = bcnt_acc(qa.x ^ a.x, d);
            d = bcnt_acc(qa.y ^ a.y, d);
            d = bcnt_acc(qa.z ^ a.z, d);
            d = bcnt_acc(qa.w ^ a.w, d);
            d = bcnt_acc(qb.x ^ b.x, d);
            d = bcnt_acc(qb.y ^ b.y, d);
            d = bcnt_acc(qb.z ^ b.z, d);
            d = bcnt_acc(qb.w ^ b.w, d);
            const uint32_t key = (d << 16) | (uint32_t)(t0 + j);
            k1 = min(k1, max(k0, key));
            k0 = min(k0, key);
        }
    }
    if (MODE == 0) {
        if (qi < nq) keys[(int64_t)pair * maxq + qi] = make_uint2(k0, k1);
    } else {
        // LoopClosure.cpp:92  m[0].distance < 0.7 * m[1].distance  (float distances, double arithmetic)
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        bool ok = false;
        if (qi < nq && k1 != 0xFFFFFFFFu) ok = (double)(float)(k0 >> 16) < ratio * (double)(float)(k1 >> 16);
        const unsigned long long m = __ballot(ok);
        if ((tid & 63) == 0 && m) atomicAdd(&s_cnt, __popcll(m));
        __syncthreads();
        if (tid == 0 && s_cnt) atomicAdd(&good[pair], s_cnt);
    }
}

#endif  // ARIA_VARIANTS

// CudaMatcher.cpp:59-67: fp32 Lowe test, matches appended in query order. One workgroup per pair; the ordered
// compaction uses wave ballots + popcount prefixes.
// qflags / tflags (optional): per-keypoint "lies in a dynamic object" flags of the query / train frame of every pair
// (k_flag_keypoints); a ratio-test survivor with a flagged endpoint is dropped and counted (src/main.cpp:164-175).
template <int NT>      // threads: 256 per pair for batches, 1024 for the single pair of the latency schedule
__global__ __launch_bounds__(NT) void k_ratio_compact(const uint2* __restrict__ keys, const int* __restrict__ nq_arr,
                                                       int nq_fixed, int maxq, float ratio,
                                                       aria_match* __restrict__ out, int* __restrict__ nout, int cap,
                                                       int* __restrict__ err, const uint8_t* __restrict__ qflags = nullptr,
                                                       const uint8_t* __restrict__ tflags = nullptr, int64_t flag_stride = 0,
                                                       int* __restrict__ nfiltered = nullptr, int nsplit = 1) {
    constexpr int NW = NT / 64;
    __shared__ int s_w[NW];
    __shared__ int s_f;
    if (threadIdx.x == 0) s_f = 0;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pair = blockIdx.x;
    const int nq = nq_arr ? nq_arr[pair] : nq_fixed;
    const uint2* kk = keys + (int64_t)pair * maxq;
    aria_match* o = out + (int64_t)pair * cap;
    int base = 0;
    // QP queries per thread and trip: their key loads (all slices) are requested together -- on an idle chip every dependent
    // round trip is a microsecond and the single-pair form (NT = 1024) is on the frame-at-a-time path: 2000 queries are one
    // trip of one round trip instead of two
    constexpr int QP = NT == 1024 ? 2 : 1;
    for (int q0 = 0; q0 < nq; q0 += NT * QP) {
        uint2 k[QP];
        uint2 os[QP][kKnnSplitMax - 1];
#pragma unroll
        for (int p = 0; p < QP; p++) {
            const int qi = q0 + p * NT + tid;
            k[p] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            if (qi < nq) {
                k[p] = kk[qi];
                if (nsplit > 1) {
#pragma unroll
                    for (int j = 0; j < kKnnSplitMax - 1; j++) os[p][j] = kk[(int64_t)min(1 + j, nsplit - 1) * maxq + qi];   // clamped loads past the last slice are not merged
                }
            }
        }
#pragma unroll
        for (int p = 0; p < QP; p++) {
            const int qi = q0 + p * NT + tid;
            bool ok = false;
            if (qi < nq) {
                // train slices of the latency schedule (one pair): two smallest of all
                if (nsplit > 1) {
#pragma unroll
                    for (int j = 0; j < kKnnSplitMax - 1; j++) {
                        if (1 + j < nsplit) {
                            const uint32_t lo = min(k[p].x, os[p][j].x);
                            k[p].y = min(max(k[p].x, os[p][j].x), min(k[p].y, os[p][j].y));
                            k[p].x = lo;
                        }
                    }
                }
                if (ratio == 0.0f) ok = k[p].x != 0xFFFFFFFFu;   // IMatcher.hpp:18 "0.0 = disabled"
                else ok = k[p].y != 0xFFFFFFFFu && (float)(k[p].x >> 16) < ratio * (float)(k[p].y >> 16);
                if (ok && qflags && (qflags[(int64_t)pair * flag_stride + qi] | tflags[(int64_t)pair * flag_stride + (k[p].x & 0xFFFFu)])) {
                    ok = false;
                    atomicAdd(&s_f, 1);
                }
            }
            const unsigned long long m = __ballot(ok);
            if (lane == 0) s_w[wv] = __popcll(m);
            __syncthreads();
            int off = base, tot = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                if (w < wv) off += s_w[w];
                tot += s_w[w];
            }
            off += __popcll(m & ((1ull << lane) - 1ull));
            if (ok) {
                if (off < cap) {
                    aria_match mm;
                    mm.query_idx = qi;
                    mm.train_idx = (int)(k[p].x & 0xFFFFu);
                    mm.distance = (float)(k[p].x >> 16);
                    o[off] = mm;
                } else {
                    atomicOr(err, ERRBIT_MATCHCAP);
                }
            }
            base += tot;
            __syncthreads();
        }
    }
    if (tid == 0) {
        nout[pair] = min(base, cap);
        if (nfiltered) nfiltered[pair] = s_f;
    }
}

// src/main.cpp:42-50 isInDynamicObject for every keypoint of every frame: flags[f][i] = 1 when keypoint i lies in one of
// frame f's boxes (x1, y1, x2, y2). mode 0: cv::Rect::contains of the point rounded to integers (half to even), half-open;
// mode 1: core::Detection::contains (include/core/Types.hpp:109-111), closed float intervals.
// box_frame_offset: frame f's keypoints are tested against the boxes of frame f + offset (no boxes where that frame does not
// exist): offset 0 for the query side, +1 for the TRAIN side of pairs (f + 1, f) -- main.cpp tests both endpoints of a match
// against the CURRENT frame's detections.
__global__ __launch_bounds__(256) void k_flag_keypoints(const aria_keypoint* __restrict__ kps, const int* __restrict__ counts,
                                                        int kp_cap, const float4* __restrict__ boxes, const int* __restrict__ nboxes,
                                                        int box_cap, int mode, uint8_t* __restrict__ flags, int box_frame_offset,
                                                        int n_frames) {
    __shared__ float4 s_box[64];
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int fb = f + box_frame_offset;
    const bool has_boxes = fb >= 0 && fb < n_frames;
    const int nb = has_boxes ? min(nboxes[fb], box_cap) : 0, n = min(counts[f], kp_cap);
    bool hit = false;
    float x = 0.f, y = 0.f;
    if (i < n) { x = kps[(int64_t)f * kp_cap + i].x; y = kps[(int64_t)f * kp_cap + i].y; }
    for (int b0 = 0; b0 < nb; b0 += 64) {
        __syncthreads();
        if (threadIdx.x < 64 && b0 + (int)threadIdx.x < nb) s_box[threadIdx.x] = boxes[(int64_t)fb * box_cap + b0 + threadIdx.x];
        __syncthreads();
        for (int b = 0; b < min(64, nb - b0); b++) {
            const float4 r = s_box[b];
            if (mode == 0) {
                const int px = __float2int_rn(x), py = __float2int_rn(y);
                hit |= (int)r.x <= px && px < (int)r.z && (int)r.y <= py && py < (int)r.w;
            } else {
                hit |= x >= r.x && x <= r.z && y >= r.y && y <= r.w;
            }
        }
    }
    if (i < kp_cap) flags[(int64_t)f * kp_cap + i] = (i < n && hit) ? 1 : 0;
}

__global__ void k_unpack_knn(const uint2* __restrict__ keys, int nq, int* __restrict__ idx, int* __restrict__ dist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    const uint2 k = keys[i];
    idx[2 * i] = k.x == 0xFFFFFFFFu ? -1 : (int)(k.x & 0xFFFFu);
    dist[2 * i] = k.x == 0xFFFFFFFFu ? INT_MAX : (int)(k.x >> 16);
    idx[2 * i + 1] = k.y == 0xFFFFFFFFu ? -1 : (int)(k.y & 0xFFFFu);
    dist[2 * i + 1] = k.y == 0xFFFFFFFFu ? INT_MAX : (int)(k.y >> 16);
}

}  // namespace

struct aria_matcher_s {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int max_query = 0, max_train = 0;
    bool knn_valu = false;          // ARIA_KNN_IMPL=valu: the vector-ALU kernel below instead of the matrix-core one
    const char* last_knn = "";      // kernel form of the most recent batch / database kNN-2 launch (aria_matcher_knn_kernel)
    uint2* d_keys = nullptr;        // grow-only scratch: [n_pairs][maxq]
    size_t keys_cap = 0;
    int* d_err = nullptr;
    // host-buffer path
    uint8_t* d_q = nullptr;
    uint8_t* d_t = nullptr;
    aria_match* d_m = nullptr;
    int* d_n = nullptr;             // [0] nmatches
    int* d_idx = nullptr;           // 2*max_query idx + 2*max_query dist
    uint8_t* h_stage = nullptr;     // pinned: q | t
    aria_match* h_m = nullptr;      // pinned
    int* h_n = nullptr;             // pinned [0] n, [1] err
    int* h_idx = nullptr;           // pinned
    // aria_matcher_match keeps the last two query sets on the device (ping-pong) and their bytes in pinned staging: a
    // front end matches frame i against frame i-1, i.e. the train set of a call is the query set of the call before,
    // which then is not uploaded again. Results come back in one copy: [4-int header: count][matches].
    uint8_t* d_pp[2] = {nullptr, nullptr};
    uint8_t* h_pp[2] = {nullptr, nullptr};
    int pp_n[2] = {-1, -1};
    bool pp_host[2] = {false, false};   // h_pp[k] mirrors d_pp[k] (false: the set came from a device pointer)
    int pp_cur = 0;
    // aria_matcher_match_device_async .. aria_matcher_finish: one pending operation
    bool dev_pending = false;
    hipEvent_t ev_done = nullptr;       // recorded behind the result copy of a pipelined device match (aria_matcher_finish waits
                                        // for it; the copy that makes the new set resident follows it on the stream)
    uint8_t* d_res = nullptr;       // [16 B header][max_query matches]
    uint8_t* h_res = nullptr;       // pinned
    // the four operations of aria_matcher_match as a hipGraph per ping-pong slot, keyed on the sizes; captured when a key
    // shows up the second time in a row for its slot (a stream of frames with the same keypoint count), eager otherwise
    struct MatchGraph {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        // key of exec: sizes, ratio, mode (MatchOps::mode) and the device pointers baked into the nodes
        int nq = -1, nt = -1, hit = -1; float ratio = -1.f; const void* pa = nullptr; const void* pb = nullptr;
        int seen_nq = -1, seen_nt = -1, seen_hit = -1; float seen_ratio = -1.f; const void* seen_pa = nullptr; const void* seen_pb = nullptr;
        void drop() {
            if (exec) hipGraphExecDestroy(exec);
            if (graph) hipGraphDestroy(graph);
            exec = nullptr; graph = nullptr; nq = nt = hit = -1;
        }
    } mg[2];
    bool graph_failed = false;
    // one-query-against-many path (IMatcher::matchMultiple / loop candidates): grow-only staging
    uint8_t* d_multi = nullptr;     // [n_cand][multi_rows * 32]
    int* d_mcnt = nullptr;          // [n_cand] rows per candidate, then [n_cand] results (matches or good counts)
    aria_match* d_mm = nullptr;     // [n_cand][max_query]
    size_t multi_bytes = 0, mm_entries = 0;
    int mcnt_cap = 0;
    // optional stage timing (HIP events on the launch stream)
    bool prof_enabled = false;
    unsigned prof_mask = ~0u;       // bit s: bracket stage s (0 knn2, 1 ratio_compact)
    struct Ev { hipEvent_t e[4]; int pairs; };
    std::vector<Ev> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[2] = {0, 0};
    int64_t prof_launches[2] = {0, 0};
    int64_t prof_pairs = 0;
    hipEvent_t prof_get() {
        if (!prof_pool.empty()) { hipEvent_t e = prof_pool.back(); prof_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        hipEventCreate(&e);
        return e;
    }
    void prof_collect() {
        for (Ev& v : prof_pending) {
            for (int s = 0; s < 2; s++) {
                if (!v.e[2 * s]) continue;          // stage not bracketed
                float t = 0.f;
                if (hipEventElapsedTime(&t, v.e[2 * s], v.e[2 * s + 1]) == hipSuccess) prof_ms[s] += t;
                prof_launches[s] += 1;
            }
            prof_pairs += v.pairs;
            for (int s = 0; s < 4; s++) if (v.e[s]) prof_pool.push_back(v.e[s]);
        }
        prof_pending.clear();
    }
};

namespace {

// kNN-2 of n_pairs pairs of at most nq_max queries: matrix-core kernel (knn2_mfma.hip) unless the handle asks for the
// VALU one
void launch_knn2(aria_matcher_s* m, int mode, int nq_max, int n_pairs, const uint8_t* q, const int* nq_arr, int nq_fixed,
                 const uint8_t* t, const int* nt_arr, int nt_fixed, int64_t q_stride, int64_t t_stride, uint2* keys,
                 int maxq, double ratio, int* good, int max_train) {
#ifndef ARIA_VARIANTS
    m->last_knn = launch_knn2_mfma(mode, nq_max, n_pairs, m->stream, q, nq_arr, nq_fixed, t, nt_arr, nt_fixed, q_stride, t_stride,
                                   keys, maxq, ratio, good, max_train, m->d_err + 1);
#else
    if (!m->knn_valu) {
        m->last_knn = launch_knn2_mfma(mode, nq_max, n_pairs, m->stream, q, nq_arr, nq_fixed, t, nt_arr, nt_fixed, q_stride, t_stride,
                                       keys, maxq, ratio, good, max_train, m->d_err + 1);
        return;
    }
    m->last_knn = "k_knn2 (vector ALU)";
    const dim3 grid((unsigned)((nq_max + 255) / 256), (unsigned)n_pairs);
    if (mode == 0)
        hipLaunchKernelGGL(k_knn2<0>, grid, dim3(256), 0, m->stream, q, nq_arr, nq_fixed, t, nt_arr, nt_fixed, q_stride,
                           t_stride, keys, maxq, ratio, good);
    else
        hipLaunchKernelGGL(k_knn2<1>, grid, dim3(256), 0, m->stream, q, nq_arr, nq_fixed, t, nt_arr, nt_fixed, q_stride,
                           t_stride, keys, maxq, ratio, good);
#endif
}

int ensure_keys(aria_matcher_s* m, size_t entries) {
    if (entries <= m->keys_cap) return ARIA_OK;
    ARIA_HIP(hipStreamSynchronize(m->stream));
    // the captured single-pair graphs have the old d_keys pointer baked into their kernel nodes: drop them, and forget
    // the "seen" keys so the next two calls capture afresh against the new buffer
    for (auto& G : m->mg) {
        G.drop();
        G.seen_nq = G.seen_nt = G.seen_hit = -1;
        G.seen_ratio = -1.f;
    }
    if (m->d_keys) hipFree(m->d_keys);
    m->d_keys = nullptr;
    m->keys_cap = 0;
    ARIA_HIP(hipMalloc(&m->d_keys, entries * sizeof(uint2)));
    m->keys_cap = entries;
    return ARIA_OK;
}

void matcher_free(aria_matcher_s* m) {
    hipFree(m->d_keys); hipFree(m->d_err); hipFree(m->d_q); hipFree(m->d_t); hipFree(m->d_m); hipFree(m->d_n);
    hipFree(m->d_idx); hipFree(m->d_multi); hipFree(m->d_mcnt); hipFree(m->d_mm);
    m->mg[0].drop(); m->mg[1].drop();
    hipFree(m->d_pp[0]); hipFree(m->d_pp[1]); hipFree(m->d_res);
    if (m->h_pp[0]) hipHostFree(m->h_pp[0]);
    if (m->h_pp[1]) hipHostFree(m->h_pp[1]);
    if (m->h_res) hipHostFree(m->h_res);
    if (m->h_stage) hipHostFree(m->h_stage);
    if (m->h_m) hipHostFree(m->h_m);
    if (m->h_n) hipHostFree(m->h_n);
    if (m->h_idx) hipHostFree(m->h_idx);
}

int matcher_alloc(aria_matcher_s* m) {
    const size_t nq = (size_t)std::max(m->max_query, 1), nt = (size_t)std::max(m->max_train, 1);
    ARIA_HIP(hipMalloc(&m->d_err, 2 * sizeof(int)));      // [0] deferred error bits, [1] narrow/wide gate of the batch kNN-2
    ARIA_HIP(memset_on(m->stream, m->d_err, 0, 2 * sizeof(int)));
    ARIA_HIP(hipMalloc(&m->d_q, nq * 32));
    ARIA_HIP(hipMalloc(&m->d_t, nt * 32));
    ARIA_HIP(hipMalloc(&m->d_m, nq * sizeof(aria_match)));
    ARIA_HIP(hipMalloc(&m->d_n, sizeof(int)));
    ARIA_HIP(hipMalloc(&m->d_idx, nq * 4 * sizeof(int)));
    ARIA_HIP(hipHostMalloc(&m->h_stage, (nq + nt) * 32));
    ARIA_HIP(hipHostMalloc(&m->h_m, nq * sizeof(aria_match)));
    ARIA_HIP(hipHostMalloc(&m->h_n, 2 * sizeof(int)));
    ARIA_HIP(hipHostMalloc(&m->h_idx, nq * 4 * sizeof(int)));
    for (int k = 0; k < 2; k++) {
        ARIA_HIP(hipMalloc(&m->d_pp[k], nq * 32));
        ARIA_HIP(hipHostMalloc(&m->h_pp[k], nq * 32));
    }
    ARIA_HIP(hipMalloc(&m->d_res, 16 + nq * sizeof(aria_match)));
    ARIA_HIP(hipHostMalloc(&m->h_res, 16 + nq * sizeof(aria_match)));
    return ensure_keys(m, nq * (size_t)kKnnSplitMax);
}

int upload_pair(aria_matcher_s* m, const uint8_t* q, int nq, const uint8_t* t, int nt) {
    std::memcpy(m->h_stage, q, (size_t)nq * 32);
    std::memcpy(m->h_stage + (size_t)m->max_query * 32, t, (size_t)nt * 32);
    ARIA_HIP(hipMemcpyAsync(m->d_q, m->h_stage, (size_t)nq * 32, hipMemcpyHostToDevice, m->stream));
    ARIA_HIP(hipMemcpyAsync(m->d_t, m->h_stage + (size_t)m->max_query * 32, (size_t)nt * 32, hipMemcpyHostToDevice, m->stream));
    return ARIA_OK;
}

}  // namespace

extern "C" {

void aria_matcher_default_config(aria_matcher_config* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->struct_size = (int)sizeof(aria_matcher_config);
    c->max_query = 4096;
    c->max_train = 4096;
}

int aria_matcher_create(const aria_matcher_config* c, aria_matcher_t* out) {
    if (!c || !out || c->struct_size != (int)sizeof(aria_matcher_config)) return ARIA_E_INVALID;
    if (c->max_query < 1 || c->max_train < 1 || c->max_train > 65535 || c->max_query > (1 << 24)) return ARIA_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    ARIA_HIP(hipGetDeviceCount(&ndev));
    if (c->device < 0 || c->device >= ndev) {
        std::snprintf(last_hip_error_buf(), 256, "device %d not present (%d devices)", c->device, ndev);
        return ARIA_E_NO_DEVICE;
    }
    ARIA_HIP(hipSetDevice(c->device));
    aria_matcher_s* m = new (std::nothrow) aria_matcher_s();
    if (!m) return ARIA_E_OOM;
    m->device = c->device;
    m->max_query = c->max_query;
    m->max_train = c->max_train;
    { const char* e = aria_getenv("ARIA_KNN_IMPL"); m->knn_valu = e && std::strcmp(e, "valu") == 0; }
    if (c->stream) {
        m->stream = (hipStream_t)c->stream;
    } else {
        hipError_t e = create_stream(&m->stream);
        if (e != hipSuccess) { delete m; return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
        m->owns_stream = true;
    }
    int rc = matcher_alloc(m);
    if (rc == ARIA_OK && hipEventCreateWithFlags(&m->ev_done, hipEventDisableTiming) != hipSuccess) rc = ARIA_E_HIP;
    if (rc != ARIA_OK) { aria_matcher_destroy(m); return rc; }
    *out = m;
    return ARIA_OK;
}

void aria_matcher_destroy(aria_matcher_t m) {
    if (!m) return;
    hipSetDevice(m->device);
    if (m->stream) hipStreamSynchronize(m->stream);
    m->prof_collect();
    for (hipEvent_t e : m->prof_pool) hipEventDestroy(e);
    if (m->ev_done) hipEventDestroy(m->ev_done);
    matcher_free(m);
    if (m->owns_stream && m->stream) hipStreamDestroy(m->stream);
    delete m;
}

void* aria_matcher_stream(aria_matcher_t m) { return m ? (void*)m->stream : nullptr; }

// cudaStreamCreate / cudaStreamDestroy of the reference adapters (OrbCudaExtractor.cpp:28,50; CudaMatcher.cpp:16,24) for a
// host that does not link the HIP runtime itself: one stream to hand to several handles so that their work is ordered.
int aria_stream_create(int device, void** stream) {
    if (!stream) return ARIA_E_INVALID;
    *stream = nullptr;
    int ndev = 0;
    ARIA_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return ARIA_E_NO_DEVICE;
    ARIA_HIP(hipSetDevice(device));
    hipStream_t s = nullptr;
    ARIA_HIP(create_stream(&s));
    *stream = (void*)s;
    return ARIA_OK;
}
int aria_stream_destroy(int device, void* stream) {
    if (!stream) return ARIA_OK;
    ARIA_HIP(hipSetDevice(device));
    ARIA_HIP(hipStreamSynchronize((hipStream_t)stream));
    ARIA_HIP(hipStreamDestroy((hipStream_t)stream));
    return ARIA_OK;
}

const char* aria_matcher_knn_kernel(aria_matcher_t m) { return m ? m->last_knn : ""; }

int aria_matcher_set_profiling(aria_matcher_t m, int enable) {
    if (!m) return ARIA_E_INVALID;
    m->prof_enabled = enable != 0;
    m->prof_mask = (enable & 1) ? ~0u : ((unsigned)enable >> 1);
    return ARIA_OK;
}

int aria_matcher_get_profile(aria_matcher_t m, int reset, double* stage_ms, int64_t* stage_launches, int64_t* pairs) {
    if (!m) return ARIA_E_INVALID;
    ARIA_HIP(hipSetDevice(m->device));
    ARIA_HIP(hipStreamSynchronize(m->stream));
    m->prof_collect();
    for (int s = 0; s < 2; s++) {
        if (stage_ms) stage_ms[s] = m->prof_ms[s];
        if (stage_launches) stage_launches[s] = m->prof_launches[s];
    }
    if (pairs) *pairs = m->prof_pairs;
    if (reset) { m->prof_ms[0] = m->prof_ms[1] = 0; m->prof_launches[0] = m->prof_launches[1] = 0; m->prof_pairs = 0; }
    return ARIA_OK;
}

int aria_matcher_sync(aria_matcher_t m) {
    if (!m) return ARIA_E_INVALID;
    ARIA_HIP(hipSetDevice(m->device));
    ARIA_HIP(hipStreamSynchronize(m->stream));
    int bits = 0;
    ARIA_HIP(memcpy_on(m->stream, &bits, m->d_err, sizeof(int), hipMemcpyDeviceToHost));
    if (bits) ARIA_HIP(memset_on(m->stream, m->d_err, 0, sizeof(int)));
    return errbits_to_status(bits);
}

// One single-pair match on m->stream, as aria_matcher_match / aria_matcher_match_device(_async) set it up. The NEW
// descriptor set goes into ping-pong slot `cur` (from pinned host staging or from a device pointer) and stays resident for
// the next call; the OTHER set is the resident slot cur ^ 1, an explicit device pointer, or (host path, no hit) an upload.
struct MatchOps {
    int cur = 0;
    int mode = 0;                 // 0 host, other resident; 1 host, other uploaded; 2 device new + resident; 3 device new + explicit other
    bool new_is_query = true;
    int nq = 0, nt = 0;           // host-known sizes (upper bounds where a device count is given)
    const uint8_t* d_new = nullptr;     // device source of the new set (modes 2, 3)
    const uint8_t* d_other = nullptr;   // explicit other set (mode 3)
    const int* d_n_new = nullptr;       // device row count of the new set (pipelined form), or nullptr = host-known
    float ratio = 0.75f;
    // pipelined form: the kernels read the new set where the caller has it, the host waits for ev_done behind the result
    // copy, and the device copy that makes the set resident comes after that (the caller's next extraction is queued on
    // the same stream, so it cannot overtake the copy)
    bool in_place = false;
};

// kNN-2 in train slices, ratio test + compaction with the slice merge, one copy back ([count][matches]).
static int enqueue_match_ops(aria_matcher_s* m, const MatchOps& o) {
    const int cur = o.cur, n_new = o.new_is_query ? o.nq : o.nt;
    const uint8_t* d_other = m->d_pp[cur ^ 1];
    if (o.mode == 1) {
        ARIA_HIP(hipMemcpyAsync(m->d_t, m->h_stage, (size_t)o.nt * 32, hipMemcpyHostToDevice, m->stream));
        d_other = m->d_t;
    } else if (o.mode == 3) {
        d_other = o.d_other;
    }
    if (o.mode <= 1) ARIA_HIP(hipMemcpyAsync(m->d_pp[cur], m->h_pp[cur], (size_t)n_new * 32, hipMemcpyHostToDevice, m->stream));
    else if (!o.in_place) ARIA_HIP(hipMemcpyAsync(m->d_pp[cur], o.d_new, (size_t)n_new * 32, hipMemcpyDeviceToDevice, m->stream));
    const uint8_t* d_new = o.in_place ? o.d_new : m->d_pp[cur];
    const uint8_t* d_q = o.new_is_query ? d_new : d_other;
    const uint8_t* d_t = o.new_is_query ? d_other : d_new;
    const int* nq_arr = o.new_is_query ? o.d_n_new : nullptr;
    const int* nt_arr = o.new_is_query ? nullptr : o.d_n_new;
    int nsplit = 1;
    if (m->knn_valu) {
        launch_knn2(m, 0, o.nq, 1, d_q, nq_arr, o.nq, d_t, nt_arr, o.nt, (int64_t)0, (int64_t)0, m->d_keys, m->max_query, 0.0,
                    nullptr, o.nt);
    } else {
        nsplit = knn2_split_count(o.nq, o.nt);
        launch_knn2_mfma_split(m->stream, d_q, o.nq, d_t, o.nt, m->d_keys, m->max_query, nsplit, nq_arr, nt_arr);
    }
    hipLaunchKernelGGL(k_ratio_compact<1024>, dim3(1), dim3(1024), 0, m->stream, m->d_keys, nq_arr, o.nq, m->max_query, o.ratio,
                       reinterpret_cast<aria_match*>(m->d_res + 16), reinterpret_cast<int*>(m->d_res), m->max_query, m->d_err,
                       nullptr, nullptr, (int64_t)0, nullptr, nsplit);
    ARIA_HIP(hipGetLastError());
    ARIA_HIP(hipMemcpyAsync(m->h_res, m->d_res, 16 + sizeof(aria_match) * (size_t)o.nq, hipMemcpyDeviceToHost, m->stream));
    return ARIA_OK;
}

// Replays the captured graph of these operations when the key (sizes, ratio, mode, baked-in pointers) repeats for the
// slot; captures it when a key shows up the second time in a row; eager launches otherwise.
static int launch_match_ops(aria_matcher_s* m, const MatchOps& o) {
    static const bool want_graph = [] { const char* e = aria_getenv("ARIA_MATCH_GRAPH"); return !(e && e[0] == '0'); }();
    aria_matcher_s::MatchGraph& G = m->mg[o.cur];
    const int mode_key = o.mode * 8 + (o.new_is_query ? 0 : 1) + (o.d_n_new ? 2 : 0) + (o.in_place ? 4 : 0);
    const void* pa = o.d_new; const void* pb = o.d_n_new ? (const void*)o.d_n_new : (const void*)o.d_other;
    const bool same = G.nq == o.nq && G.nt == o.nt && G.hit == mode_key && G.ratio == o.ratio && G.pa == pa && G.pb == pb;
    const bool seen = G.seen_nq == o.nq && G.seen_nt == o.nt && G.seen_hit == mode_key && G.seen_ratio == o.ratio &&
                      G.seen_pa == pa && G.seen_pb == pb;
    if (want_graph && !m->graph_failed && G.exec && same) {
        ARIA_HIP(hipGraphLaunch(G.exec, m->stream));
    } else if (want_graph && !m->graph_failed && seen) {
        G.drop();
        ARIA_HIP(hipStreamSynchronize(m->stream));
        hipError_t e = hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal);
        int rc = ARIA_OK;
        if (e == hipSuccess) {
            rc = enqueue_match_ops(m, o);
            e = hipStreamEndCapture(m->stream, &G.graph);
        }
        if (e == hipSuccess && rc == ARIA_OK) e = hipGraphInstantiate(&G.exec, G.graph, nullptr, nullptr, 0);
        if (e != hipSuccess || rc != ARIA_OK || !G.exec) {
            G.drop();
            (void)hipGetLastError();
            m->graph_failed = true;
            rc = enqueue_match_ops(m, o);
            if (rc != ARIA_OK) return rc;
        } else {
            G.nq = o.nq; G.nt = o.nt; G.hit = mode_key; G.ratio = o.ratio; G.pa = pa; G.pb = pb;
            ARIA_HIP(hipGraphLaunch(G.exec, m->stream));
        }
    } else {
        int rc = enqueue_match_ops(m, o);
        if (rc != ARIA_OK) return rc;
    }
    G.seen_nq = o.nq; G.seen_nt = o.nt; G.seen_hit = mode_key; G.seen_ratio = o.ratio; G.seen_pa = pa; G.seen_pb = pb;
    if (o.in_place) {
        ARIA_HIP(hipEventRecord(m->ev_done, m->stream));
        const int n_new = o.new_is_query ? o.nq : o.nt;
        ARIA_HIP(hipMemcpyAsync(m->d_pp[o.cur], o.d_new, (size_t)n_new * 32, hipMemcpyDeviceToDevice, m->stream));
    }
    return ARIA_OK;
}

// stream is synchronised: hand the pinned result block to the caller
static int fetch_match_result(aria_matcher_s* m, aria_match* matches, int cap, int* n_out) {
    const int n = reinterpret_cast<const int*>(m->h_res)[0];
    *n_out = n;
    if (n > cap) return ARIA_E_OUTPUT_TOO_SMALL;
    if (n > 0) {
        if (!matches) return ARIA_E_INVALID;
        std::memcpy(matches, m->h_res + 16, sizeof(aria_match) * (size_t)n);
    }
    return ARIA_OK;
}

int aria_matcher_match(aria_matcher_t m, const uint8_t* q, int nq, const uint8_t* t, int nt, float ratio,
                       aria_match* matches, int cap, int* n_out) {
    if (!m || !n_out || nq < 0 || nt < 0 || cap < 0) return ARIA_E_INVALID;
    *n_out = 0;
    if (m->dev_pending) return ARIA_E_BUSY;
    if (nq == 0 || nt == 0) return ARIA_OK;   // CudaMatcher.cpp:35-37
    if (!q || !t) return ARIA_E_INVALID;
    if (nq > m->max_query || nt > m->max_train) return ARIA_E_TOO_LARGE;
    ARIA_HIP(hipSetDevice(m->device));
    // train set = the query set of the previous call (frame i-1)? then it is on the device already
    const int prev = m->pp_cur, cur = prev ^ 1;
    const bool hit = m->pp_host[prev] && m->pp_n[prev] == nt && nt <= m->max_query &&
                     std::memcmp(m->h_pp[prev], t, (size_t)nt * 32) == 0;
    if (!hit) std::memcpy(m->h_stage, t, (size_t)nt * 32);
    std::memcpy(m->h_pp[cur], q, (size_t)nq * 32);
    m->pp_n[cur] = nq;
    m->pp_host[cur] = true;
    m->pp_cur = cur;
    MatchOps o;
    o.cur = cur; o.mode = hit ? 0 : 1; o.new_is_query = true; o.nq = nq; o.nt = nt; o.ratio = ratio;
    int rc = launch_match_ops(m, o);
    if (rc != ARIA_OK) return rc;
    ARIA_HIP(hipStreamSynchronize(m->stream));
    return fetch_match_result(m, matches, cap, n_out);
}

// ---- device-resident single-pair forms: the role CudaMatcher::matchGpu was declared for (CudaMatcher.hpp:22-28) ----
static int match_device_setup(aria_matcher_s* m, const uint8_t* d_query, int nq, const uint8_t* d_train, int nt, float ratio,
                              const int* d_n_new, MatchOps* out) {
    if (m->dev_pending) return ARIA_E_BUSY;
    if (!d_query && !d_train) return ARIA_E_INVALID;
    if (nq > m->max_query || nt > m->max_train) return ARIA_E_TOO_LARGE;
    const int prev = m->pp_cur, cur = prev ^ 1;
    MatchOps o;
    o.cur = cur; o.nq = nq; o.nt = nt; o.ratio = ratio; o.d_n_new = d_n_new;
    if (d_query && d_train) {            // both explicit: the query becomes the resident set
        o.mode = 3; o.new_is_query = true; o.d_new = d_query; o.d_other = d_train;
    } else {
        o.mode = 2; o.new_is_query = d_query != nullptr; o.d_new = d_query ? d_query : d_train;
        const int n_res = o.new_is_query ? nt : nq;
        // with a device count the resident set's size is the caller's bound; otherwise it must be what is resident
        if (m->pp_n[prev] < 0 || n_res != m->pp_n[prev]) return ARIA_E_INVALID;
    }
    const int n_new = o.new_is_query ? nq : nt;
    if (n_new > m->max_query) return ARIA_E_TOO_LARGE;      // the ping-pong slots hold max_query rows
    *out = o;
    return ARIA_OK;
}

int aria_matcher_retain_device(aria_matcher_t m, const uint8_t* d_desc, int n) {
    if (!m || n < 0 || (n > 0 && !d_desc)) return ARIA_E_INVALID;
    if (m->dev_pending) return ARIA_E_BUSY;
    if (n > m->max_query) return ARIA_E_TOO_LARGE;
    ARIA_HIP(hipSetDevice(m->device));
    const int cur = m->pp_cur ^ 1;
    if (n > 0) ARIA_HIP(hipMemcpyAsync(m->d_pp[cur], d_desc, (size_t)n * 32, hipMemcpyDeviceToDevice, m->stream));
    ARIA_HIP(hipStreamSynchronize(m->stream));
    m->pp_n[cur] = n; m->pp_host[cur] = false; m->pp_cur = cur;
    return ARIA_OK;
}

int aria_matcher_resident_rows(aria_matcher_t m) { return m ? m->pp_n[m->pp_cur] : ARIA_E_INVALID; }

int aria_matcher_match_device(aria_matcher_t m, const uint8_t* d_query, int nq, const uint8_t* d_train, int nt, float ratio,
                              aria_match* matches, int cap, int* n_out) {
    if (!m || !n_out || nq < 0 || nt < 0 || cap < 0) return ARIA_E_INVALID;
    *n_out = 0;
    ARIA_HIP(hipSetDevice(m->device));
    if (nq == 0 || nt == 0) {                 // CudaMatcher.cpp:35-37; the new set still becomes the resident one
        const uint8_t* d_new = d_query ? d_query : d_train;
        return aria_matcher_retain_device(m, d_new, d_query ? nq : nt);
    }
    MatchOps o;
    int rc = match_device_setup(m, d_query, nq, d_train, nt, ratio, nullptr, &o);
    if (rc != ARIA_OK) return rc;
    m->pp_n[o.cur] = o.new_is_query ? nq : nt; m->pp_host[o.cur] = false; m->pp_cur = o.cur;
    rc = launch_match_ops(m, o);
    if (rc != ARIA_OK) return rc;
    ARIA_HIP(hipStreamSynchronize(m->stream));
    return fetch_match_result(m, matches, cap, n_out);
}

int aria_matcher_match_device_async(aria_matcher_t m, const uint8_t* d_new, const int* d_n_new, int n_new_max, int new_is_query,
                                    float ratio) {
    if (!m || !d_new || !d_n_new || n_new_max < 1) return ARIA_E_INVALID;
    if (m->dev_pending) return ARIA_E_BUSY;
    ARIA_HIP(hipSetDevice(m->device));
    const int n_res = m->pp_n[m->pp_cur];
    if (n_res < 1) return ARIA_E_INVALID;     // nothing resident to match against (aria_matcher_retain_device first)
    MatchOps o;
    int rc = match_device_setup(m, new_is_query ? d_new : nullptr, new_is_query ? n_new_max : n_res,
                                new_is_query ? nullptr : d_new, new_is_query ? n_res : n_new_max, ratio, d_n_new, &o);
    if (rc != ARIA_OK) return rc;
    o.in_place = true;
    rc = launch_match_ops(m, o);
    if (rc != ARIA_OK) return rc;
    m->pp_n[o.cur] = -1; m->pp_host[o.cur] = false; m->pp_cur = o.cur;     // row count: told by aria_matcher_finish
    m->dev_pending = true;
    return ARIA_OK;
}

int aria_matcher_finish(aria_matcher_t m, int n_new, aria_match* matches, int cap, int* n_out) {
    if (!m || !n_out || cap < 0 || n_new < 0) return ARIA_E_INVALID;
    *n_out = 0;
    if (!m->dev_pending) return ARIA_E_NOT_PENDING;
    ARIA_HIP(hipSetDevice(m->device));
    m->dev_pending = false;
    ARIA_HIP(hipEventSynchronize(m->ev_done));       // the result copy is done; the copy that keeps the set resident may still run
    m->pp_n[m->pp_cur] = n_new;
    if (n_new == 0) return ARIA_OK;           // CudaMatcher.cpp:35-37 (the kernels saw a zero count and wrote no match)
    return fetch_match_result(m, matches, cap, n_out);
}

int aria_matcher_knn2(aria_matcher_t m, const uint8_t* q, int nq, const uint8_t* t, int nt, int* idx, int* dist) {
    if (!m || nq < 0 || nt < 0) return ARIA_E_INVALID;
    if (nq == 0) return ARIA_OK;
    if (!q || !idx || !dist || (nt > 0 && !t)) return ARIA_E_INVALID;
    if (nq > m->max_query || nt > m->max_train) return ARIA_E_TOO_LARGE;
    ARIA_HIP(hipSetDevice(m->device));
    int rc = upload_pair(m, q, nq, t ? t : q, nt);
    if (rc != ARIA_OK) return rc;
    launch_knn2(m, 0, nq, 1, m->d_q, nullptr, nq, m->d_t, nullptr, nt, (int64_t)0, (int64_t)0, m->d_keys,
                m->max_query, 0.0, nullptr, nt);
    hipLaunchKernelGGL(k_unpack_knn, dim3((nq + 255) / 256), dim3(256), 0, m->stream, m->d_keys, nq, m->d_idx,
                       m->d_idx + 2 * (size_t)m->max_query);
    ARIA_HIP(hipGetLastError());
    ARIA_HIP(hipMemcpyAsync(m->h_idx, m->d_idx, sizeof(int) * 4 * (size_t)m->max_query, hipMemcpyDeviceToHost, m->stream));
    ARIA_HIP(hipStreamSynchronize(m->stream));
    std::memcpy(idx, m->h_idx, sizeof(int) * 2 * (size_t)nq);
    std::memcpy(dist, m->h_idx + 2 * (size_t)m->max_query, sizeof(int) * 2 * (size_t)nq);
    return ARIA_OK;
}

int aria_matcher_match_batch_device(aria_matcher_t m, const uint8_t* d_query, const int* d_nq, const uint8_t* d_train,
                                    const int* d_nt, int n_pairs, int64_t desc_stride, float ratio,
                                    aria_match* d_matches, int* d_nmatches, int match_cap) {
    if (!m || !d_query || !d_nq || !d_train || !d_nt || !d_matches || !d_nmatches || n_pairs < 0 || match_cap < 1 ||
        desc_stride < 32 || (desc_stride & 31))
        return ARIA_E_INVALID;
    if (n_pairs == 0) return ARIA_OK;
    const int64_t maxq = desc_stride / 32;
    if (maxq > 65535) return ARIA_E_TOO_LARGE;   // train index is packed in 16 bits
    ARIA_HIP(hipSetDevice(m->device));
    int rc = ensure_keys(m, (size_t)n_pairs * (size_t)maxq);
    if (rc != ARIA_OK) return rc;
    // profiling: drain the stream, then bracket the selected kernels with an event pair (see orb_kernels.h Profiler)
    aria_matcher_s::Ev ev;
    ev.pairs = n_pairs;
    const bool p0 = m->prof_enabled && (m->prof_mask & 1u), p1 = m->prof_enabled && (m->prof_mask & 2u);
    for (int s = 0; s < 4; s++) ev.e[s] = nullptr;
    if (p0) {
        ev.e[0] = m->prof_get(); ev.e[1] = m->prof_get();
        hipStreamSynchronize(m->stream);
        hipEventRecord(ev.e[0], m->stream);
    }
    launch_knn2(m, 0, (int)maxq, n_pairs, d_query, d_nq, 0, d_train, d_nt, 0, desc_stride,
                desc_stride, m->d_keys, (int)maxq, 0.0, nullptr, (int)maxq);
    if (p0) { hipEventRecord(ev.e[1], m->stream); hipStreamSynchronize(m->stream); }
    if (p1) {
        ev.e[2] = m->prof_get(); ev.e[3] = m->prof_get();
        hipStreamSynchronize(m->stream);
        hipEventRecord(ev.e[2], m->stream);
    }
    hipLaunchKernelGGL(k_ratio_compact<256>, dim3(n_pairs), dim3(256), 0, m->stream, m->d_keys, d_nq, 0, (int)maxq, ratio,
                       d_matches, d_nmatches, match_cap, m->d_err);
    if (p1) { hipEventRecord(ev.e[3], m->stream); hipStreamSynchronize(m->stream); }
    if (m->prof_enabled) m->prof_pending.push_back(ev);
    ARIA_HIP(hipGetLastError());
    return ARIA_OK;
}

int aria_flag_keypoints_shifted_device(void* stream, const aria_keypoint* d_kps, const int* d_counts, int n_frames, int kp_cap,
                                       const aria_box* d_boxes, const int* d_nboxes, int box_cap, int mode, int box_frame_offset,
                                       uint8_t* d_flags) {
    if (!d_kps || !d_counts || !d_boxes || !d_nboxes || !d_flags || n_frames < 0 || kp_cap < 1 || box_cap < 1 || (mode != 0 && mode != 1))
        return ARIA_E_INVALID;
    if (n_frames == 0) return ARIA_OK;
    hipLaunchKernelGGL(k_flag_keypoints, dim3((unsigned)((kp_cap + 255) / 256), (unsigned)n_frames), dim3(256), 0, (hipStream_t)stream,
                       d_kps, d_counts, kp_cap, reinterpret_cast<const float4*>(d_boxes), d_nboxes, box_cap, mode, d_flags,
                       box_frame_offset, n_frames);
    ARIA_HIP(hipGetLastError());
    return ARIA_OK;
}

int aria_flag_keypoints_device(void* stream, const aria_keypoint* d_kps, const int* d_counts, int n_frames, int kp_cap,
                               const aria_box* d_boxes, const int* d_nboxes, int box_cap, int mode, uint8_t* d_flags) {
    return aria_flag_keypoints_shifted_device(stream, d_kps, d_counts, n_frames, kp_cap, d_boxes, d_nboxes, box_cap, mode, 0, d_flags);
}

int aria_matcher_match_batch_filtered_device(aria_matcher_t m, const uint8_t* d_query, const int* d_nq, const uint8_t* d_train,
                                             const int* d_nt, int n_pairs, int64_t desc_stride, float ratio,
                                             const uint8_t* d_qflags, const uint8_t* d_tflags, int64_t flag_stride,
                                             aria_match* d_matches, int* d_nmatches, int match_cap, int* d_nfiltered) {
    if (!m || !d_query || !d_nq || !d_train || !d_nt || !d_matches || !d_nmatches || !d_qflags || !d_tflags || n_pairs < 0 ||
        match_cap < 1 || desc_stride < 32 || (desc_stride & 31) || flag_stride < desc_stride / 32)
        return ARIA_E_INVALID;
    if (n_pairs == 0) return ARIA_OK;
    const int64_t maxq = desc_stride / 32;
    if (maxq > 65535) return ARIA_E_TOO_LARGE;
    ARIA_HIP(hipSetDevice(m->device));
    int rc = ensure_keys(m, (size_t)n_pairs * (size_t)maxq);
    if (rc != ARIA_OK) return rc;
    launch_knn2(m, 0, (int)maxq, n_pairs, d_query, d_nq, 0, d_train, d_nt, 0, desc_stride, desc_stride, m->d_keys, (int)maxq,
                0.0, nullptr, (int)maxq);
    hipLaunchKernelGGL(k_ratio_compact<256>, dim3(n_pairs), dim3(256), 0, m->stream, m->d_keys, d_nq, 0, (int)maxq, ratio,
                       d_matches, d_nmatches, match_cap, m->d_err, d_qflags, d_tflags, flag_stride, d_nfiltered);
    ARIA_HIP(hipGetLastError());
    return ARIA_OK;
}

int aria_matcher_match_db_device(aria_matcher_t m, const uint8_t* d_query, int nq, const uint8_t* d_db,
                                 const int* d_kf_counts, int n_kf, int64_t desc_stride, double ratio, int* d_good) {
    if (!m || !d_query || !d_db || !d_kf_counts || !d_good || nq < 0 || n_kf < 0 || desc_stride < 32 || (desc_stride & 31))
        return ARIA_E_INVALID;
    if (desc_stride / 32 > 65535) return ARIA_E_TOO_LARGE;
    if (n_kf == 0) return ARIA_OK;
    ARIA_HIP(hipSetDevice(m->device));
    ARIA_HIP(hipMemsetAsync(d_good, 0, sizeof(int) * (size_t)n_kf, m->stream));
    if (nq == 0) return ARIA_OK;
    launch_knn2(m, 1, nq, n_kf, d_query, nullptr, nq, d_db, d_kf_counts, 0, (int64_t)0, desc_stride,
                nullptr, 0, ratio, d_good, (int)(desc_stride / 32));
    ARIA_HIP(hipGetLastError());
    return ARIA_OK;
}


// ---- one query against many candidates ----------------------------------------------------------------------
namespace {
// Uploads the query and every candidate's descriptors (candidate c at d_multi + c * rows * 32) and the row counts.
int stage_multi(aria_matcher_s* m, const uint8_t* q, int nq, const uint8_t* const* trains, const int* nts, int n_cand,
                int* rows_out) {
    int rows = 1;
    for (int c = 0; c < n_cand; c++) {
        if (nts[c] < 0 || (nts[c] > 0 && !trains[c])) return ARIA_E_INVALID;
        rows = std::max(rows, nts[c]);
    }
    if (rows > 65535) return ARIA_E_TOO_LARGE;
    if (nq > m->max_query) return ARIA_E_TOO_LARGE;
    const size_t need = (size_t)n_cand * rows * 32;
    if (need > m->multi_bytes) {
        ARIA_HIP(hipStreamSynchronize(m->stream));
        hipFree(m->d_multi); m->d_multi = nullptr; m->multi_bytes = 0;
        ARIA_HIP(hipMalloc(&m->d_multi, need));
        m->multi_bytes = need;
    }
    if (n_cand > m->mcnt_cap) {
        ARIA_HIP(hipStreamSynchronize(m->stream));
        hipFree(m->d_mcnt); m->d_mcnt = nullptr; m->mcnt_cap = 0;
        ARIA_HIP(hipMalloc(&m->d_mcnt, sizeof(int) * 2 * (size_t)n_cand));
        m->mcnt_cap = n_cand;
    }
    std::memcpy(m->h_stage, q, (size_t)nq * 32);
    ARIA_HIP(hipMemcpyAsync(m->d_q, m->h_stage, (size_t)nq * 32, hipMemcpyHostToDevice, m->stream));
    for (int c = 0; c < n_cand; c++)
        if (nts[c] > 0)
            ARIA_HIP(hipMemcpyAsync(m->d_multi + (size_t)c * rows * 32, trains[c], (size_t)nts[c] * 32, hipMemcpyHostToDevice, m->stream));
    ARIA_HIP(hipMemcpyAsync(m->d_mcnt, nts, sizeof(int) * (size_t)n_cand, hipMemcpyHostToDevice, m->stream));
    *rows_out = rows;
    return ARIA_OK;
}
}  // namespace

int aria_matcher_match_multi(aria_matcher_t m, const uint8_t* q, int nq, const uint8_t* const* trains, const int* nts,
                             int n_cand, float ratio, aria_match* matches, int cap_per_cand, int* n_out) {
    if (!m || nq < 0 || n_cand < 0 || cap_per_cand < 0 || (n_cand > 0 && (!trains || !nts || !n_out))) return ARIA_E_INVALID;
    for (int c = 0; c < n_cand; c++) n_out[c] = 0;
    if (n_cand == 0 || nq == 0) return ARIA_OK;            // CudaMatcher.cpp:35-37 for every candidate
    if (!q) return ARIA_E_INVALID;
    ARIA_HIP(hipSetDevice(m->device));
    int rows = 0;
    int rc = stage_multi(m, q, nq, trains, nts, n_cand, &rows);
    if (rc != ARIA_OK) return rc;
    rc = ensure_keys(m, (size_t)n_cand * (size_t)m->max_query);
    if (rc != ARIA_OK) return rc;
    if ((size_t)n_cand * (size_t)m->max_query > m->mm_entries) {
        ARIA_HIP(hipStreamSynchronize(m->stream));
        hipFree(m->d_mm); m->d_mm = nullptr; m->mm_entries = 0;
        ARIA_HIP(hipMalloc(&m->d_mm, sizeof(aria_match) * (size_t)n_cand * (size_t)m->max_query));
        m->mm_entries = (size_t)n_cand * (size_t)m->max_query;
    }
    // one launch over all candidates: the query block is shared (stride 0), candidate c is train block c
    launch_knn2(m, 0, nq, n_cand, m->d_q, nullptr, nq, m->d_multi, m->d_mcnt, 0, (int64_t)0, (int64_t)rows * 32, m->d_keys,
                m->max_query, 0.0, nullptr, rows);
    hipLaunchKernelGGL(k_ratio_compact<256>, dim3(n_cand), dim3(256), 0, m->stream, m->d_keys, nullptr, nq, m->max_query, ratio,
                       m->d_mm, m->d_mcnt + n_cand, m->max_query, m->d_err);
    ARIA_HIP(hipGetLastError());
    std::vector<int> cnt((size_t)n_cand);
    ARIA_HIP(hipMemcpyAsync(cnt.data(), m->d_mcnt + n_cand, sizeof(int) * (size_t)n_cand, hipMemcpyDeviceToHost, m->stream));
    ARIA_HIP(hipStreamSynchronize(m->stream));
    int status = ARIA_OK;
    for (int c = 0; c < n_cand; c++) {
        n_out[c] = cnt[(size_t)c];
        if (cnt[(size_t)c] > cap_per_cand) { status = ARIA_E_OUTPUT_TOO_SMALL; continue; }
        if (cnt[(size_t)c] > 0) {
            if (!matches) return ARIA_E_INVALID;
            ARIA_HIP(memcpy_on(m->stream, matches + (size_t)c * cap_per_cand, m->d_mm + (size_t)c * m->max_query,
                               sizeof(aria_match) * (size_t)cnt[(size_t)c], hipMemcpyDeviceToHost));
        }
    }
    return status;
}

int aria_matcher_count_good_multi(aria_matcher_t m, const uint8_t* q, int nq, const uint8_t* const* trains, const int* nts,
                                  int n_cand, double ratio, int* good) {
    if (!m || nq < 0 || n_cand < 0 || (n_cand > 0 && (!trains || !nts || !good))) return ARIA_E_INVALID;
    for (int c = 0; c < n_cand; c++) good[c] = 0;
    if (n_cand == 0 || nq == 0) return ARIA_OK;
    if (!q) return ARIA_E_INVALID;
    ARIA_HIP(hipSetDevice(m->device));
    int rows = 0;
    int rc = stage_multi(m, q, nq, trains, nts, n_cand, &rows);
    if (rc != ARIA_OK) return rc;
    rc = aria_matcher_match_db_device(m, m->d_q, nq, m->d_multi, m->d_mcnt, n_cand, (int64_t)rows * 32, ratio, m->d_mcnt + n_cand);
    if (rc != ARIA_OK) return rc;
    ARIA_HIP(hipMemcpyAsync(good, m->d_mcnt + n_cand, sizeof(int) * (size_t)n_cand, hipMemcpyDeviceToHost, m->stream));
    ARIA_HIP(hipStreamSynchronize(m->stream));
    return ARIA_OK;
}

// ---- HBM-resident keyframe descriptor database (LoopClosureDetector's deque, src/legacy/LoopClosure.cpp:24-31) ----
struct aria_kfdb_s {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int cap = 0, rows = 0;
    int head = 0, n = 0;                 // ring: logical keyframe i lives in slot (head + i) % cap
    uint8_t* d_desc = nullptr;           // [cap][rows * 32]
    int* d_counts = nullptr;             // [cap]
    int* d_good = nullptr;               // [cap]
    std::vector<long long> ids;          // per slot
    std::vector<int> counts;             // per slot (host mirror)
};

int aria_kfdb_create(int device, void* stream, int capacity, int rows, aria_kfdb_t* out) {
    if (!out || capacity < 1 || rows < 1 || rows > 65535) return ARIA_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    ARIA_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return ARIA_E_NO_DEVICE;
    ARIA_HIP(hipSetDevice(device));
    aria_kfdb_s* db = new (std::nothrow) aria_kfdb_s();
    if (!db) return ARIA_E_OOM;
    db->device = device; db->cap = capacity; db->rows = rows;
    db->ids.assign((size_t)capacity, -1); db->counts.assign((size_t)capacity, 0);
    if (stream) db->stream = (hipStream_t)stream;
    else {
        hipError_t e = create_stream(&db->stream);
        if (e != hipSuccess) { delete db; return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
        db->owns_stream = true;
    }
    hipError_t e = hipMalloc(&db->d_desc, (size_t)capacity * rows * 32);
    if (e == hipSuccess) e = hipMalloc(&db->d_counts, sizeof(int) * (size_t)capacity);
    if (e == hipSuccess) e = hipMalloc(&db->d_good, sizeof(int) * (size_t)capacity);
    if (e == hipSuccess) e = memset_on(db->stream, db->d_counts, 0, sizeof(int) * (size_t)capacity);
    if (e != hipSuccess) { aria_kfdb_destroy(db); return hip_fail(e, "aria_kfdb_create", __FILE__, __LINE__); }
    *out = db;
    return ARIA_OK;
}

void aria_kfdb_destroy(aria_kfdb_t db) {
    if (!db) return;
    hipSetDevice(db->device);
    if (db->stream) hipStreamSynchronize(db->stream);
    hipFree(db->d_desc); hipFree(db->d_counts); hipFree(db->d_good);
    if (db->owns_stream && db->stream) hipStreamDestroy(db->stream);
    delete db;
}

int aria_kfdb_size(aria_kfdb_t db) { return db ? db->n : ARIA_E_INVALID; }

static int kfdb_add(aria_kfdb_s* db, long long id, const uint8_t* desc, int n, hipMemcpyKind kind) {
    if (!db || n < 0 || (n > 0 && !desc)) return ARIA_E_INVALID;
    if (n > db->rows) return ARIA_E_TOO_LARGE;
    ARIA_HIP(hipSetDevice(db->device));
    int slot;
    if (db->n == db->cap) { slot = db->head; db->head = (db->head + 1) % db->cap; }      // pop_front, LoopClosure.cpp:28-30
    else { slot = (db->head + db->n) % db->cap; db->n++; }
    if (n > 0) ARIA_HIP(hipMemcpyAsync(db->d_desc + (size_t)slot * db->rows * 32, desc, (size_t)n * 32, kind, db->stream));
    ARIA_HIP(hipMemcpyAsync(db->d_counts + slot, &n, sizeof(int), hipMemcpyHostToDevice, db->stream));
    ARIA_HIP(hipStreamSynchronize(db->stream));          // the caller may free its buffer; &n is a stack address
    db->ids[(size_t)slot] = id;
    db->counts[(size_t)slot] = n;
    return ARIA_OK;
}
int aria_kfdb_add(aria_kfdb_t db, long long id, const uint8_t* desc, int n) { return kfdb_add(db, id, desc, n, hipMemcpyHostToDevice); }
int aria_kfdb_add_device(aria_kfdb_t db, long long id, const uint8_t* d_desc, int n) { return kfdb_add(db, id, d_desc, n, hipMemcpyDeviceToDevice); }

int aria_kfdb_info(aria_kfdb_t db, int index, long long* id, int* count) {
    if (!db || index < 0 || index >= db->n) return ARIA_E_INVALID;
    const int slot = (db->head + index) % db->cap;
    if (id) *id = db->ids[(size_t)slot];
    if (count) *count = db->counts[(size_t)slot];
    return ARIA_OK;
}

int aria_kfdb_fetch(aria_kfdb_t db, int index, uint8_t* desc, int cap_rows, int* n_out) {
    if (!db || index < 0 || index >= db->n || cap_rows < 0) return ARIA_E_INVALID;
    const int slot = (db->head + index) % db->cap;
    const int n = db->counts[(size_t)slot];
    if (n_out) *n_out = n;
    if (n > cap_rows) return ARIA_E_OUTPUT_TOO_SMALL;
    ARIA_HIP(hipSetDevice(db->device));
    if (n > 0) {
        if (!desc) return ARIA_E_INVALID;
        ARIA_HIP(memcpy_on(db->stream, desc, db->d_desc + (size_t)slot * db->rows * 32, (size_t)n * 32, hipMemcpyDeviceToHost));
    }
    return ARIA_OK;
}

// LoopClosureDetector::verifyGeometry's match list (src/legacy/LoopClosure.cpp:120-131): the query against keyframe
// `index` where it lies in the database -- the keyframe's descriptors are neither downloaded nor uploaded again.
int aria_kfdb_match(aria_kfdb_t db, aria_matcher_t m, int index, const uint8_t* q, int nq, float ratio, aria_match* matches,
                    int cap, int* n_out) {
    if (!db || !m || !n_out || index < 0 || index >= db->n || nq < 0 || cap < 0 || (nq > 0 && !q)) return ARIA_E_INVALID;
    if (db->device != m->device) return ARIA_E_INVALID;
    *n_out = 0;
    if (m->dev_pending) return ARIA_E_BUSY;
    const int slot = (db->head + index) % db->cap;
    const int nt = db->counts[(size_t)slot];
    if (nq == 0 || nt == 0) return ARIA_OK;                  // CudaMatcher.cpp:35-37
    if (nq > m->max_query || nt > m->max_train) return ARIA_E_TOO_LARGE;
    ARIA_HIP(hipSetDevice(db->device));
    std::memcpy(m->h_stage, q, (size_t)nq * 32);
    ARIA_HIP(hipMemcpyAsync(m->d_q, m->h_stage, (size_t)nq * 32, hipMemcpyHostToDevice, m->stream));
    launch_knn2(m, 0, nq, 1, m->d_q, nullptr, nq, db->d_desc + (size_t)slot * db->rows * 32, nullptr, nt, (int64_t)0, (int64_t)0,
                m->d_keys, m->max_query, 0.0, nullptr, nt);
    hipLaunchKernelGGL(k_ratio_compact<1024>, dim3(1), dim3(1024), 0, m->stream, m->d_keys, nullptr, nq, m->max_query, ratio,
                       reinterpret_cast<aria_match*>(m->d_res + 16), reinterpret_cast<int*>(m->d_res), m->max_query, m->d_err);
    ARIA_HIP(hipGetLastError());
    ARIA_HIP(hipMemcpyAsync(m->h_res, m->d_res, 16 + sizeof(aria_match) * (size_t)nq, hipMemcpyDeviceToHost, m->stream));
    ARIA_HIP(hipStreamSynchronize(m->stream));
    return fetch_match_result(m, matches, cap, n_out);
}

int aria_kfdb_scan(aria_kfdb_t db, aria_matcher_t m, const uint8_t* q, int nq, double ratio, int* good, int cap, int* n_out) {
    if (!db || !m || nq < 0 || cap < 0 || (nq > 0 && !q)) return ARIA_E_INVALID;
    if (db->device != m->device) return ARIA_E_INVALID;
    if (n_out) *n_out = db->n;
    if (db->n > cap) return ARIA_E_OUTPUT_TOO_SMALL;
    if (db->n == 0) return ARIA_OK;
    if (!good) return ARIA_E_INVALID;
    if (nq > m->max_query) return ARIA_E_TOO_LARGE;
    ARIA_HIP(hipSetDevice(db->device));
    std::vector<int> phys((size_t)db->cap, 0);
    if (nq > 0) {
        std::memcpy(m->h_stage, q, (size_t)nq * 32);
        ARIA_HIP(hipMemcpyAsync(m->d_q, m->h_stage, (size_t)nq * 32, hipMemcpyHostToDevice, m->stream));
        // every physical slot is scanned (empty ones hold count 0); one launch for the whole database
        int rc = aria_matcher_match_db_device(m, m->d_q, nq, db->d_desc, db->d_counts, db->cap, (int64_t)db->rows * 32, ratio, db->d_good);
        if (rc != ARIA_OK) return rc;
        ARIA_HIP(hipMemcpyAsync(phys.data(), db->d_good, sizeof(int) * (size_t)db->cap, hipMemcpyDeviceToHost, m->stream));
        ARIA_HIP(hipStreamSynchronize(m->stream));
    }
    for (int i = 0; i < db->n; i++) good[i] = phys[(size_t)((db->head + i) % db->cap)];
    return ARIA_OK;
}

}  // extern "C"
