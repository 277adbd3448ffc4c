// Internal interface between the matcher's C-ABI (match_hip.hip) and its kNN-2 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace aria {

// Matrix-core kNN-2 (knn2_mfma.hip). Same contract as k_knn2<MODE> in match_hip.hip, for n_pairs pairs of at most
// nq_max queries: mode 0 stores (best, runner-up) keys (distance << 16 | train index, 0xFFFFFFFF = none) per query,
// mode 1 counts the queries passing the double-precision ratio test into good[pair]. max_train = upper bound of
// every pair's train count (selects the key layout). Returns the name of the kernel form it launched ("k_knn2_fp4",
// "k_knn2_mfma", or "k_knn2_fp4|k_knn2_mfma" when both layouts were launched behind the device-side gate: the FP4 one does
// the batch unless a pair's train count exceeds 4096), for aria_matcher_knn_kernel.
const char* launch_knn2_mfma(int mode, int nq_max, int n_pairs, hipStream_t st, const uint8_t* q, const int* nq_arr, int nq_fixed,
                      const uint8_t* t, const int* nt_arr, int nt_fixed, int64_t q_stride, int64_t t_stride,
                      uint2* keys, int maxq, double ratio, int* good, int max_train, int* gate = nullptr);

// Latency schedule for ONE pair (frame-at-a-time host path): the train set is cut into nsplit slices of whole 64-train
// tiles, every (query block, slice) is a workgroup, slice s writes keys[s * maxq + query] with GLOBAL train indices;
// the consumer takes the two smallest of the 2 nsplit keys of a query (k_ratio_compact's nsplit argument). A single
// 2000 x 2000 pair is 8 workgroups on 256 CUs otherwise. knn2_split_count picks nsplit (<= kKnnSplitMax).
constexpr int kKnnSplitMax = 16;
int knn2_split_count(int nq, int nt);
// nq_arr / nt_arr (optional): device row counts of the query / train set (one int); nq / nt are then upper bounds that
// size the grid and pick the key layout.
void launch_knn2_mfma_split(hipStream_t st, const uint8_t* q, int nq, const uint8_t* t, int nt, uint2* keys, int maxq,
                            int nsplit, const int* nq_arr = nullptr, const int* nt_arr = nullptr);

}  // namespace aria
