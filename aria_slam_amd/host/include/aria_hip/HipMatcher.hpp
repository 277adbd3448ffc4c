// aria::adapters::hip::HipMatcher -- MI355X drop-in for aria::adapters::gpu::CudaMatcher
// (reference include/adapters/gpu/CudaMatcher.hpp, src/adapters/gpu/CudaMatcher.cpp): brute-force Hamming kNN-2 +
// Lowe ratio through the C-ABI. Tie-breaking follows CPU cv::BFMatcher (lower train index first).
#pragma once
#include <utility>

#include "aria_hip/compat.hpp"

struct aria_matcher_s;

namespace aria::adapters::hip {

class HipMatcher : public interfaces::IMatcher {
public:
    explicit HipMatcher(void* stream = nullptr, int device = 0);   // CudaMatcher.cpp:7-19
    ~HipMatcher() override;
    HipMatcher(const HipMatcher&) = delete;
    HipMatcher& operator=(const HipMatcher&) = delete;

    // Appends to `matches`, never clears (CudaMatcher.cpp:65); empty input leaves it untouched (:35-37).
    // ratio_threshold 0 = test disabled, as IMatcher.hpp:18 documents (see INTEGRATION.md for the divergence
    // from the reference adapter, which would return nothing).
    void match(const core::Frame& query, const core::Frame& train, std::vector<core::Match>& matches,
               float ratio_threshold = 0.75f) override;

    // One batch on the device instead of the interface's default loop (IMatcher.hpp:27-37): the query is uploaded once,
    // ONE kNN-2 launch covers every candidate. Resizes the outer vector and appends per candidate, like the default.
    void matchMultiple(const core::Frame& query, const std::vector<core::Frame>& candidates,
                       std::vector<std::vector<core::Match>>& all_matches, float ratio_threshold = 0.75f) override;

    // LoopClosureDetector::findCandidates semantics (reference src/legacy/LoopClosure.cpp:72-114) over clean-
    // architecture frames: kNN-2 of the query against every keyframe, ratio 0.7 in double, score = good/|query|,
    // keep > 0.1, best 5. Returns (index into keyframes, score).
    std::vector<std::pair<int, double>> findLoopCandidates(const core::Frame& query,
                                                          const std::vector<core::Frame>& keyframes,
                                                          int min_frames_between);

    // CudaMatcher::matchGpu (include/adapters/gpu/CudaMatcher.hpp:22-28, "GPU-to-GPU matching, zero-copy when used with
    // OrbCudaExtractor"): descriptor sets given as device pointers (OrbHipExtractor::deviceResult()), matches appended on
    // the host. The matcher keeps ONE set resident on the device between calls; nullptr for d_query or d_train stands for
    // it (its row count must then be nq / nt). Afterwards the resident set is the non-null one (the query if both are
    // given): frame i against frame i-1 needs no descriptor upload at all.
    void matchDevice(const std::uint8_t* d_query, int nq, const std::uint8_t* d_train, int nt, std::vector<core::Match>& matches,
                     float ratio_threshold = 0.75f);
    void retainDevice(const std::uint8_t* d_desc, int n);      // make a set resident without matching (first frame)
    int residentRows() const;                                   // -1: nothing resident
    // Pipelined form, for an extractor and a matcher constructed on the SAME stream: queue the match of the new set
    // (row count read on the device from *d_count, at most rows_max) against the resident set behind the extraction, then
    // finishDevice() once the frame's keypoint count is known. Returns false (nothing queued) when nothing is resident.
    bool matchDeviceAsync(const std::uint8_t* d_new, const int* d_count, int rows_max, bool new_is_query, float ratio_threshold = 0.75f);
    void finishDevice(int n_new, std::vector<core::Match>& matches);

    aria_matcher_s* handle() const { return m_; }
    void* streamArg() const { return stream_; }
    int device() const { return device_; }
    void reserve(int nq, int nt) { ensure(nq, nt); }       // creates / grows the C handle
    // rows per set the handle accepts as it is; growing it (reserve, or a larger call) drops the resident set
    bool fits(int nq, int nt) const { return m_ && nq <= cap_q_ && nt <= cap_t_; }

private:
    void ensure(int nq, int nt);
    aria_matcher_s* m_ = nullptr;
    void* stream_;
    int device_;
    int cap_q_ = 0, cap_t_ = 0;
    std::vector<core::Match> buf_;
};

}  // namespace aria::adapters::hip
