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

    aria_matcher_s* handle() const { return m_; }
    void reserve(int nq, int nt) { ensure(nq, nt); }       // creates / grows the C handle

private:
    void ensure(int nq, int nt);
    aria_matcher_s* m_ = nullptr;
    void* stream_;
    int device_;
    int cap_q_ = 0, cap_t_ = 0;
    std::vector<core::Match> buf_;
};

}  // namespace aria::adapters::hip
