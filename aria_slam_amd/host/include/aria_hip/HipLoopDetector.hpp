// aria::adapters::hip::HipLoopDetector -- the "LoopDetectorGpu" that include/interfaces/ILoopDetector.hpp:10 names and the
// reference never wrote (docs/milestones/H14_GPU_LOOPCLOSURE_AUDIT.md designs it): LoopClosureDetector's candidate search
// (src/legacy/LoopClosure.cpp:24-31, 33-70, 72-114) with the keyframe descriptors resident in HBM and the scan of the
// whole database as ONE kernel launch (aria_kfdb_scan), instead of one CPU BFMatcher::knnMatch per keyframe.
//
// What is kept from the reference: the 500-keyframe deque with pop_front (:28-30); detect() returns nothing until the
// database holds min_frames_between keyframes (:34-36); candidates = keyframes at least min_frames_between older than
// the query (:81) with descriptors (:83), kNN-2 + ratio 0.7 in double (:92), score = good / max(1, |query keypoints|)
// (:98), kept above 0.1 (:99), best five by score (:105-111); then per candidate: score >= min_score (:42), and a
// verification step. The reference's verification is geometric (essential-matrix RANSAC in OpenCV, :116-190), which is
// outside the feature front-end: here a candidate is accepted when its ratio-0.7 match list has at least min_matches
// entries (the same bound verifyGeometry applies to its inliers), unless the caller installs a verifier (setVerifier),
// which receives the query keyframe, the matched keyframe's id and the candidate (match list filled) and may reject it or fill in
// the relative pose.
#pragma once
#include <functional>
#include <optional>
#include <utility>
#include <vector>

#include "aria_hip/HipMatcher.hpp"
#include "aria_hip/compat.hpp"

struct aria_kfdb_s;

namespace aria::adapters::hip {

class HipLoopDetector : public interfaces::ILoopDetector {
public:
    // defaults of LoopClosureDetector (include/legacy/LoopClosure.hpp); euroc_eval uses (200, 0.4, 50), euroc_eval.cpp:103.
    // slot_rows: descriptors per keyframe slot (>= the extractor's aria_orb_kp_capacity()).
    explicit HipLoopDetector(int min_frames_between = 30, double min_score = 0.3, int min_matches = 30, int slot_rows = 4096,
                             int capacity = 500, void* stream = nullptr, int device = 0);
    ~HipLoopDetector() override;
    HipLoopDetector(const HipLoopDetector&) = delete;
    HipLoopDetector& operator=(const HipLoopDetector&) = delete;

    void addKeyFrame(const core::KeyFrame& kf) override;                                  // LoopClosure.cpp:24-31
    std::optional<core::LoopCandidate> detect(const core::KeyFrame& query) override;      // :33-70
    int getLoopCount() const override { return loop_count_; }
    void setMinFramesBetween(int n) override { min_frames_between_ = n; }
    void setMinScore(double s) override { min_score_ = s; }
    void setMinMatches(int n) override { min_matches_ = n; }

    // (index into the database, oldest first; score): LoopClosure.cpp:72-114
    std::vector<std::pair<int, double>> findCandidates(const core::KeyFrame& query);
    int size() const;
    std::uint64_t keyframeId(int index) const;

    // May reject the candidate (return false) or refine it: candidate.matches holds the ratio-0.7 match list on entry,
    // candidate.relative_pose is the verifier's to fill in (Eigen::Matrix4d in the reference's core/Types.hpp:120; the
    // candidate is passed whole so that this header does not have to name that type).
    using Verifier = std::function<bool(const core::KeyFrame& query, std::uint64_t match_id, core::LoopCandidate& candidate)>;
    void setVerifier(Verifier v) { verifier_ = std::move(v); }

private:
    aria_kfdb_s* db_ = nullptr;
    HipMatcher matcher_;
    int min_frames_between_;
    double min_score_;
    int min_matches_;
    int slot_rows_;
    int loop_count_ = 0;
    Verifier verifier_;
    std::vector<int> good_;
    std::vector<core::Match> match_buf_;
};

}  // namespace aria::adapters::hip
