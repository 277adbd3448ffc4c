// Stand-in for the two reference headers the adapters need, for builds where the reference tree (and Eigen,
// which its core/Types.hpp includes) is not on the include path -- e.g. this repository's own tests.
//
// In an aria-slam checkout, compile the adapters with -DARIA_HIP_USE_REFERENCE_HEADERS -I<aria-slam>/include and
// this file forwards to the real ports (include/interfaces/IFeatureExtractor.hpp, include/interfaces/IMatcher.hpp,
// include/core/Types.hpp). Otherwise it declares layout-compatible minimal versions of exactly the members the
// hot path touches: KeyPoint (24 B), Match (12 B), the Frame fields the extractor fills, and the two ports.
#pragma once

#ifdef ARIA_HIP_USE_REFERENCE_HEADERS
#include "interfaces/IFeatureExtractor.hpp"
#include "interfaces/ILoopDetector.hpp"
#include "interfaces/IMatcher.hpp"
#else

#include <cstddef>
#include <cstdint>
#include <memory>
#include <optional>
#include <vector>

namespace aria::core {

struct KeyPoint {
    float x, y, size, angle, response;
    int octave;
};

struct Frame {
    std::uint64_t id = 0;
    double timestamp = 0.0;
    int width = 0, height = 0;
    std::vector<KeyPoint> keypoints;
    std::vector<std::uint8_t> descriptors;                    // N x 32, row-major
    alignas(16) double pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};   // Eigen::Matrix4d in the reference
    std::size_t descriptorSize() const { return 32; }
    std::size_t numKeypoints() const { return keypoints.size(); }
};

struct Match {
    int query_idx, train_idx;
    float distance;
};

// include/core/Types.hpp:103-112 (the box test of SlamPipeline::filterDynamicKeypoints, SlamPipeline.hpp:96-99)
struct Detection {
    float x1, y1, x2, y2;
    float confidence;
    int class_id;
    bool contains(float x, float y) const { return x >= x1 && x <= x2 && y >= y1 && y <= y2; }
};

// include/core/Types.hpp:33-45: the members the loop-closure path touches (pose members are Eigen types there)
struct KeyFrame {
    std::uint64_t id = 0;
    double timestamp = 0.0;
    Frame frame;
    alignas(16) double pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};

// include/core/Types.hpp:114-121
struct LoopCandidate {
    std::uint64_t query_id = 0;
    std::uint64_t match_id = 0;
    double score = 0.0;
    std::vector<Match> matches;
    alignas(16) double relative_pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};

}  // namespace aria::core

namespace aria::interfaces {

class IFeatureExtractor {
public:
    virtual ~IFeatureExtractor() = default;
    virtual void extract(const std::uint8_t* image_data, int width, int height, core::Frame& frame) = 0;
    virtual void extractAsync(const std::uint8_t* image_data, int width, int height, core::Frame& frame) {
        extract(image_data, width, height, frame);
    }
    virtual void sync() {}
    virtual void setMaxFeatures(int n) = 0;
    virtual int getMaxFeatures() const = 0;
};
using FeatureExtractorPtr = std::unique_ptr<IFeatureExtractor>;

class IMatcher {
public:
    virtual ~IMatcher() = default;
    virtual void match(const core::Frame& query, const core::Frame& train, std::vector<core::Match>& matches,
                       float ratio_threshold = 0.75f) = 0;
    virtual void matchMultiple(const core::Frame& query, const std::vector<core::Frame>& candidates,
                               std::vector<std::vector<core::Match>>& all_matches, float ratio_threshold = 0.75f) {
        all_matches.resize(candidates.size());
        for (std::size_t i = 0; i < candidates.size(); i++) match(query, candidates[i], all_matches[i], ratio_threshold);
    }
};
using MatcherPtr = std::unique_ptr<IMatcher>;

// include/interfaces/ILoopDetector.hpp:11-31
class ILoopDetector {
public:
    virtual ~ILoopDetector() = default;
    virtual void addKeyFrame(const core::KeyFrame& kf) = 0;
    virtual std::optional<core::LoopCandidate> detect(const core::KeyFrame& query) = 0;
    virtual int getLoopCount() const = 0;
    virtual void setMinFramesBetween(int n) = 0;
    virtual void setMinScore(double s) = 0;
    virtual void setMinMatches(int n) = 0;
};
using LoopDetectorPtr = std::unique_ptr<ILoopDetector>;

}  // namespace aria::interfaces

#endif  // ARIA_HIP_USE_REFERENCE_HEADERS
