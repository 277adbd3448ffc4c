// aria::pipeline::FrontEnd -- the front half of SlamPipeline::processFrame over the reference's ports
// (SURVEY.md 8f row 1). The reference declares SlamPipeline (include/pipeline/SlamPipeline.hpp:29-106) but ships no
// implementation; its sketch of processFrame (docs/milestones/H12_CLEAN_ARCHITECTURE.md:595-605) is
//     extractor_->extract(image_data, width, height, frame);
//     matcher_->match(frame, *prev_frame_, matches);          // query = current, train = previous
// followed by pose / fusion / mapping stages that are out of scope here. FrontEnd runs exactly those two calls,
// keeps the previous frame, and hands {frame, matches} to the caller (or a callback) -- so any IFeatureExtractor /
// IMatcher pair (the HIP adapters, or mocks) can be driven the way SlamPipeline is designed to drive them.
#pragma once
#include <functional>
#include <memory>
#include <optional>
#include <vector>

#include "aria_hip/compat.hpp"

namespace aria::pipeline {

struct FrontEndConfig {
    float ratio_threshold = 0.75f;     // IMatcher default (IMatcher.hpp:23); euroc_eval uses 0.75 too (euroc_eval.cpp:172)
    // false: query = current, train = previous (SlamPipeline sketch, H12...:601)
    // true : query = previous, train = current (what the legacy executables do, src/euroc_eval.cpp:168-169)
    bool legacy_order = false;
    // Loop-closure step of the evaluation loop (src/euroc_eval.cpp:230-245), run when a loop detector is injected: a frame
    // whose match list has at least keyframe_min_matches entries (the reference's pose stage needs 8 points,
    // euroc_eval.cpp:179) becomes a keyframe -- detect() against the database first, then addKeyFrame().
    int keyframe_min_matches = 8;
    // PipelineConfig::filter_dynamic_objects (include/pipeline/SlamPipeline.hpp:20): matches with an endpoint inside a
    // detection of a dynamic class are dropped, as the legacy executable does (src/main.cpp:29-50, 164-175). The detector is
    // not part of this repository: its boxes for the current frame arrive through FrontEnd::setDetections().
    bool filter_dynamic_objects = true;
    // When the injected pair is OrbHipExtractor + HipMatcher on one device, the descriptors stay on the device between the
    // two calls -- the getGpuDescriptors() / matchGpu hand-off the reference declares (OrbCudaExtractor.hpp:34-35,
    // CudaMatcher.hpp:22-28): no descriptor upload per frame. If both were also constructed on the SAME stream
    // (factory::createHip does that) the match is queued behind extractAsync and one wait covers both (the async shape of
    // docs/milestones/H12_CLEAN_ARCHITECTURE.md:711-716). Results are identical either way; false = always the host port calls.
    bool device_handoff = true;
};

// COCO ids of src/main.cpp:29-40: person, bicycle, car, motorcycle, bus, train, truck, bird, cat, dog
bool isDynamicClass(int class_id);

struct FrontEndResult {
    const core::Frame* frame = nullptr;       // the frame just extracted (owned by the FrontEnd until the next call)
    const core::Frame* previous = nullptr;    // nullptr on the first frame
    std::vector<core::Match> matches;         // empty on the first frame
    int filtered_count = 0;                   // matches dropped by the dynamic-object filter (main.cpp:172)
    bool is_keyframe = false;                 // the frame was handed to the loop detector
    std::optional<core::LoopCandidate> loop;  // what ILoopDetector::detect returned for it
};

class FrontEnd {
public:
    FrontEnd(interfaces::FeatureExtractorPtr extractor, interfaces::MatcherPtr matcher, const FrontEndConfig& cfg = {});
    // with the loop detector SlamPipeline's constructor takes third (include/pipeline/SlamPipeline.hpp:32-40)
    FrontEnd(interfaces::FeatureExtractorPtr extractor, interfaces::MatcherPtr matcher, interfaces::LoopDetectorPtr loop_detector,
             const FrontEndConfig& cfg = {});

    // image_data: grayscale, row-major, width*height bytes (the extractor port's contract, IFeatureExtractor.hpp:14)
    const FrontEndResult& processFrame(const std::uint8_t* image_data, int width, int height, double timestamp);

    // Detections of the NEXT frame to be processed (the reference runs YOLO beside ORB on the same image, main.cpp:132-150);
    // consumed by that processFrame call.
    void setDetections(std::vector<core::Detection> detections) { detections_ = std::move(detections); }
    void setCallback(std::function<void(const FrontEndResult&)> cb) { callback_ = std::move(cb); }
    std::uint64_t framesProcessed() const { return next_id_; }
    interfaces::IFeatureExtractor& extractor() { return *extractor_; }
    interfaces::IMatcher& matcher() { return *matcher_; }
    interfaces::ILoopDetector* loopDetector() { return loop_detector_.get(); }
    // Something that must outlive the components (factory::createHip: the stream they share); released last.
    void setKeepAlive(std::shared_ptr<void> k) { keep_alive_ = std::move(k); }
    // 0: host port calls, 1: device hand-off (two waits), 2: device hand-off queued behind extractAsync (one wait)
    int handoffMode() const { return handoff_mode_; }

private:
    void extractAndMatch(const std::uint8_t* image_data, int width, int height, core::Frame& f);
    std::shared_ptr<void> keep_alive_;            // declared first: destroyed after the components below
    interfaces::FeatureExtractorPtr extractor_;
    interfaces::MatcherPtr matcher_;
    interfaces::LoopDetectorPtr loop_detector_;
    FrontEndConfig cfg_;
    std::vector<core::Detection> detections_;
    std::unique_ptr<core::Frame> cur_, prev_;
    FrontEndResult result_;
    std::function<void(const FrontEndResult&)> callback_;
    std::uint64_t next_id_ = 0;
    int handoff_mode_ = 0;
};

}  // namespace aria::pipeline
