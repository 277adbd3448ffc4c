// See aria_hip/FrontEnd.hpp.
#include "aria_hip/FrontEnd.hpp"

#include <stdexcept>
#include <utility>

namespace aria::pipeline {

bool isDynamicClass(int class_id) {                        // src/main.cpp:29-40
    switch (class_id) {
        case 0: case 1: case 2: case 3: case 5: case 6: case 7: case 14: case 15: case 16: return true;
        default: return false;
    }
}

namespace {
bool inDynamicObject(const core::KeyPoint& kp, const std::vector<core::Detection>& dets) {     // main.cpp:43-50
    for (const core::Detection& d : dets)
        if (isDynamicClass(d.class_id) && d.contains(kp.x, kp.y)) return true;
    return false;
}
}  // namespace

FrontEnd::FrontEnd(interfaces::FeatureExtractorPtr extractor, interfaces::MatcherPtr matcher, const FrontEndConfig& cfg)
    : extractor_(std::move(extractor)), matcher_(std::move(matcher)), cfg_(cfg) {
    if (!extractor_ || !matcher_) throw std::invalid_argument("FrontEnd: extractor and matcher must not be null");
}

FrontEnd::FrontEnd(interfaces::FeatureExtractorPtr extractor, interfaces::MatcherPtr matcher,
                   interfaces::LoopDetectorPtr loop_detector, const FrontEndConfig& cfg)
    : FrontEnd(std::move(extractor), std::move(matcher), cfg) {
    loop_detector_ = std::move(loop_detector);
}

const FrontEndResult& FrontEnd::processFrame(const std::uint8_t* image_data, int width, int height, double timestamp) {
    std::unique_ptr<core::Frame> f = std::make_unique<core::Frame>();
    f->id = next_id_++;
    f->timestamp = timestamp;
    extractor_->extract(image_data, width, height, *f);            // H12_CLEAN_ARCHITECTURE.md:597

    prev_ = std::move(cur_);
    cur_ = std::move(f);
    result_.frame = cur_.get();
    result_.previous = prev_.get();
    result_.matches.clear();
    if (prev_) {
        if (cfg_.legacy_order) matcher_->match(*prev_, *cur_, result_.matches, cfg_.ratio_threshold);   // euroc_eval.cpp:168-169
        else matcher_->match(*cur_, *prev_, result_.matches, cfg_.ratio_threshold);                      // H12...:601
    }
    result_.filtered_count = 0;
    if (cfg_.filter_dynamic_objects && prev_ && !detections_.empty()) {        // main.cpp:164-175
        const core::Frame& qf = cfg_.legacy_order ? *prev_ : *cur_;
        const core::Frame& tf = cfg_.legacy_order ? *cur_ : *prev_;
        std::size_t kept = 0;
        for (const core::Match& m : result_.matches) {
            if (inDynamicObject(qf.keypoints[(std::size_t)m.query_idx], detections_) ||
                inDynamicObject(tf.keypoints[(std::size_t)m.train_idx], detections_))
                result_.filtered_count++;
            else
                result_.matches[kept++] = m;
        }
        result_.matches.resize(kept);
    }
    detections_.clear();
    result_.is_keyframe = false;
    result_.loop.reset();
    if (loop_detector_ && prev_ && (int)result_.matches.size() >= cfg_.keyframe_min_matches) {
        core::KeyFrame kf;                                        // euroc_eval.cpp:230-236
        kf.id = cur_->id;
        kf.timestamp = cur_->timestamp;
        kf.frame = *cur_;
        result_.loop = loop_detector_->detect(kf);                // :238-239
        loop_detector_->addKeyFrame(kf);                          // :247
        result_.is_keyframe = true;
    }
    if (callback_) callback_(result_);
    return result_;
}

}  // namespace aria::pipeline
