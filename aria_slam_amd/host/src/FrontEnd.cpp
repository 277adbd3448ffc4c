// See aria_hip/FrontEnd.hpp.
#include "aria_hip/FrontEnd.hpp"

#include <algorithm>
#include <stdexcept>
#include <utility>

#include "aria_hip/HipMatcher.hpp"
#include "aria_hip/OrbHipExtractor.hpp"

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
    auto* hx = dynamic_cast<adapters::hip::OrbHipExtractor*>(extractor_.get());
    auto* hm = dynamic_cast<adapters::hip::HipMatcher*>(matcher_.get());
    if (cfg_.device_handoff && hx && hm && hx->device() == hm->device())
        handoff_mode_ = (hx->streamArg() && hx->streamArg() == hm->streamArg()) ? 2 : 1;
}

// extract + match of one frame. Host port calls in general; with the HIP pair the descriptors are handed over on the
// device (see FrontEndConfig::device_handoff) -- the matcher keeps the previous frame's set resident.
void FrontEnd::extractAndMatch(const std::uint8_t* image_data, int width, int height, core::Frame& f) {
    const bool have_prev = cur_ != nullptr;             // cur_ still is the previous frame here
    if (handoff_mode_ == 0) {
        extractor_->extract(image_data, width, height, f);            // H12_CLEAN_ARCHITECTURE.md:597
        if (have_prev) {
            if (cfg_.legacy_order) matcher_->match(*cur_, f, result_.matches, cfg_.ratio_threshold);   // euroc_eval.cpp:168-169
            else matcher_->match(f, *cur_, result_.matches, cfg_.ratio_threshold);                      // H12...:601
        }
        return;
    }
    auto& hx = static_cast<adapters::hip::OrbHipExtractor&>(*extractor_);
    auto& hm = static_cast<adapters::hip::HipMatcher&>(*matcher_);
    const int n_prev = have_prev ? (int)cur_->numKeypoints() : 0;
    bool queued = false;
    adapters::hip::OrbHipExtractor::DeviceResult before;
    if (handoff_mode_ == 2) {
        hx.reserve(width, height);
        hx.extractAsync(image_data, width, height, f);
        before = hx.deviceResult();
        // CudaMatcher.cpp:35-37: nothing to match against an empty previous frame
        if (have_prev && n_prev > 0 && hm.residentRows() == n_prev)
            queued = hm.matchDeviceAsync(before.descriptors, before.count, before.rows, !cfg_.legacy_order, cfg_.ratio_threshold);
        hx.sync();
    } else {
        hx.extract(image_data, width, height, f);
    }
    const adapters::hip::OrbHipExtractor::DeviceResult dr = hx.deviceResult();
    const int n = (int)f.numKeypoints();
    if (queued) {
        std::vector<core::Match> got;
        hm.finishDevice(n, got);
        // a tie storm made the extractor grow (and move) its result block during sync(): what was queued read a partial
        // set -- match again from the block as it is now
        if (dr.descriptors == before.descriptors && n <= before.rows) {
            result_.matches.insert(result_.matches.end(), got.begin(), got.end());
            return;
        }
        hm.retainDevice(nullptr, 0);
    }
    // (a frame larger than the matcher handle was created for -- a tie storm, or setMaxFeatures raised mid-sequence -- makes
    // the handle grow, which drops the resident set: such a frame takes the host port call below and becomes resident after)
    if (have_prev && n > 0 && n_prev > 0 && hm.residentRows() == n_prev && hm.fits(std::max(n, n_prev), std::max(n, n_prev))) {
        if (cfg_.legacy_order) hm.matchDevice(nullptr, n_prev, dr.descriptors, n, result_.matches, cfg_.ratio_threshold);
        else hm.matchDevice(dr.descriptors, n, nullptr, n_prev, result_.matches, cfg_.ratio_threshold);
    } else if (have_prev && n > 0 && n_prev > 0) {
        // the resident set is not the previous frame's (another caller used the matcher): ordinary port call
        if (cfg_.legacy_order) matcher_->match(*cur_, f, result_.matches, cfg_.ratio_threshold);
        else matcher_->match(f, *cur_, result_.matches, cfg_.ratio_threshold);
        hm.retainDevice(dr.descriptors, n);
    } else {
        hm.reserve(dr.rows, dr.rows);                    // (before anything is resident: growing the handle drops it)
        hm.retainDevice(dr.descriptors, n);              // first frame, or an empty one: only becomes the resident set
    }
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
    result_.matches.clear();
    extractAndMatch(image_data, width, height, *f);

    prev_ = std::move(cur_);
    cur_ = std::move(f);
    result_.frame = cur_.get();
    result_.previous = prev_.get();
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
