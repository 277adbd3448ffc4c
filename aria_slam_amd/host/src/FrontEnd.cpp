// See aria_hip/FrontEnd.hpp.
#include "aria_hip/FrontEnd.hpp"

#include <stdexcept>
#include <utility>

namespace aria::pipeline {

FrontEnd::FrontEnd(interfaces::FeatureExtractorPtr extractor, interfaces::MatcherPtr matcher, const FrontEndConfig& cfg)
    : extractor_(std::move(extractor)), matcher_(std::move(matcher)), cfg_(cfg) {
    if (!extractor_ || !matcher_) throw std::invalid_argument("FrontEnd: extractor and matcher must not be null");
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
    if (callback_) callback_(result_);
    return result_;
}

}  // namespace aria::pipeline
