// aria::factory -- the HIP execution mode of the reference's PipelineFactory (include/factory/PipelineFactory.hpp:9-47):
// FactoryConfig{mode, cuda_device, max_features} selects which adapters are injected into SlamPipeline's constructor
// (include/pipeline/SlamPipeline.hpp:32-40). createHipComponents() builds the three front-end components the HIP mode
// injects -- extractor, matcher, loop detector -- and createHip() wires them into the FrontEnd (the part of SlamPipeline
// this repository implements). INTEGRATION.md shows the ExecutionMode::HIP branch a maintainer adds to
// PipelineFactory::create with exactly these calls.
#pragma once
#include <memory>

#include "aria_hip/FrontEnd.hpp"
#include "aria_hip/compat.hpp"

namespace aria::factory {

struct HipFactoryConfig {
    int hip_device = 0;            // FactoryConfig::cuda_device (PipelineFactory.hpp:24)
    int max_features = 1000;       // FactoryConfig::max_features (:27)
    void* stream = nullptr;        // borrowed hipStream_t shared by the three components, or nullptr: createHip() then makes
                                   // one stream for extractor + matcher (and keeps it alive as long as the FrontEnd)
    pipeline::FrontEndConfig frontend;
    bool enable_loop_closure = true;        // PipelineConfig::enable_loop_closure (SlamPipeline.hpp:17)
    int loop_min_frames_between = 200;      // src/euroc_eval.cpp:103
    double loop_min_score = 0.4;
    int loop_min_matches = 50;
};

struct HipComponents {
    interfaces::FeatureExtractorPtr extractor;
    interfaces::MatcherPtr matcher;
    interfaces::LoopDetectorPtr loop_detector;      // null when loop closure is disabled
    std::shared_ptr<void> shared_stream;            // owner of the stream made for the components (empty: caller's stream or
                                                    // none); must be released after them
};

HipComponents createHipComponents(const HipFactoryConfig& cfg);
std::unique_ptr<pipeline::FrontEnd> createHip(const HipFactoryConfig& cfg = {});

}  // namespace aria::factory
