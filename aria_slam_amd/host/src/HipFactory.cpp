// See aria_hip/HipFactory.hpp.
#include "aria_hip/HipFactory.hpp"

#include <utility>

#include "aria_hip/HipLoopDetector.hpp"
#include "aria_hip/HipMatcher.hpp"
#include "aria_hip/OrbHipExtractor.hpp"

namespace aria::factory {

HipComponents createHipComponents(const HipFactoryConfig& cfg) {
    HipComponents c;
    c.extractor = std::make_unique<adapters::hip::OrbHipExtractor>(cfg.max_features, cfg.stream, cfg.hip_device);
    c.matcher = std::make_unique<adapters::hip::HipMatcher>(cfg.stream, cfg.hip_device);
    if (cfg.enable_loop_closure)
        // slot rows: nfeatures + 64 rows of tie slack per level (aria_orb_kp_capacity), rounded up
        c.loop_detector = std::make_unique<adapters::hip::HipLoopDetector>(
            cfg.loop_min_frames_between, cfg.loop_min_score, cfg.loop_min_matches, ((cfg.max_features + 8 * 64 + 63) / 64) * 64, 500,
            cfg.stream, cfg.hip_device);
    return c;
}

std::unique_ptr<pipeline::FrontEnd> createHip(const HipFactoryConfig& cfg) {
    HipComponents c = createHipComponents(cfg);
    return std::make_unique<pipeline::FrontEnd>(std::move(c.extractor), std::move(c.matcher), std::move(c.loop_detector), cfg.frontend);
}

}  // namespace aria::factory
