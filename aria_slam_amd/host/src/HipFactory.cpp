// See aria_hip/HipFactory.hpp.
#include "aria_hip/HipFactory.hpp"

#include <utility>

#include "aria_hip/HipLoopDetector.hpp"
#include "aria_hip/HipMatcher.hpp"
#include "aria_hip/OrbHipExtractor.hpp"
#include "aria_orb_hip.h"

namespace aria::factory {

HipComponents createHipComponents(const HipFactoryConfig& cfg) {
    HipComponents c;
    void* stream = cfg.stream;
    if (!stream) {
        // one stream for extractor and matcher: the matcher's work can then be queued behind the extraction
        // (FrontEndConfig::device_handoff). If the device is not usable the components report that themselves.
        const int dev = cfg.hip_device;
        if (aria_stream_create(dev, &stream) == ARIA_OK && stream)
            c.shared_stream = std::shared_ptr<void>(stream, [dev](void* s) { aria_stream_destroy(dev, s); });
        else
            stream = nullptr;
    }
    c.extractor = std::make_unique<adapters::hip::OrbHipExtractor>(cfg.max_features, stream, cfg.hip_device);
    c.matcher = std::make_unique<adapters::hip::HipMatcher>(stream, cfg.hip_device);
    if (cfg.enable_loop_closure)
        // slot rows: nfeatures + 64 rows of tie slack per level (aria_orb_kp_capacity), rounded up
        c.loop_detector = std::make_unique<adapters::hip::HipLoopDetector>(
            cfg.loop_min_frames_between, cfg.loop_min_score, cfg.loop_min_matches, ((cfg.max_features + 8 * 64 + 63) / 64) * 64, 500,
            cfg.stream, cfg.hip_device);
    return c;
}

std::unique_ptr<pipeline::FrontEnd> createHip(const HipFactoryConfig& cfg) {
    HipComponents c = createHipComponents(cfg);
    auto fe = std::make_unique<pipeline::FrontEnd>(std::move(c.extractor), std::move(c.matcher), std::move(c.loop_detector), cfg.frontend);
    fe->setKeepAlive(std::move(c.shared_stream));
    return fe;
}

}  // namespace aria::factory
