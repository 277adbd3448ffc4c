// Exercises the C++ adapters the way SlamPipeline::processFrame is designed to (reference
// docs/milestones/H12_CLEAN_ARCHITECTURE.md:595-605): extractor_->extract(image, w, h, frame);
// matcher_->match(frame, *prev_frame_, matches). Prints FNV-1a hashes that tests/test_cpp_adapters.py compares
// with the python binding's results (which the GPU parity tests compare with the oracle).
//   adapter_selftest nogpu   -> expects a std::runtime_error from the first extract (no CPU fallback)
//   adapter_selftest gpu     -> runs the full sequence on device 0
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "aria_hip/HipMatcher.hpp"
#include "aria_hip/OrbHipExtractor.hpp"
#include "aria_orb_hip.h"

using namespace aria;

static unsigned long long fnv(const void* p, size_t n, unsigned long long h = 1469598103934665603ull) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "gpu";
    const int W = 640, H = 480;
    std::vector<uint8_t> a((size_t)W * H), b((size_t)W * H);
    aria_synth_frame_pair(1, W, H, a.data(), b.data());

    // the ports, held the way PipelineFactory would hand them to SlamPipeline (unique_ptr to the interface)
    interfaces::FeatureExtractorPtr extractor = std::make_unique<adapters::hip::OrbHipExtractor>(2000);
    interfaces::MatcherPtr matcher = std::make_unique<adapters::hip::HipMatcher>();

    if (mode == "nogpu") {
        try {
            core::Frame f;
            extractor->extract(a.data(), W, H, f);
        } catch (const std::runtime_error& e) {
            std::printf("OK nogpu: %s\n", e.what());
            return 0;
        }
        std::printf("FAIL: extract succeeded without a GPU\n");
        return 1;
    }

    core::Frame fa, fb, fb_async;
    extractor->extract(a.data(), W, H, fa);
    extractor->extract(b.data(), W, H, fb);
    extractor->sync();   // nothing pending: must be a no-op
    extractor->extractAsync(b.data(), W, H, fb_async);
    extractor->sync();
    std::printf("n_a %zu n_b %zu\n", fa.numKeypoints(), fb.numKeypoints());
    std::printf("kp_a %016llx desc_a %016llx\n", fnv(fa.keypoints.data(), fa.keypoints.size() * sizeof(core::KeyPoint)),
                fnv(fa.descriptors.data(), fa.descriptors.size()));
    std::printf("kp_b %016llx desc_b %016llx\n", fnv(fb.keypoints.data(), fb.keypoints.size() * sizeof(core::KeyPoint)),
                fnv(fb.descriptors.data(), fb.descriptors.size()));
    const bool async_same = fb.keypoints.size() == fb_async.keypoints.size() && fb.descriptors == fb_async.descriptors &&
                            std::memcmp(fb.keypoints.data(), fb_async.keypoints.data(), fb.keypoints.size() * sizeof(core::KeyPoint)) == 0 &&
                            fb_async.width == W && fb_async.height == H;
    std::printf("async_same %d\n", (int)async_same);

    std::vector<core::Match> matches;
    matches.push_back({-7, -7, -7.f});                       // match() must append, not clear
    matcher->match(fb, fa, matches);                         // query = current, train = previous
    std::printf("n_matches %zu first_kept %d\n", matches.size() - 1, matches[0].query_idx == -7);
    std::printf("matches %016llx\n", fnv(matches.data() + 1, (matches.size() - 1) * sizeof(core::Match)));

    core::Frame empty;
    size_t before = matches.size();
    matcher->match(empty, fa, matches);
    matcher->match(fb, empty, matches);
    std::printf("empty_untouched %d\n", (int)(matches.size() == before));

    std::vector<core::Frame> cands = {fa, fb, empty};
    std::vector<std::vector<core::Match>> all(7);
    matcher->matchMultiple(fb, cands, all, 0.75f);
    std::printf("multi %zu %zu %zu %zu\n", all.size(), all[0].size(), all[1].size(), all[2].size());

    extractor->setMaxFeatures(500);
    core::Frame f500;
    extractor->extract(a.data(), W, H, f500);
    std::printf("n_500 %zu max %d\n", f500.numKeypoints(), extractor->getMaxFeatures());
    std::printf("kp_500 %016llx desc_500 %016llx\n", fnv(f500.keypoints.data(), f500.keypoints.size() * sizeof(core::KeyPoint)),
                fnv(f500.descriptors.data(), f500.descriptors.size()));

    auto* hm = static_cast<adapters::hip::HipMatcher*>(matcher.get());
    fa.id = 0; fb.id = 100;
    std::vector<core::Frame> kfs = {fa};
    auto lc = hm->findLoopCandidates(fb, kfs, 30);
    std::printf("loop %zu %.6f\n", lc.size(), lc.empty() ? 0.0 : lc[0].second);
    std::printf("DONE\n");
    return 0;
}
