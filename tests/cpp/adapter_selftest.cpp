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

#include "aria_hip/FrontEnd.hpp"
#include "aria_hip/HipFactory.hpp"
#include "aria_hip/HipLoopDetector.hpp"
#include "aria_hip/HipMatcher.hpp"
#include "aria_hip/OrbHipExtractor.hpp"
#include "aria_hip/Shard.hpp"
#include "aria_orb_hip.h"

using namespace aria;

static unsigned long long fnv(const void* p, size_t n, unsigned long long h = 1469598103934665603ull) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

// Mocks in the spirit of the reference's planned MockExtractor / MockMatcher
// (docs/milestones/H12_CLEAN_ARCHITECTURE.md:629-645, 1795-1815): no GPU, deterministic outputs.
struct MockExtractor : interfaces::IFeatureExtractor {
    int max_features = 3, calls = 0;
    void extract(const uint8_t* image, int w, int h, core::Frame& f) override {
        calls++;
        f.width = w; f.height = h;
        f.keypoints.assign((size_t)max_features, core::KeyPoint{(float)image[0], 2.f, 31.f, 0.f, 1.f, 0});
        f.descriptors.assign((size_t)max_features * 32, image[0]);
    }
    void setMaxFeatures(int n) override { max_features = n; }
    int getMaxFeatures() const override { return max_features; }
};
struct MockMatcher : interfaces::IMatcher {
    std::vector<std::pair<unsigned long long, unsigned long long>> calls;   // (query.id, train.id)
    float last_ratio = -1.f;
    void match(const core::Frame& q, const core::Frame& t, std::vector<core::Match>& m, float ratio) override {
        calls.push_back({q.id, t.id});
        last_ratio = ratio;
        m.push_back({0, 0, (float)(q.descriptors[0] ^ t.descriptors[0])});
    }
};

static int frontend_mock_test() {
    int ok = 1;
    for (int legacy = 0; legacy < 2; legacy++) {
        auto ex = std::make_unique<MockExtractor>();
        auto mt = std::make_unique<MockMatcher>();
        MockMatcher* mtp = mt.get();
        pipeline::FrontEndConfig cfg;
        cfg.legacy_order = legacy != 0;
        cfg.ratio_threshold = 0.7f;
        pipeline::FrontEnd fe(std::move(ex), std::move(mt), cfg);
        int cb = 0;
        fe.setCallback([&](const pipeline::FrontEndResult& r) { cb++; ok &= (r.frame != nullptr); });
        std::vector<uint8_t> img(640 * 480, 128);                       // the prose test's constant-128 image (H12...:1803)
        const auto& r0 = fe.processFrame(img.data(), 640, 480, 0.0);
        ok &= r0.matches.empty() && r0.previous == nullptr && r0.frame->id == 0 && r0.frame->numKeypoints() == 3;
        img[0] = 130;
        const auto& r1 = fe.processFrame(img.data(), 640, 480, 0.05);
        ok &= r1.matches.size() == 1 && r1.previous && r1.previous->id == 0 && r1.frame->id == 1 && r1.frame->timestamp == 0.05;
        ok &= mtp->calls.size() == 1 && mtp->last_ratio == 0.7f;
        if (legacy) ok &= mtp->calls[0].first == 0 && mtp->calls[0].second == 1;       // prev -> current (euroc_eval.cpp:168)
        else ok &= mtp->calls[0].first == 1 && mtp->calls[0].second == 0;              // current -> prev (H12...:601)
        ok &= r1.matches[0].distance == (float)(130 ^ 128);
        fe.processFrame(img.data(), 640, 480, 0.1);
        ok &= cb == 3 && fe.framesProcessed() == 3 && mtp->calls.size() == 2;
        // dynamic-object filter (main.cpp:42-50, 164-175): the mock keypoints sit at (image[0], 2); a "person" box around
        // them drops the match, a "chair" (class 56) box does not, and detections are consumed by one frame
        fe.setDetections({core::Detection{100.f, 0.f, 200.f, 10.f, 0.9f, 56}});
        ok &= fe.processFrame(img.data(), 640, 480, 0.15).matches.size() == 1;
        fe.setDetections({core::Detection{100.f, 0.f, 200.f, 10.f, 0.9f, 0}});
        const auto& r4 = fe.processFrame(img.data(), 640, 480, 0.2);
        ok &= r4.matches.empty() && r4.filtered_count == 1;
        ok &= fe.processFrame(img.data(), 640, 480, 0.25).matches.size() == 1;
        ok &= pipeline::isDynamicClass(16) && !pipeline::isDynamicClass(4);
    }
    std::printf("%s frontend_mock\n", ok ? "OK" : "FAIL");
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "gpu";
    if (mode == "frontend_mock") return frontend_mock_test();
    if (mode == "shard") {       // the partition rule of euroc_frontend --shards, for tests/test_cpp_adapters.py (no GPU)
        const std::size_t ns[] = {0, 1, 2, 7, 100, 101, 4096, 32768};
        for (std::size_t n : ns)
            for (int g = 1; g <= 8; g++)
                for (int r = 0; r < g; r++) {
                    const pipeline::ShardPlan p = pipeline::shardPlan(n, r, g);
                    std::printf("%zu %d %d %zu %zu %zu\n", n, g, r, p.lo, p.hi, p.first);
                }
        return 0;
    }
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
    // the batched matchMultiple must give, per candidate, exactly what match() gives
    std::printf("multi_hash %016llx %016llx\n", fnv(all[0].data(), all[0].size() * sizeof(core::Match)),
                fnv(all[1].data(), all[1].size() * sizeof(core::Match)));

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
    // ILoopDetector over the HBM-resident database: capacity 3 -> the fourth keyframe drops the oldest (LoopClosure.cpp:28-30)
    {
        adapters::hip::HipLoopDetector ld(1, 0.3, 30, 4096, 3);
        core::KeyFrame k0, k1, k2, k3, q;
        k0.id = 0; k0.frame = fa; k1.id = 1; k1.frame = fb; k2.id = 2; k2.frame = f500; k3.id = 3; k3.frame = fa;
        ld.addKeyFrame(k0); ld.addKeyFrame(k1); ld.addKeyFrame(k2); ld.addKeyFrame(k3);
        q.id = 50; q.frame = fb;
        auto cands = ld.findCandidates(q);
        std::printf("ld_size %d oldest %llu ncand %zu\n", ld.size(), (unsigned long long)ld.keyframeId(0), cands.size());
        for (auto& c : cands) std::printf("ld_cand %llu %.17g\n", (unsigned long long)ld.keyframeId(c.first), c.second);
        auto loop = ld.detect(q);
        std::printf("ld_loop %d %llu %zu %d\n", (int)loop.has_value(), loop ? (unsigned long long)loop->match_id : 0ull,
                    loop ? loop->matches.size() : (size_t)0, ld.getLoopCount());
        q.id = 3;                                            // nothing is min_frames_between older than this query ... except id <= 2
        ld.setMinFramesBetween(10);
        std::printf("ld_recent %d\n", (int)ld.detect(q).has_value());
    }
    // PipelineFactory's HIP mode: the three components, wired into the front end
    {
        factory::HipFactoryConfig fc;
        fc.max_features = 2000;
        fc.loop_min_frames_between = 1;
        auto fe = factory::createHip(fc);
        const auto& r0 = fe->processFrame(a.data(), W, H, 0.0);
        const size_t n0 = r0.frame->numKeypoints();
        const auto& r1 = fe->processFrame(b.data(), W, H, 0.05);
        std::printf("factory %zu %zu %zu %d %d\n", n0, r1.frame->numKeypoints(), r1.matches.size(), (int)r1.is_keyframe,
                    fe->loopDetector() != nullptr);
    }
    // getGpuDescriptors() / matchGpu hand-off (OrbCudaExtractor.hpp:34-35, CudaMatcher.hpp:22-28): same matches as the host
    // port calls, in both query/train orders, with the previous frame's set resident in the matcher
    {
        adapters::hip::OrbHipExtractor ex(2000);
        adapters::hip::HipMatcher mt;
        core::Frame ga, gb;
        ex.extract(a.data(), W, H, ga);
        auto da = ex.deviceResult();
        mt.retainDevice(da.descriptors, (int)ga.numKeypoints());
        ex.extract(b.data(), W, H, gb);
        auto db = ex.deviceResult();
        std::vector<core::Match> m_cur_prev, m_prev_cur, m_both;
        mt.matchDevice(db.descriptors, (int)gb.numKeypoints(), nullptr, (int)ga.numKeypoints(), m_cur_prev);     // b vs resident a
        std::printf("dev_n %d %d rows %d resident %d\n", da.n, db.n, db.rows, mt.residentRows());
        // resident is b now; legacy order for the next frame (a again): query = previous (b), train = current (a)
        ex.extract(a.data(), W, H, ga);
        auto da2 = ex.deviceResult();
        mt.matchDevice(nullptr, (int)gb.numKeypoints(), da2.descriptors, (int)ga.numKeypoints(), m_prev_cur);     // resident b vs a
        mt.matchDevice(da2.descriptors, (int)ga.numKeypoints(), da2.descriptors, (int)ga.numKeypoints(), m_both); // a vs a, both explicit
        std::printf("dev_match %016llx %016llx %016llx\n", fnv(m_cur_prev.data(), m_cur_prev.size() * sizeof(core::Match)),
                    fnv(m_prev_cur.data(), m_prev_cur.size() * sizeof(core::Match)), fnv(m_both.data(), m_both.size() * sizeof(core::Match)));
        std::printf("dev_ptr_stable %d\n", (int)(da.descriptors == db.descriptors && db.descriptors == da2.descriptors && da.descriptors != nullptr));
    }
    // FrontEnd over the same four frames in its three hand-off modes and both orders: identical matches
    for (int legacy = 0; legacy < 2; legacy++) {
        for (int mode = 0; mode < 3; mode++) {
            factory::HipFactoryConfig fc;
            fc.max_features = 2000;
            fc.enable_loop_closure = false;
            fc.frontend.legacy_order = legacy != 0;
            fc.frontend.device_handoff = mode != 0;
            std::unique_ptr<pipeline::FrontEnd> fe;
            if (mode == 2) fe = factory::createHip(fc);                 // one stream for both components
            else fe = std::make_unique<pipeline::FrontEnd>(std::make_unique<adapters::hip::OrbHipExtractor>(2000),
                                                           std::make_unique<adapters::hip::HipMatcher>(), fc.frontend);
            std::printf("fe_%d_%d mode %d", legacy, mode, fe->handoffMode());
            const uint8_t* seq[4] = {a.data(), b.data(), a.data(), b.data()};
            for (int i = 0; i < 4; i++) {
                const auto& r = fe->processFrame(seq[i], W, H, 0.05 * i);
                std::printf(" %zu:%016llx", r.matches.size(), fnv(r.matches.data(), r.matches.size() * sizeof(core::Match)));
            }
            std::printf("\n");
        }
    }
    // setMaxFeatures raised above the matcher handle's capacity (4096 rows) in the middle of a sequence. The
    // larger frame makes the matcher grow, which drops its resident set: FrontEnd must take the host port call for that
    // frame instead of throwing, and be back on the device hand-off afterwards. All three modes give the same matches.
    for (int mode = 0; mode < 3; mode++) {
        factory::HipFactoryConfig fc;
        fc.max_features = 2000;
        fc.enable_loop_closure = false;
        fc.frontend.device_handoff = mode != 0;
        std::unique_ptr<pipeline::FrontEnd> fe;
        if (mode == 2) fe = factory::createHip(fc);
        else fe = std::make_unique<pipeline::FrontEnd>(std::make_unique<adapters::hip::OrbHipExtractor>(2000),
                                                       std::make_unique<adapters::hip::HipMatcher>(), fc.frontend);
        std::printf("fe_grow_%d", mode);
        const uint8_t* seq[4] = {a.data(), b.data(), a.data(), b.data()};
        for (int i = 0; i < 4; i++) {
            if (i == 1) fe->extractor().setMaxFeatures(6000);
            const auto& r = fe->processFrame(seq[i], W, H, 0.05 * i);
            std::printf(" %zu:%zu:%016llx", r.frame->numKeypoints(), r.matches.size(), fnv(r.matches.data(), r.matches.size() * sizeof(core::Match)));
        }
        std::printf("\n");
    }
    std::printf("DONE\n");
    return 0;
}
