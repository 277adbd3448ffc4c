// euroc_frontend <dataset_path> [max_features=2000] [--legacy-order] [--csv out.csv] [--loop]
//
// The feature front-end of the reference's only end-to-end harness, src/euroc_eval.cpp:128-176, driven through the
// ports instead of cv::cuda::ORB / cv::cuda::DescriptorMatcher: for every image of an ASL/EuRoC sequence
//   extract ORB (2000 features by default, euroc_eval.cpp:88) -> kNN-2 + ratio 0.75 against the previous frame
// (:168-175) -> report; with --loop also the loop-closure candidate step of :103, 230-247 (every frame with >= 8 matches
// becomes a keyframe: detect against the HBM-resident database with LoopClosureDetector(200, 0.4, 50)'s parameters, then
// add). Pose estimation, EKF, YOLO, geometric verification and mapping are out of scope.
// Prints the progress line every 100 frames like the reference (:271-277) and a summary; --csv writes
// "frame,timestamp,keypoints,matches,hash,keyframe,loop_match_id,loop_score" per frame, hash = FNV-1a 64 over the frame's
// keypoint records, descriptor rows and match records (what the parity test compares with the oracle's).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <memory>
#include <string>
#include <vector>

#include "aria_hip/AslSequence.hpp"
#include "aria_hip/FrontEnd.hpp"
#include "aria_hip/HipFactory.hpp"

using namespace aria;

static std::uint64_t fnv1a(const void* p, std::size_t n, std::uint64_t h) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (std::size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "Usage: %s <dataset_path> [max_features] [--legacy-order] [--csv file]\n", argv[0]);
        return -1;                                                        // euroc_eval.cpp:64-70
    }
    int max_features = 2000;
    bool legacy = false, loop = false;
    std::string csv;
    for (int i = 2; i < argc; i++) {
        if (!std::strcmp(argv[i], "--legacy-order")) legacy = true;
        else if (!std::strcmp(argv[i], "--loop")) loop = true;
        else if (!std::strcmp(argv[i], "--csv") && i + 1 < argc) csv = argv[++i];
        else max_features = std::atoi(argv[i]);
    }
    io::AslSequence seq;
    if (!seq.load(argv[1])) {
        std::fprintf(stderr, "Failed to load dataset from %s\n", argv[1]);
        return -1;                                                        // euroc_eval.cpp:75-78
    }
    std::printf("Loaded: %zu images\n", seq.size());

    factory::HipFactoryConfig fc;                                        // PipelineFactory's HIP mode (aria_hip/HipFactory.hpp)
    fc.max_features = max_features;
    fc.frontend.legacy_order = legacy;
    fc.enable_loop_closure = loop;                                       // LoopClosureDetector(200, 0.4, 50), euroc_eval.cpp:103
    std::unique_ptr<pipeline::FrontEnd> fep = factory::createHip(fc);
    pipeline::FrontEnd& fe = *fep;
    std::ofstream out;
    if (!csv.empty()) { out.open(csv); out << std::setprecision(17); out << "frame,timestamp,keypoints,matches,hash,keyframe,loop_match_id,loop_score\n"; }
    long long n_keyframes = 0, n_loops = 0;

    std::vector<std::uint8_t> gray;
    int w = 0, h = 0;
    long long total_kp = 0, total_matches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto t_last = t0;
    for (std::size_t i = 0; i < seq.size(); i++) {
        seq.read(i, gray, w, h);
        const pipeline::FrontEndResult& r = fe.processFrame(gray.data(), w, h, seq.at(i).timestamp);
        total_kp += (long long)r.frame->numKeypoints();
        total_matches += (long long)r.matches.size();
        n_keyframes += r.is_keyframe ? 1 : 0;
        n_loops += r.loop ? 1 : 0;
        if (out.is_open()) {
            std::uint64_t hsh = 14695981039346656037ull;
            hsh = fnv1a(r.frame->keypoints.data(), r.frame->keypoints.size() * sizeof(core::KeyPoint), hsh);
            hsh = fnv1a(r.frame->descriptors.data(), r.frame->numKeypoints() * 32, hsh);
            hsh = fnv1a(r.matches.data(), r.matches.size() * sizeof(core::Match), hsh);
            out << i << ',' << std::to_string(seq.at(i).timestamp) << ',' << r.frame->numKeypoints() << ',' << r.matches.size() << ','
                << hsh << ',' << (r.is_keyframe ? 1 : 0) << ',' << (r.loop ? (long long)r.loop->match_id : -1) << ','
                << (r.loop ? r.loop->score : 0.0) << '\n';
        }
        if ((i + 1) % 100 == 0) {                                          // euroc_eval.cpp:271-277
            const auto now = std::chrono::steady_clock::now();
            const double fps = 100.0 / std::chrono::duration<double>(now - t_last).count();
            t_last = now;
            std::printf("Frame %zu/%zu | FPS: %.1f | keypoints: %zu | matches: %zu\n", i + 1, seq.size(), fps,
                        r.frame->numKeypoints(), r.matches.size());
        }
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("frames %zu size %dx%d mean_keypoints %.2f mean_matches %.2f fps %.1f (PNG decode + H2D + extract + match + D2H)\n",
                seq.size(), w, h, seq.size() ? (double)total_kp / seq.size() : 0.0,
                seq.size() > 1 ? (double)total_matches / (seq.size() - 1) : 0.0, seq.size() / secs);
    if (loop) std::printf("keyframes %lld loops %lld\n", n_keyframes, n_loops);
    return 0;
}
