// euroc_frontend <dataset_path> [max_features=2000] [--legacy-order] [--csv out.csv]
//
// The feature front-end of the reference's only end-to-end harness, src/euroc_eval.cpp:128-176, driven through the
// ports instead of cv::cuda::ORB / cv::cuda::DescriptorMatcher: for every image of an ASL/EuRoC sequence
//   extract ORB (2000 features by default, euroc_eval.cpp:88) -> kNN-2 + ratio 0.75 against the previous frame
// (:168-175) -> report. Pose estimation, EKF, YOLO, loop closure and mapping (:179-245) are out of scope.
// Prints the progress line every 100 frames like the reference (:271-277) and a summary; --csv writes
// "frame,timestamp,keypoints,matches" per frame.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "aria_hip/AslSequence.hpp"
#include "aria_hip/FrontEnd.hpp"
#include "aria_hip/HipMatcher.hpp"
#include "aria_hip/OrbHipExtractor.hpp"

using namespace aria;

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "Usage: %s <dataset_path> [max_features] [--legacy-order] [--csv file]\n", argv[0]);
        return -1;                                                        // euroc_eval.cpp:64-70
    }
    int max_features = 2000;
    bool legacy = false;
    std::string csv;
    for (int i = 2; i < argc; i++) {
        if (!std::strcmp(argv[i], "--legacy-order")) legacy = true;
        else if (!std::strcmp(argv[i], "--csv") && i + 1 < argc) csv = argv[++i];
        else max_features = std::atoi(argv[i]);
    }
    io::AslSequence seq;
    if (!seq.load(argv[1])) {
        std::fprintf(stderr, "Failed to load dataset from %s\n", argv[1]);
        return -1;                                                        // euroc_eval.cpp:75-78
    }
    std::printf("Loaded: %zu images\n", seq.size());

    pipeline::FrontEndConfig cfg;
    cfg.legacy_order = legacy;
    pipeline::FrontEnd fe(std::make_unique<adapters::hip::OrbHipExtractor>(max_features),
                          std::make_unique<adapters::hip::HipMatcher>(), cfg);
    std::ofstream out;
    if (!csv.empty()) { out.open(csv); out << "frame,timestamp,keypoints,matches\n"; }

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
        if (out.is_open())
            out << i << ',' << std::to_string(seq.at(i).timestamp) << ',' << r.frame->numKeypoints() << ',' << r.matches.size() << '\n';
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
    return 0;
}
