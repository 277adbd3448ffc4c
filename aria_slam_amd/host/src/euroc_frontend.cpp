// euroc_frontend <dataset_path> [max_features=2000] [--legacy-order] [--csv out.csv] [--loop] [--devices N] [--shards K]
//                [--batch B] [--decode-threads T]
//
// The feature front-end of the reference's only end-to-end harness, src/euroc_eval.cpp:128-176, driven through the
// ports instead of cv::cuda::ORB / cv::cuda::DescriptorMatcher: for every image of an ASL/EuRoC sequence
//   extract ORB (2000 features by default, euroc_eval.cpp:88) -> kNN-2 + ratio 0.75 against the previous frame
// (:168-175) -> report; with --loop also the loop-closure candidate step of :103, 230-247 (every frame with >= 8 matches
// becomes a keyframe: detect against the HBM-resident database with LoopClosureDetector(200, 0.4, 50)'s parameters, then
// add). Pose estimation, EKF, YOLO, geometric verification and mapping are out of scope.
//
// --devices N / --shards K: the sequence is cut into K contiguous shards (default K = N) with a one-frame halo
// (aria_hip/Shard.hpp); every shard gets its own host thread and its own extractor + matcher handles (PipelineFactory's
// HIP mode) on device (shard mod N), and the per-frame results are merged in frame order -- the reference's loop over a
// recorded sequence, spread over the GPUs of a node. No data crosses between shards. The loop-closure step consumes the
// merged stream in frame order on device 0 afterwards (it is a sequential scan over the sequence by nature), so --loop
// gives the same keyframes and loops for every K. K > N runs several logical shards on one device.
//
// --batch B (round 4): every shard runs the CHUNKED pipeline of aria_hip/BatchFrontEnd.hpp instead of one processFrame call
// per image -- T decode workers fill pinned buffers, the copy stream uploads chunk c + 1 while the compute stream extracts
// all B frames of chunk c in one batch pass and matches its B pairs in one launch (the kernels bench.py measures), results
// come back per chunk. Same per-frame results (hashes, keyframes, loops) as the frame-at-a-time run; the summary names
// decode, staging and kernel time apart. The loop-closure step then runs over the merged stream as in the sharded mode.
//
// Prints the progress line every 100 frames like the reference (:271-277) and a summary; --csv writes
// "frame,timestamp,keypoints,matches,hash,keyframe,loop_match_id,loop_score" per frame, hash = FNV-1a 64 over the frame's
// keypoint records, descriptor rows and match records (what the parity test compares with the oracle's).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <exception>
#include <fstream>
#include <iomanip>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "aria_hip/AslSequence.hpp"
#include "aria_hip/BatchFrontEnd.hpp"
#include "aria_hip/FrontEnd.hpp"
#include "aria_hip/HipFactory.hpp"
#include "aria_hip/HipLoopDetector.hpp"
#include "aria_hip/Shard.hpp"
#include "aria_orb_hip.h"

using namespace aria;

static std::uint64_t fnv1a(const void* p, std::size_t n, std::uint64_t h) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (std::size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

namespace {
struct FrameRecord {
    std::size_t keypoints = 0, matches = 0;
    std::uint64_t hash = 0;
    bool is_keyframe = false;
    long long loop_match_id = -1;
    double loop_score = 0.0;
    std::unique_ptr<core::Frame> frame;      // kept only for the loop-closure step of a sharded / batched run, and only if it can be a keyframe
};

std::uint64_t frame_hash(const core::Frame& f, const std::vector<core::Match>& m) {
    std::uint64_t h = 14695981039346656037ull;
    h = fnv1a(f.keypoints.data(), f.keypoints.size() * sizeof(core::KeyPoint), h);
    h = fnv1a(f.descriptors.data(), f.numKeypoints() * 32, h);
    return fnv1a(m.data(), m.size() * sizeof(core::Match), h);
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "Usage: %s <dataset_path> [max_features] [--legacy-order] [--csv file] [--loop] [--devices N] [--shards K] [--batch B] [--decode-threads T]\n", argv[0]);
        return -1;                                                        // euroc_eval.cpp:64-70
    }
    int max_features = 2000, devices = 1, shards = 0, batch = 0, decode_threads = 4;
    bool legacy = false, loop = false;
    std::string csv;
    for (int i = 2; i < argc; i++) {
        if (!std::strcmp(argv[i], "--legacy-order")) legacy = true;
        else if (!std::strcmp(argv[i], "--loop")) loop = true;
        else if (!std::strcmp(argv[i], "--csv") && i + 1 < argc) csv = argv[++i];
        else if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) devices = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--shards") && i + 1 < argc) shards = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--batch") && i + 1 < argc) batch = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--decode-threads") && i + 1 < argc) decode_threads = std::atoi(argv[++i]);
        else max_features = std::atoi(argv[i]);
    }
    if (devices < 1) devices = 1;
    if (shards < 1) shards = devices;
    if (batch < 0) batch = 0;
    if (decode_threads < 1) decode_threads = 1;
    {
        // --devices names HIP ordinals 0 .. N-1: refuse up front what the machine does not have (FactoryConfig::cuda_device,
        // include/factory/PipelineFactory.hpp:24, is taken on trust by the reference)
        int ndev = 0;
        const int rc = aria_device_count(&ndev);
        if (rc != ARIA_OK || ndev < devices) {
            std::fprintf(stderr, "--devices %d: %d HIP device%s present%s%s\n", devices, ndev, ndev == 1 ? "" : "s",
                         rc != ARIA_OK ? ": " : "", rc != ARIA_OK ? aria_status_string(rc) : "");
            return -1;
        }
    }
    io::AslSequence seq;
    if (!seq.load(argv[1])) {
        std::fprintf(stderr, "Failed to load dataset from %s\n", argv[1]);
        return -1;                                                        // euroc_eval.cpp:75-78
    }
    std::printf("Loaded: %zu images\n", seq.size());
    const std::size_t N = seq.size();
    if ((std::size_t)shards > N && N > 0) shards = (int)N;

    std::vector<FrameRecord> rec(N);
    std::vector<std::string> errors((size_t)shards);
    std::atomic<std::size_t> done{0};
    int w = 0, h = 0;
    const bool sharded = shards > 1;
    const bool posthoc_loop = loop && (sharded || batch > 0);      // the loop step over the merged stream, after the shards
    std::vector<pipeline::BatchStats> bstats((size_t)shards);
    const auto t0 = std::chrono::steady_clock::now();

    // one shard = one FrontEnd (its own extractor + matcher handles on its device) over frames [first, hi)
    auto run_shard = [&](int s) {
        try {
            const pipeline::ShardPlan sp = pipeline::shardPlan(N, s, shards);
            if (batch > 0) {
                pipeline::BatchFrontEndConfig bc;
                bc.hip_device = s % devices;
                bc.max_features = max_features;
                bc.chunk = batch;
                bc.decode_threads = decode_threads;
                bc.legacy_order = legacy;
                pipeline::BatchFrontEnd bfe(bc);
                bfe.run(seq, sp.first, sp.lo, sp.hi, [&](std::size_t i, const core::Frame& f, const std::vector<core::Match>& m) {
                    FrameRecord& o = rec[i];
                    o.keypoints = f.numKeypoints();
                    o.matches = m.size();
                    o.hash = frame_hash(f, m);
                    // only frames that can become keyframes are kept for the loop step (>= keyframe_min_matches matches,
                    // euroc_eval.cpp:179): a long sequence does not hold a copy of every frame until the end
                    if (posthoc_loop && (int)m.size() >= pipeline::FrontEndConfig{}.keyframe_min_matches) o.frame = std::make_unique<core::Frame>(f);
                    ++done;
                });
                bstats[(size_t)s] = bfe.stats();
                if (s == 0) { w = bfe.width(); h = bfe.height(); }
                return;
            }
            factory::HipFactoryConfig fc;                                 // PipelineFactory's HIP mode (aria_hip/HipFactory.hpp)
            fc.hip_device = s % devices;
            fc.max_features = max_features;
            fc.frontend.legacy_order = legacy;
            fc.enable_loop_closure = loop && !posthoc_loop;               // LoopClosureDetector(200, 0.4, 50), euroc_eval.cpp:103
            std::unique_ptr<pipeline::FrontEnd> fe = factory::createHip(fc);
            std::vector<std::uint8_t> gray;
            int fw = 0, fh = 0;
            auto t_last = std::chrono::steady_clock::now();
            for (std::size_t i = sp.first; i < sp.hi; i++) {
                seq.read(i, gray, fw, fh);
                const pipeline::FrontEndResult& r = fe->processFrame(gray.data(), fw, fh, seq.at(i).timestamp);
                if (i < sp.lo) continue;                                   // the halo frame only provides the previous descriptors
                FrameRecord& o = rec[i];
                o.keypoints = r.frame->numKeypoints();
                o.matches = r.matches.size();
                o.hash = frame_hash(*r.frame, r.matches);
                o.is_keyframe = r.is_keyframe;
                if (r.loop) { o.loop_match_id = (long long)r.loop->match_id; o.loop_score = r.loop->score; }
                if (posthoc_loop && (int)r.matches.size() >= fc.frontend.keyframe_min_matches) o.frame = std::make_unique<core::Frame>(*r.frame);
                const std::size_t d = ++done;
                if (s == 0) { w = fw; h = fh; }
                if (!sharded && d % 100 == 0) {                            // euroc_eval.cpp:271-277
                    const auto now = std::chrono::steady_clock::now();
                    const double fps = 100.0 / std::chrono::duration<double>(now - t_last).count();
                    t_last = now;
                    std::printf("Frame %zu/%zu | FPS: %.1f | keypoints: %zu | matches: %zu\n", d, N, fps, o.keypoints, o.matches);
                }
            }
        } catch (const std::exception& e) {
            errors[(size_t)s] = e.what();
        }
    };
    if (!sharded) {
        run_shard(0);
    } else {
        std::vector<std::thread> th;
        for (int s = 0; s < shards; s++) th.emplace_back(run_shard, s);
        for (auto& t : th) t.join();
    }
    for (int s = 0; s < shards; s++)
        if (!errors[(size_t)s].empty()) { std::fprintf(stderr, "shard %d: %s\n", s, errors[(size_t)s].c_str()); return 1; }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    // the loop-closure step of a sharded or batched run: the merged stream, in frame order (euroc_eval.cpp:230-247)
    if (posthoc_loop && N > 0) {
        factory::HipFactoryConfig fc;
        fc.max_features = max_features;
        adapters::hip::HipLoopDetector ld(fc.loop_min_frames_between, fc.loop_min_score, fc.loop_min_matches,
                                          ((max_features + 8 * 64 + 63) / 64) * 64, 500, nullptr, 0);
        for (std::size_t i = 1; i < N; i++) {
            FrameRecord& o = rec[i];
            if ((int)o.matches < fc.frontend.keyframe_min_matches || !o.frame) continue;
            core::KeyFrame kf;
            kf.id = i;                                                     // FrontEnd numbers frames from 0 in sequence order
            kf.timestamp = seq.at(i).timestamp;
            kf.frame = *o.frame;
            kf.frame.id = i;
            auto lp = ld.detect(kf);
            ld.addKeyFrame(kf);
            o.is_keyframe = true;
            if (lp) { o.loop_match_id = (long long)lp->match_id; o.loop_score = lp->score; }
        }
    }

    long long total_kp = 0, total_matches = 0, n_keyframes = 0, n_loops = 0;
    std::ofstream out;
    if (!csv.empty()) { out.open(csv); out << std::setprecision(17); out << "frame,timestamp,keypoints,matches,hash,keyframe,loop_match_id,loop_score\n"; }
    for (std::size_t i = 0; i < N; i++) {
        const FrameRecord& o = rec[i];
        total_kp += (long long)o.keypoints;
        total_matches += (long long)o.matches;
        n_keyframes += o.is_keyframe ? 1 : 0;
        n_loops += o.loop_match_id >= 0 ? 1 : 0;
        if (out.is_open())
            out << i << ',' << std::to_string(seq.at(i).timestamp) << ',' << o.keypoints << ',' << o.matches << ',' << o.hash << ','
                << (o.is_keyframe ? 1 : 0) << ',' << o.loop_match_id << ',' << o.loop_score << '\n';
    }
    std::printf("frames %zu size %dx%d mean_keypoints %.2f mean_matches %.2f fps %.1f (PNG decode + H2D + extract + match + D2H; %d shard%s on %d device%s)\n",
                N, w, h, N ? (double)total_kp / N : 0.0, N > 1 ? (double)total_matches / (N - 1) : 0.0, N / secs, shards,
                shards > 1 ? "s" : "", devices, devices > 1 ? "s" : "");
    if (batch > 0) {
        pipeline::BatchStats t;
        for (const pipeline::BatchStats& b : bstats) {
            t.frames += b.frames; t.chunks += b.chunks; t.decode_s += b.decode_s; t.h2d_s += b.h2d_s; t.h2d_bytes += b.h2d_bytes;
            t.gpu_s += b.gpu_s; t.d2h_s += b.d2h_s; t.d2h_bytes += b.d2h_bytes; t.deliver_s += b.deliver_s; t.wall_s = std::max(t.wall_s, b.wall_s);
        }
        // stage times are summed over shards and chunks; the stages overlap (decode of chunk c + 1 and its upload run beside
        // the kernels of chunk c), so they do not add up to the wall time
        std::printf("batch %d frames/chunk, %zu chunks, %d decode threads/shard | decode %.3f s (%.0f frames/s) | H2D %.3f s (%.2f GB/s) | "
                    "extract+match kernels %.3f s (%.0f frames/s) | D2H %.3f s (%.2f GB/s) | deliver %.3f s | wall %.3f s\n",
                    batch, t.chunks, decode_threads, t.decode_s, t.decode_s > 0 ? t.frames / t.decode_s : 0.0, t.h2d_s,
                    t.h2d_s > 0 ? t.h2d_bytes / t.h2d_s * 1e-9 : 0.0, t.gpu_s, t.gpu_s > 0 ? t.frames / t.gpu_s : 0.0, t.d2h_s,
                    t.d2h_s > 0 ? t.d2h_bytes / t.d2h_s * 1e-9 : 0.0, t.deliver_s, t.wall_s);
    }
    if (loop) std::printf("keyframes %lld loops %lld\n", n_keyframes, n_loops);
    return 0;
}
