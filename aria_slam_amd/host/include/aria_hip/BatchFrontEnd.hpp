// aria::pipeline::BatchFrontEnd -- the recorded-sequence form of the front end (round 4): the loop of the reference's
// harness, src/euroc_eval.cpp:128-176 (read image -> extract -> match against the previous frame), run over CHUNKS of frames
// through the device-resident batch entry points of the C-ABI (aria_orb_extract_batch_device +
// aria_matcher_match_batch_device) instead of one IFeatureExtractor::extract call per image.
//
//   producer thread : PNG decode (src/legacy/EuRoCReader.cpp:277-309) of chunk c + 1 by `decode_threads` workers into a ring
//                     of PINNED host buffers
//   copy stream     : H2D of chunk c + 1 (the reference uploads synchronously per frame: src/legacy/Frame.cpp:19,
//                     OrbCudaExtractor.cpp:83) while
//   compute stream  : chunk c -- every frame extracted in one batch pass, then kNN-2 + ratio test of every pair (f, f - 1)
//                     in one launch; the descriptors of a chunk's last frame are carried to the next chunk on the device
//                     (no frame is extracted twice, no descriptor is uploaded: the getGpuDescriptors() / matchGpu hand-off of
//                     OrbCudaExtractor.hpp:34-35 / CudaMatcher.hpp:22-28 at sequence scale)
//   result stream   : D2H of counts, keypoints, descriptors, matches into pinned buffers, handed to the caller's sink frame
//                     by frame, in order.
//
// Same results as FrontEnd::processFrame on the same images (tests/test_frontend_io.py compares the per-frame hashes of
// euroc_frontend --batch with the frame-at-a-time run's). The dynamic-object filter and the loop-closure step are the
// caller's (euroc_frontend runs the latter over the merged stream, as in its sharded mode).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

#include "aria_hip/AslSequence.hpp"
#include "aria_hip/compat.hpp"

namespace aria::pipeline {

struct BatchFrontEndConfig {
    int hip_device = 0;            // FactoryConfig::cuda_device (include/factory/PipelineFactory.hpp:24)
    int max_features = 2000;       // src/euroc_eval.cpp:88
    int chunk = 256;               // frames per batch pass (device scratch ~3.5 MB per 640x480 frame of the chunk)
    int decode_threads = 4;        // PNG decode workers of the producer
    float ratio_threshold = 0.75f; // src/euroc_eval.cpp:172
    bool legacy_order = false;     // true: query = previous, train = current (src/euroc_eval.cpp:168-169)
};

// Where the time went, per stage (summed over chunks). The stages overlap: wall_s is not their sum.
struct BatchStats {
    std::size_t frames = 0, chunks = 0;
    double decode_s = 0;           // producer: PNG decode into pinned memory (wall time of the decode stage, all workers)
    double h2d_s = 0;              // copy stream, by events
    double h2d_bytes = 0;
    double gpu_s = 0;              // compute stream: carry + batch extraction + batch match, by events
    double d2h_s = 0;              // result stream, by events
    double d2h_bytes = 0;
    double deliver_s = 0;          // host: building core::Frame / match vectors and the caller's sink
    double wall_s = 0;
};

// index = position of the frame in the sequence; the Frame and the match vector are only valid during the call
using BatchSink = std::function<void(std::size_t index, const core::Frame& frame, const std::vector<core::Match>& matches)>;

class BatchFrontEnd {
public:
    explicit BatchFrontEnd(const BatchFrontEndConfig& cfg);
    ~BatchFrontEnd();
    BatchFrontEnd(const BatchFrontEnd&) = delete;
    BatchFrontEnd& operator=(const BatchFrontEnd&) = delete;

    // Frames [first, hi) of the sequence are extracted; frames in [first, lo) (a shard's one-frame halo, aria_hip/Shard.hpp)
    // only provide the previous descriptors and are not delivered. The first extracted frame has no matches
    // (CudaMatcher.cpp:35-37: nothing to match against an empty set). Throws std::runtime_error on a failing status,
    // a bad image file, or images of different sizes.
    void run(const io::AslSequence& seq, std::size_t first, std::size_t lo, std::size_t hi, const BatchSink& sink);

    const BatchStats& stats() const { return stats_; }
    int width() const { return w_; }
    int height() const { return h_; }

private:
    struct Impl;
    Impl* p_;
    BatchFrontEndConfig cfg_;
    BatchStats stats_;
    int w_ = 0, h_ = 0;
};

}  // namespace aria::pipeline
