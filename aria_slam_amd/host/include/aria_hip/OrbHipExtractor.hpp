// aria::adapters::hip::OrbHipExtractor -- MI355X drop-in for aria::adapters::gpu::OrbCudaExtractor
// (reference include/adapters/gpu/OrbCudaExtractor.hpp, src/adapters/gpu/OrbCudaExtractor.cpp).
// Same port (IFeatureExtractor), same constructor shape (max_features, stream), same extract / extractAsync /
// sync / setMaxFeatures / getMaxFeatures behaviour; the work goes through the C-ABI in include/aria_orb_hip.h
// instead of cv::cuda::ORB. Results follow CPU cv::ORB (the reference's Frame.cpp:45-49 path), in canonical
// keypoint order. Errors surface as std::runtime_error (the reference let cv::Exception propagate).
#pragma once
#include "aria_hip/compat.hpp"

struct aria_orb_s;

namespace aria::adapters::hip {

class OrbHipExtractor : public interfaces::IFeatureExtractor {
public:
    // stream: a hipStream_t passed as void* (borrowed), or nullptr for an internally owned stream
    // (OrbCudaExtractor.cpp:21-29). device: HIP ordinal (FactoryConfig::cuda_device, PipelineFactory.hpp:24).
    explicit OrbHipExtractor(int max_features = 1000, void* stream = nullptr, int device = 0);
    ~OrbHipExtractor() override;
    OrbHipExtractor(const OrbHipExtractor&) = delete;
    OrbHipExtractor& operator=(const OrbHipExtractor&) = delete;

    void extract(const std::uint8_t* image_data, int width, int height, core::Frame& frame) override;
    void extractAsync(const std::uint8_t* image_data, int width, int height, core::Frame& frame) override;
    void sync() override;
    void setMaxFeatures(int n) override;
    int getMaxFeatures() const override { return max_features_; }

    // OrbCudaExtractor::getGpuDescriptors() (include/adapters/gpu/OrbCudaExtractor.hpp:34-35, "get descriptors without
    // download (for GPU matching)"): where the result of the last extract / extractAsync lies on the device, for
    // HipMatcher::matchDevice*. The pointers stay the same from frame to frame; `count` is the device copy of the keypoint
    // count (what work queued behind extractAsync on the same stream reads), `n` the host copy (-1 while pending), `rows`
    // the block's row capacity. All null before the first extraction.
    struct DeviceResult {
        const core::KeyPoint* keypoints = nullptr;
        const std::uint8_t* descriptors = nullptr;
        const int* count = nullptr;
        int n = -1, rows = 0;
    };
    DeviceResult deviceResult() const;
    // creates (or grows) the C handle for images up to width x height, so that deviceResult()'s pointers exist before the
    // first extractAsync
    void reserve(int width, int height) { ensure(width, height); }

    // The C handle, for callers that want the device-resident batch entry points.
    aria_orb_s* handle() const { return h_; }
    void* streamArg() const { return stream_; }      // the stream given to the constructor (nullptr = own stream)
    int device() const { return device_; }

private:
    void ensure(int width, int height);
    void fill(core::Frame& frame, int width, int height, int n);
    int refetch(int rows);

    aria_orb_s* h_ = nullptr;
    void* stream_;
    int device_;
    int max_features_;
    int cap_w_ = 0, cap_h_ = 0;
    std::vector<core::KeyPoint> kp_buf_;
    std::vector<std::uint8_t> desc_buf_;
    core::Frame* pending_frame_ = nullptr;     // OrbCudaExtractor.hpp:47-49: one pending slot
    int pending_width_ = 0, pending_height_ = 0;
};

}  // namespace aria::adapters::hip
