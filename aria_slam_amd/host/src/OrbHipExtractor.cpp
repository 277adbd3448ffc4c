// See aria_hip/OrbHipExtractor.hpp. Mirrors the call sequence of the reference adapter
// (src/adapters/gpu/OrbCudaExtractor.cpp:64-216) with the C-ABI in place of cv::cuda::ORB.
#include "aria_hip/OrbHipExtractor.hpp"

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>

#include "aria_orb_hip.h"

namespace aria::adapters::hip {

static_assert(sizeof(core::KeyPoint) == sizeof(aria_keypoint), "KeyPoint must stay 24 bytes (Types.hpp:9-15)");

namespace {
[[noreturn]] void fail(const char* where, int status) {
    std::string msg = std::string("OrbHipExtractor: ") + where + ": " + aria_status_string(status);
    const char* hip = aria_last_hip_error();
    if (hip && hip[0]) msg += std::string(" [") + hip + "]";
    throw std::runtime_error(msg);
}
}  // namespace

OrbHipExtractor::OrbHipExtractor(int max_features, void* stream, int device)
    : stream_(stream), device_(device), max_features_(max_features) {
    if (max_features < 0) fail("constructor", ARIA_E_INVALID);
}

OrbHipExtractor::~OrbHipExtractor() { aria_orb_destroy(h_); }

// The reference allocates GpuMats per call; here device scratch is sized once for the largest image seen.
void OrbHipExtractor::ensure(int width, int height) {
    if (h_ && width <= cap_w_ && height <= cap_h_) return;
    if (pending_frame_) fail("resize while an async extract is pending", ARIA_E_BUSY);
    aria_orb_destroy(h_);
    h_ = nullptr;
    aria_orb_config cfg;
    aria_orb_default_config(&cfg);
    cfg.device = device_;
    cfg.stream = stream_;
    cfg.max_width = std::max(width, cap_w_);
    cfg.max_height = std::max(height, cap_h_);
    cfg.max_features = max_features_;
    cfg.max_batch = 1;
    int rc = aria_orb_create(&cfg, &h_);
    if (rc != ARIA_OK) fail("aria_orb_create", rc);
    cap_w_ = cfg.max_width;
    cap_h_ = cfg.max_height;
    const int cap = aria_orb_kp_capacity(h_);
    kp_buf_.resize((size_t)cap);
    desc_buf_.resize((size_t)cap * 32);
}

// OpenCV keeps every keypoint that ties with the last kept one, so a frame can need more rows than the plan provides:
// grow the buffers to what the call reported and copy the (still resident) result out again.
int OrbHipExtractor::refetch(int rows) {
    kp_buf_.resize((size_t)rows);
    desc_buf_.resize((size_t)rows * 32);
    int n = 0;
    return aria_orb_fetch_last(h_, reinterpret_cast<aria_keypoint*>(kp_buf_.data()), desc_buf_.data(), rows, &n);
}

void OrbHipExtractor::fill(core::Frame& frame, int width, int height, int n) {
    frame.width = width;                                       // OrbCudaExtractor.cpp:109-110
    frame.height = height;
    frame.keypoints.assign(kp_buf_.begin(), kp_buf_.begin() + n);          // :111-123 (clear + refill)
    frame.descriptors.assign(desc_buf_.begin(), desc_buf_.begin() + (size_t)n * 32);   // :126-127
}

void OrbHipExtractor::extract(const std::uint8_t* image_data, int width, int height, core::Frame& frame) {
    ensure(width, height);
    int n = 0;
    int rc = aria_orb_extract(h_, image_data, width, height, width, reinterpret_cast<aria_keypoint*>(kp_buf_.data()),
                              desc_buf_.data(), (int)kp_buf_.size(), &n);
    if (rc == ARIA_E_OUTPUT_TOO_SMALL) rc = refetch(n);            // tie storm: more keypoints than the plan's rows
    if (rc != ARIA_OK) fail("aria_orb_extract", rc);
    fill(frame, width, height, n);
}

void OrbHipExtractor::extractAsync(const std::uint8_t* image_data, int width, int height, core::Frame& frame) {
    ensure(width, height);
    int rc = aria_orb_extract_async(h_, image_data, width, height, width);
    if (rc != ARIA_OK) fail("aria_orb_extract_async", rc);
    pending_frame_ = &frame;                                   // OrbCudaExtractor.cpp:143-145
    pending_width_ = width;
    pending_height_ = height;
}

void OrbHipExtractor::sync() {
    if (!pending_frame_) return;                               // OrbCudaExtractor.cpp:177
    int n = 0;
    int rc = aria_orb_sync(h_, reinterpret_cast<aria_keypoint*>(kp_buf_.data()), desc_buf_.data(), (int)kp_buf_.size(), &n);
    if (rc == ARIA_E_OUTPUT_TOO_SMALL) rc = refetch(n);
    core::Frame* f = pending_frame_;
    pending_frame_ = nullptr;                                  // :209
    if (rc != ARIA_OK) fail("aria_orb_sync", rc);
    fill(*f, pending_width_, pending_height_, n);
}

OrbHipExtractor::DeviceResult OrbHipExtractor::deviceResult() const {
    DeviceResult r;
    if (!h_) return r;
    const aria_keypoint* kp = nullptr;
    int rc = aria_orb_last_device(h_, &kp, &r.descriptors, &r.count, &r.n, &r.rows);
    if (rc != ARIA_OK) return DeviceResult{};
    r.keypoints = reinterpret_cast<const core::KeyPoint*>(kp);
    return r;
}

void OrbHipExtractor::setMaxFeatures(int n) {                  // OrbCudaExtractor.cpp:212-216
    if (n < 0) fail("setMaxFeatures", ARIA_E_INVALID);
    max_features_ = n;
    if (!h_) return;
    int rc = aria_orb_set_max_features(h_, n);
    if (rc != ARIA_OK) fail("aria_orb_set_max_features", rc);
    const int cap = aria_orb_kp_capacity(h_);
    kp_buf_.resize((size_t)cap);
    desc_buf_.resize((size_t)cap * 32);
}

}  // namespace aria::adapters::hip
