// See aria_hip/HipMatcher.hpp. Mirrors src/adapters/gpu/CudaMatcher.cpp:28-68 and
// src/legacy/LoopClosure.cpp:72-114 of the reference.
#include "aria_hip/HipMatcher.hpp"

#include <algorithm>
#include <stdexcept>
#include <string>

#include "aria_orb_hip.h"

namespace aria::adapters::hip {

static_assert(sizeof(core::Match) == sizeof(aria_match), "Match must stay 12 bytes (Types.hpp:97-101)");

namespace {
[[noreturn]] void fail(const char* where, int status) {
    std::string msg = std::string("HipMatcher: ") + where + ": " + aria_status_string(status);
    const char* hip = aria_last_hip_error();
    if (hip && hip[0]) msg += std::string(" [") + hip + "]";
    throw std::runtime_error(msg);
}
}  // namespace

HipMatcher::HipMatcher(void* stream, int device) : stream_(stream), device_(device) {}
HipMatcher::~HipMatcher() { aria_matcher_destroy(m_); }

void HipMatcher::ensure(int nq, int nt) {
    if (m_ && nq <= cap_q_ && nt <= cap_t_) return;
    aria_matcher_destroy(m_);
    m_ = nullptr;
    aria_matcher_config cfg;
    aria_matcher_default_config(&cfg);
    cfg.device = device_;
    cfg.stream = stream_;
    cfg.max_query = std::max({nq, cap_q_, 4096});
    cfg.max_train = std::max({nt, cap_t_, 4096});
    int rc = aria_matcher_create(&cfg, &m_);
    if (rc != ARIA_OK) fail("aria_matcher_create", rc);
    cap_q_ = cfg.max_query;
    cap_t_ = cfg.max_train;
}

void HipMatcher::match(const core::Frame& query, const core::Frame& train, std::vector<core::Match>& matches,
                       float ratio_threshold) {
    if (query.descriptors.empty() || train.descriptors.empty()) return;        // CudaMatcher.cpp:35-37
    const int nq = (int)query.numKeypoints(), nt = (int)train.numKeypoints();   // :41-42 rows = numKeypoints()
    if ((size_t)nq * 32 > query.descriptors.size() || (size_t)nt * 32 > train.descriptors.size())
        fail("match: descriptors shorter than numKeypoints()*32", ARIA_E_INVALID);
    ensure(nq, nt);
    buf_.resize((size_t)std::max(nq, 1));
    int n = 0;
    int rc = aria_matcher_match(m_, query.descriptors.data(), nq, train.descriptors.data(), nt, ratio_threshold,
                                reinterpret_cast<aria_match*>(buf_.data()), (int)buf_.size(), &n);
    if (rc != ARIA_OK) fail("aria_matcher_match", rc);
    matches.insert(matches.end(), buf_.begin(), buf_.begin() + n);            // :65 push_back, never cleared
}

void HipMatcher::matchDevice(const std::uint8_t* d_query, int nq, const std::uint8_t* d_train, int nt,
                             std::vector<core::Match>& matches, float ratio_threshold) {
    ensure(std::max(nq, 1), std::max(nt, 1));
    buf_.resize((size_t)std::max(nq, 1));
    int n = 0;
    int rc = aria_matcher_match_device(m_, d_query, nq, d_train, nt, ratio_threshold, reinterpret_cast<aria_match*>(buf_.data()),
                                       (int)buf_.size(), &n);
    if (rc != ARIA_OK) fail("aria_matcher_match_device", rc);
    matches.insert(matches.end(), buf_.begin(), buf_.begin() + n);
}

void HipMatcher::retainDevice(const std::uint8_t* d_desc, int n) {
    ensure(std::max(n, 1), 1);
    int rc = aria_matcher_retain_device(m_, d_desc, n);
    if (rc != ARIA_OK) fail("aria_matcher_retain_device", rc);
}

int HipMatcher::residentRows() const { return m_ ? aria_matcher_resident_rows(m_) : -1; }

bool HipMatcher::matchDeviceAsync(const std::uint8_t* d_new, const int* d_count, int rows_max, bool new_is_query,
                                  float ratio_threshold) {
    if (!m_ || rows_max > cap_q_ || rows_max > cap_t_ || aria_matcher_resident_rows(m_) < 1) return false;
    int rc = aria_matcher_match_device_async(m_, d_new, d_count, rows_max, new_is_query ? 1 : 0, ratio_threshold);
    if (rc != ARIA_OK) fail("aria_matcher_match_device_async", rc);
    return true;
}

void HipMatcher::finishDevice(int n_new, std::vector<core::Match>& matches) {
    buf_.resize((size_t)std::max(cap_q_, 1));
    int n = 0;
    int rc = aria_matcher_finish(m_, n_new, reinterpret_cast<aria_match*>(buf_.data()), (int)buf_.size(), &n);
    if (rc != ARIA_OK) fail("aria_matcher_finish", rc);
    matches.insert(matches.end(), buf_.begin(), buf_.begin() + n);
}

void HipMatcher::matchMultiple(const core::Frame& query, const std::vector<core::Frame>& candidates,
                               std::vector<std::vector<core::Match>>& all_matches, float ratio_threshold) {
    all_matches.resize(candidates.size());                                      // IMatcher.hpp:33
    if (candidates.empty() || query.descriptors.empty()) return;                // every match() would return at once
    const int nq = (int)query.numKeypoints();
    if ((size_t)nq * 32 > query.descriptors.size()) fail("matchMultiple: descriptors shorter than numKeypoints()*32", ARIA_E_INVALID);
    std::vector<const std::uint8_t*> ptrs(candidates.size());
    std::vector<int> nts(candidates.size()), n_out(candidates.size());
    int max_nt = 1;
    for (size_t i = 0; i < candidates.size(); i++) {
        const core::Frame& c = candidates[i];
        nts[i] = c.descriptors.empty() ? 0 : (int)c.numKeypoints();             // empty train set: untouched (CudaMatcher.cpp:35-37)
        if ((size_t)nts[i] * 32 > c.descriptors.size()) fail("matchMultiple: descriptors shorter than numKeypoints()*32", ARIA_E_INVALID);
        ptrs[i] = c.descriptors.data();
        max_nt = std::max(max_nt, nts[i]);
    }
    ensure(nq, max_nt);
    buf_.resize(candidates.size() * (size_t)nq);
    int rc = aria_matcher_match_multi(m_, query.descriptors.data(), nq, ptrs.data(), nts.data(), (int)candidates.size(),
                                      ratio_threshold, reinterpret_cast<aria_match*>(buf_.data()), nq, n_out.data());
    if (rc != ARIA_OK) fail("aria_matcher_match_multi", rc);
    for (size_t i = 0; i < candidates.size(); i++)
        all_matches[i].insert(all_matches[i].end(), buf_.begin() + i * (size_t)nq, buf_.begin() + i * (size_t)nq + n_out[i]);
}

std::vector<std::pair<int, double>> HipMatcher::findLoopCandidates(const core::Frame& query,
                                                                  const std::vector<core::Frame>& keyframes,
                                                                  int min_frames_between) {
    std::vector<std::pair<int, double>> candidates;
    if (query.descriptors.empty()) return candidates;                          // LoopClosure.cpp:75
    const int nq = (int)query.numKeypoints();
    if ((size_t)nq * 32 > query.descriptors.size()) fail("findLoopCandidates: descriptors shorter than numKeypoints()*32", ARIA_E_INVALID);
    // the keyframes that pass the host-side filters (:81, :83) go to the device in one batch: one upload of the query,
    // one kNN-2 launch over all of them (the reference runs one CPU knnMatch per keyframe, :87)
    std::vector<int> which;
    std::vector<const std::uint8_t*> ptrs;
    std::vector<int> nts;
    int max_nt = 1;
    for (size_t i = 0; i < keyframes.size(); i++) {
        const core::Frame& kf = keyframes[i];
        if ((long long)query.id - (long long)kf.id < (long long)min_frames_between) continue;   // :81
        if (kf.descriptors.empty()) continue;                                  // :83
        if (kf.numKeypoints() * 32 > kf.descriptors.size()) fail("findLoopCandidates: descriptors shorter than numKeypoints()*32", ARIA_E_INVALID);
        which.push_back((int)i);
        ptrs.push_back(kf.descriptors.data());
        nts.push_back((int)kf.numKeypoints());
        max_nt = std::max(max_nt, nts.back());
    }
    if (which.empty()) return candidates;
    ensure(nq, max_nt);
    std::vector<int> good(which.size());
    int rc = aria_matcher_count_good_multi(m_, query.descriptors.data(), nq, ptrs.data(), nts.data(), (int)which.size(), 0.7,
                                           good.data());                       // :90-95, double literal 0.7
    if (rc != ARIA_OK) fail("aria_matcher_count_good_multi", rc);
    for (size_t k = 0; k < which.size(); k++) {
        const double score = (double)good[k] / std::max(1, nq);               // :98
        if (score > 0.1) candidates.push_back({which[k], score});              // :99
    }
    std::stable_sort(candidates.begin(), candidates.end(),
                     [](const auto& a, const auto& b) { return a.second > b.second; });   // :105-106
    if (candidates.size() > 5) candidates.resize(5);                           // :109-111
    return candidates;
}

}  // namespace aria::adapters::hip
