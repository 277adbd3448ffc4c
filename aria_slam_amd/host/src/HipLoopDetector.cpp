// See aria_hip/HipLoopDetector.hpp. Control flow of src/legacy/LoopClosure.cpp:24-114 of the reference.
#include "aria_hip/HipLoopDetector.hpp"

#include <algorithm>
#include <stdexcept>
#include <string>

#include "aria_orb_hip.h"

namespace aria::adapters::hip {

namespace {
[[noreturn]] void fail(const char* where, int status) {
    std::string msg = std::string("HipLoopDetector: ") + where + ": " + aria_status_string(status);
    const char* hip = aria_last_hip_error();
    if (hip && hip[0]) msg += std::string(" [") + hip + "]";
    throw std::runtime_error(msg);
}
}  // namespace

HipLoopDetector::HipLoopDetector(int min_frames_between, double min_score, int min_matches, int slot_rows, int capacity,
                                 void* stream, int device)
    : matcher_(stream, device), min_frames_between_(min_frames_between), min_score_(min_score), min_matches_(min_matches),
      slot_rows_(slot_rows) {
    int rc = aria_kfdb_create(device, stream, capacity, slot_rows, &db_);
    if (rc != ARIA_OK) fail("aria_kfdb_create", rc);
    good_.resize((size_t)capacity);
}

HipLoopDetector::~HipLoopDetector() { aria_kfdb_destroy(db_); }

int HipLoopDetector::size() const { return aria_kfdb_size(db_); }

std::uint64_t HipLoopDetector::keyframeId(int index) const {
    long long id = -1;
    int rc = aria_kfdb_info(db_, index, &id, nullptr);
    if (rc != ARIA_OK) fail("aria_kfdb_info", rc);
    return (std::uint64_t)id;
}

void HipLoopDetector::addKeyFrame(const core::KeyFrame& kf) {
    const int n = (int)kf.frame.numKeypoints();
    if ((size_t)n * 32 > kf.frame.descriptors.size()) fail("addKeyFrame: descriptors shorter than numKeypoints()*32", ARIA_E_INVALID);
    int rc = aria_kfdb_add(db_, (long long)kf.id, kf.frame.descriptors.data(), n);       // push_back + pop_front beyond 500
    if (rc != ARIA_OK) fail("aria_kfdb_add", rc);
}

std::vector<std::pair<int, double>> HipLoopDetector::findCandidates(const core::KeyFrame& query) {
    std::vector<std::pair<int, double>> candidates;
    if (query.frame.descriptors.empty()) return candidates;                        // LoopClosure.cpp:75
    const int nq = (int)query.frame.numKeypoints();
    if ((size_t)nq * 32 > query.frame.descriptors.size()) fail("findCandidates: descriptors shorter than numKeypoints()*32", ARIA_E_INVALID);
    matcher_.reserve(nq, slot_rows_);
    int n_kf = 0;
    int rc = aria_kfdb_scan(db_, matcher_.handle(), query.frame.descriptors.data(), nq, 0.7, good_.data(), (int)good_.size(), &n_kf);
    if (rc != ARIA_OK) fail("aria_kfdb_scan", rc);
    for (int i = 0; i < n_kf; i++) {
        long long id = 0;
        int cnt = 0;
        aria_kfdb_info(db_, i, &id, &cnt);
        if ((long long)query.id - id < (long long)min_frames_between_) continue;   // :81
        if (cnt <= 0) continue;                                                     // :83
        const double score = (double)good_[(size_t)i] / std::max(1, nq);           // :98
        if (score > 0.1) candidates.push_back({i, score});                          // :99
    }
    std::stable_sort(candidates.begin(), candidates.end(),
                     [](const auto& a, const auto& b) { return a.second > b.second; });   // :105-106
    if (candidates.size() > 5) candidates.resize(5);                                // :109-111
    return candidates;
}

std::optional<core::LoopCandidate> HipLoopDetector::detect(const core::KeyFrame& query) {
    if (size() < min_frames_between_) return std::nullopt;                          // :34-36
    const auto candidates = findCandidates(query);                                  // :39
    const int nq = (int)query.frame.numKeypoints();
    for (const auto& [idx, score] : candidates) {
        if (score < min_score_) continue;                                           // :42
        long long id = 0;
        int cnt = 0;
        aria_kfdb_info(db_, idx, &id, &cnt);
        if ((long long)query.id - id < (long long)min_frames_between_) continue;    // :47
        // verification: the ratio-0.7 match list (LoopClosure.cpp:120-131), at least min_matches of them. The keyframe is
        // matched where it lies in the database (no download + upload of its descriptors)
        match_buf_.resize((size_t)std::max(nq, 1));
        int n = 0;
        int rc = aria_kfdb_match(db_, matcher_.handle(), idx, query.frame.descriptors.data(), nq, 0.7f,
                                 reinterpret_cast<aria_match*>(match_buf_.data()), (int)match_buf_.size(), &n);
        if (rc != ARIA_OK) fail("aria_kfdb_match", rc);
        core::LoopCandidate cand{};
        cand.matches.assign(match_buf_.begin(), match_buf_.begin() + n);
        if ((int)cand.matches.size() < min_matches_) continue;
        if (verifier_ && !verifier_(query, (std::uint64_t)id, cand)) continue;
        cand.query_id = query.id;                                                   // :57-61
        cand.match_id = (std::uint64_t)id;
        cand.score = score;
        loop_count_++;                                                              // :63
        return cand;
    }
    return std::nullopt;
}

}  // namespace aria::adapters::hip
