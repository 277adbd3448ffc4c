// See aria_hip/BatchFrontEnd.hpp.
#include "aria_hip/BatchFrontEnd.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>

#include "aria_orb_hip.h"

namespace aria::pipeline {

static_assert(sizeof(core::KeyPoint) == sizeof(aria_keypoint) && sizeof(core::Match) == sizeof(aria_match),
              "the C records are the reference's records (include/core/Types.hpp:9-15, 97-101)");

namespace {
constexpr int kHostSlots = 3;      // pinned image buffers: the producer may be two chunks ahead of the copy stream
constexpr int kDevSlots = 2;       // device images + outputs, and pinned result buffers: chunk c computes while c - 1 drains

[[noreturn]] void fail(const char* where, int status) {
    std::string msg = std::string("BatchFrontEnd: ") + where + ": " + aria_status_string(status);
    const char* hip = aria_last_hip_error();
    if (hip && hip[0]) msg += std::string(" (") + hip + ")";
    throw std::runtime_error(msg);
}
void ck(int status, const char* where) { if (status != ARIA_OK) fail(where, status); }
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

struct BatchFrontEnd::Impl {
    int dev = 0;
    void* s_in = nullptr;          // copy stream (H2D)
    void* s_c = nullptr;           // compute stream: both handles live on it
    void* s_out = nullptr;         // result stream (D2H)
    aria_orb_t orb = nullptr;
    aria_matcher_t mat = nullptr;
    int kp_cap = 0, chunk = 0;
    std::size_t img_bytes = 0;
    // host image ring (pinned)
    std::uint8_t* h_img[kHostSlots] = {};
    // per device slot
    std::uint8_t* d_img[kDevSlots] = {};
    std::uint8_t* d_kps[kDevSlots] = {};
    std::uint8_t* d_desc[kDevSlots] = {};
    int* d_cnt[kDevSlots] = {};
    std::uint8_t* d_matches[kDevSlots] = {};
    int* d_nm[kDevSlots] = {};
    std::uint8_t* h_res[kDevSlots] = {};       // pinned: counts | nmatches | keypoints | descriptors | matches of one chunk
    void* ev[kDevSlots][6] = {};               // h2d start / end, compute start / end, d2h start / end
    std::size_t off_cnt = 0, off_nm = 0, off_kps = 0, off_desc = 0, off_mat = 0, res_bytes = 0;

    void release() {
        if (orb) { aria_orb_destroy(orb); orb = nullptr; }
        if (mat) { aria_matcher_destroy(mat); mat = nullptr; }
        for (int s = 0; s < kDevSlots; s++) {
            aria_device_free(dev, d_img[s]); aria_device_free(dev, d_kps[s]); aria_device_free(dev, d_desc[s]);
            aria_device_free(dev, d_cnt[s]); aria_device_free(dev, d_matches[s]); aria_device_free(dev, d_nm[s]);
            aria_host_free_pinned(h_res[s]);
            d_img[s] = d_kps[s] = d_desc[s] = d_matches[s] = h_res[s] = nullptr; d_cnt[s] = d_nm[s] = nullptr;
            for (int k = 0; k < 6; k++) { aria_event_destroy(dev, ev[s][k]); ev[s][k] = nullptr; }
        }
        for (int s = 0; s < kHostSlots; s++) { aria_host_free_pinned(h_img[s]); h_img[s] = nullptr; }
        aria_stream_destroy(dev, s_in); aria_stream_destroy(dev, s_c); aria_stream_destroy(dev, s_out);
        s_in = s_c = s_out = nullptr;
    }
};

BatchFrontEnd::BatchFrontEnd(const BatchFrontEndConfig& cfg) : p_(new Impl), cfg_(cfg) {
    if (cfg_.chunk < 1 || cfg_.max_features < 1 || cfg_.decode_threads < 1) { delete p_; throw std::invalid_argument("BatchFrontEnd: bad configuration"); }
    int ndev = 0;
    const int rc = aria_device_count(&ndev);
    if (rc != ARIA_OK || cfg_.hip_device < 0 || cfg_.hip_device >= ndev) {
        delete p_;
        fail("device", rc != ARIA_OK ? rc : ARIA_E_NO_DEVICE);      // no CPU fallback: a front end without its GPU is an error
    }
    p_->dev = cfg_.hip_device;
    p_->chunk = cfg_.chunk;
}

BatchFrontEnd::~BatchFrontEnd() {
    p_->release();
    delete p_;
}

void BatchFrontEnd::run(const io::AslSequence& seq, std::size_t first, std::size_t lo, std::size_t hi, const BatchSink& sink) {
    stats_ = BatchStats{};
    if (first >= hi) return;
    if (hi > seq.size() || lo < first) throw std::invalid_argument("BatchFrontEnd::run: bad frame range");
    Impl& P = *p_;
    const double t_wall0 = now_s();
    const int dev = P.dev, chunk = P.chunk;

    // ---- image size (the first frame is decoded once ahead of the pipeline), handles and buffers ----
    {
        std::vector<std::uint8_t> g;
        int w = 0, h = 0;
        seq.read(first, g, w, h);
        if (P.orb && (w != w_ || h != h_)) P.release();
        w_ = w; h_ = h;
    }
    P.img_bytes = (std::size_t)w_ * (std::size_t)h_;
    if (!P.orb) {
        ck(aria_stream_create(dev, &P.s_in), "aria_stream_create");
        ck(aria_stream_create(dev, &P.s_c), "aria_stream_create");
        ck(aria_stream_create(dev, &P.s_out), "aria_stream_create");
        aria_orb_config oc;
        aria_orb_default_config(&oc);
        oc.device = dev; oc.stream = P.s_c; oc.max_width = w_; oc.max_height = h_;
        oc.max_features = cfg_.max_features; oc.max_batch = chunk;
        ck(aria_orb_create(&oc, &P.orb), "aria_orb_create");
        P.kp_cap = aria_orb_kp_capacity(P.orb);
        aria_matcher_config mc;
        aria_matcher_default_config(&mc);
        mc.device = dev; mc.stream = P.s_c; mc.max_query = P.kp_cap; mc.max_train = P.kp_cap;
        ck(aria_matcher_create(&mc, &P.mat), "aria_matcher_create");
        const std::size_t cap = (std::size_t)P.kp_cap, nb = (std::size_t)chunk;
        P.off_cnt = 0;
        P.off_nm = P.off_cnt + 4 * (nb + 1);
        P.off_kps = (P.off_nm + 4 * nb + 63) & ~(std::size_t)63;
        P.off_desc = P.off_kps + nb * cap * sizeof(aria_keypoint);
        P.off_mat = P.off_desc + nb * cap * 32;
        P.res_bytes = P.off_mat + nb * cap * sizeof(aria_match);
        for (int s = 0; s < kHostSlots; s++) ck(aria_host_alloc_pinned(nb * P.img_bytes, (void**)&P.h_img[s]), "aria_host_alloc_pinned");
        for (int s = 0; s < kDevSlots; s++) {
            ck(aria_device_alloc(dev, nb * P.img_bytes, (void**)&P.d_img[s]), "aria_device_alloc");
            ck(aria_device_alloc(dev, (nb + 1) * cap * sizeof(aria_keypoint), (void**)&P.d_kps[s]), "aria_device_alloc");
            ck(aria_device_alloc(dev, (nb + 1) * cap * 32, (void**)&P.d_desc[s]), "aria_device_alloc");
            ck(aria_device_alloc(dev, (nb + 1) * 4, (void**)&P.d_cnt[s]), "aria_device_alloc");
            ck(aria_device_alloc(dev, nb * cap * sizeof(aria_match), (void**)&P.d_matches[s]), "aria_device_alloc");
            ck(aria_device_alloc(dev, nb * 4, (void**)&P.d_nm[s]), "aria_device_alloc");
            ck(aria_host_alloc_pinned(P.res_bytes, (void**)&P.h_res[s]), "aria_host_alloc_pinned");
            for (int k = 0; k < 6; k++) ck(aria_event_create(dev, &P.ev[s][k]), "aria_event_create");
        }
    }
    const std::size_t cap = (std::size_t)P.kp_cap;
    const std::size_t n_total = hi - first;
    const std::size_t n_chunks = (n_total + (std::size_t)chunk - 1) / (std::size_t)chunk;

    // ---- producer: decode chunk by chunk into the pinned ring ----
    std::mutex mu;
    std::condition_variable cv;
    int filled[kHostSlots];                     // chunk id held by the slot, -1 = free
    for (int s = 0; s < kHostSlots; s++) filled[s] = -1;
    std::exception_ptr producer_error;
    std::atomic<bool> stop{false};
    double decode_s = 0;
    std::thread producer([&] {
        try {
            for (std::size_t c = 0; c < n_chunks && !stop; c++) {
                const int hs = (int)(c % kHostSlots);
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return filled[hs] < 0 || stop.load(); });
                    if (stop) return;
                }
                const std::size_t a = first + c * (std::size_t)chunk, n = std::min((std::size_t)chunk, hi - a);
                const double t0 = now_s();
                const int T = (int)std::min<std::size_t>((std::size_t)cfg_.decode_threads, n);
                std::vector<std::exception_ptr> errs((std::size_t)T);
                auto work = [&](int t) {
                    try {
                        std::vector<std::uint8_t> g;
                        for (std::size_t j = (std::size_t)t; j < n; j += (std::size_t)T) {
                            int w = 0, h = 0;
                            seq.read(a + j, g, w, h);
                            if (w != w_ || h != h_) throw std::runtime_error("BatchFrontEnd: image " + seq.at(a + j).path + " has another size than the first one");
                            std::memcpy(P.h_img[hs] + j * P.img_bytes, g.data(), P.img_bytes);
                        }
                    } catch (...) { errs[(std::size_t)t] = std::current_exception(); }
                };
                std::vector<std::thread> th;
                for (int t = 1; t < T; t++) th.emplace_back(work, t);
                work(0);
                for (auto& x : th) x.join();
                for (auto& e : errs) if (e) std::rethrow_exception(e);
                decode_s += now_s() - t0;
                { std::lock_guard<std::mutex> lk(mu); filled[hs] = (int)c; }
                cv.notify_all();
            }
        } catch (...) {
            { std::lock_guard<std::mutex> lk(mu); producer_error = std::current_exception(); }
            cv.notify_all();
        }
    });
    struct Joiner {                             // whatever happens below, the producer is stopped and joined
        std::thread& t; std::atomic<bool>& stop; std::condition_variable& cv;
        ~Joiner() { stop = true; cv.notify_all(); if (t.joinable()) t.join(); }
    } joiner{producer, stop, cv};

    // ---- consumer: copy, compute, results ----
    std::size_t chunk_n[kDevSlots] = {0, 0}, chunk_a[kDevSlots] = {0, 0};
    core::Frame fr;
    std::vector<core::Match> mv;
    auto drain = [&](std::size_t c) {
        const int ds = (int)(c % kDevSlots);
        ck(aria_event_synchronize(dev, P.ev[ds][5]), "aria_event_synchronize");
        { std::lock_guard<std::mutex> lk(mu); filled[c % kHostSlots] = -1; }      // its upload finished long ago: the producer may refill it
        cv.notify_all();
        ck(aria_orb_check(P.orb), "aria_orb_check");                             // deferred errors of the pass (tie storms beyond kp_cap, overflows)
        float ms = 0;
        ck(aria_event_elapsed_ms(P.ev[ds][0], P.ev[ds][1], &ms), "aria_event_elapsed_ms"); stats_.h2d_s += ms * 1e-3;
        ck(aria_event_elapsed_ms(P.ev[ds][2], P.ev[ds][3], &ms), "aria_event_elapsed_ms"); stats_.gpu_s += ms * 1e-3;
        ck(aria_event_elapsed_ms(P.ev[ds][4], P.ev[ds][5], &ms), "aria_event_elapsed_ms"); stats_.d2h_s += ms * 1e-3;
        const double t0 = now_s();
        const std::uint8_t* R = P.h_res[ds];
        const int* cnt = reinterpret_cast<const int*>(R + P.off_cnt);
        const int* nm = reinterpret_cast<const int*>(R + P.off_nm);
        for (std::size_t j = 0; j < chunk_n[ds]; j++) {
            const std::size_t idx = chunk_a[ds] + j;
            if (idx < lo) continue;                                               // the halo frame only provides the previous descriptors
            const std::size_t n = (std::size_t)std::min(std::max(cnt[1 + j], 0), P.kp_cap);
            fr.id = idx;
            fr.timestamp = seq.at(idx).timestamp;
            fr.width = w_; fr.height = h_;
            const core::KeyPoint* kp = reinterpret_cast<const core::KeyPoint*>(R + P.off_kps + j * cap * sizeof(aria_keypoint));
            fr.keypoints.assign(kp, kp + n);                                      // OrbCudaExtractor.cpp:109-123
            const std::uint8_t* de = R + P.off_desc + j * cap * 32;
            fr.descriptors.assign(de, de + n * 32);                               // :126-127
            const std::size_t m = (std::size_t)std::min(std::max(nm[j], 0), P.kp_cap);
            const core::Match* mp = reinterpret_cast<const core::Match*>(R + P.off_mat + j * cap * sizeof(aria_match));
            mv.assign(mp, mp + m);
            sink(idx, fr, mv);
        }
        stats_.deliver_s += now_s() - t0;
        stats_.frames += chunk_n[ds];
        stats_.chunks++;
    };

    for (std::size_t c = 0; c < n_chunks; c++) {
        const int hs = (int)(c % kHostSlots), ds = (int)(c % kDevSlots);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return filled[hs] == (int)c || producer_error; });
            if (producer_error) std::rethrow_exception(producer_error);
        }
        const std::size_t a = first + c * (std::size_t)chunk, n = std::min((std::size_t)chunk, hi - a);
        chunk_a[ds] = a; chunk_n[ds] = n;
        // copy stream: chunk c to the device (device slot ds was last read by chunk c - 2, drained in the previous iteration)
        ck(aria_event_record(dev, P.ev[ds][0], P.s_in), "aria_event_record");
        ck(aria_copy_h2d_async(dev, P.s_in, P.d_img[ds], P.h_img[hs], n * P.img_bytes), "aria_copy_h2d_async");
        ck(aria_event_record(dev, P.ev[ds][1], P.s_in), "aria_event_record");
        stats_.h2d_bytes += (double)(n * P.img_bytes);
        // compute stream: the previous chunk's last descriptor set becomes row 0, then extraction and the pairs (f, f - 1)
        ck(aria_stream_wait_event(dev, P.s_c, P.ev[ds][1]), "aria_stream_wait_event");
        ck(aria_event_record(dev, P.ev[ds][2], P.s_c), "aria_event_record");
        if (c == 0) {
            ck(aria_fill_async(dev, P.s_c, P.d_cnt[ds], 0, 4), "aria_fill_async");           // no previous frame: an empty train set
        } else {
            const int ps = ds ^ 1;
            const std::size_t last = chunk_n[ps];                                  // row of the previous chunk's last frame
            ck(aria_copy_d2d_async(dev, P.s_c, P.d_desc[ds], P.d_desc[ps] + last * cap * 32, cap * 32), "aria_copy_d2d_async");
            ck(aria_copy_d2d_async(dev, P.s_c, P.d_cnt[ds], P.d_cnt[ps] + last, 4), "aria_copy_d2d_async");
        }
        ck(aria_orb_extract_batch_device(P.orb, P.d_img[ds], (int)n, w_, h_, (std::int64_t)P.img_bytes, w_,
                                         reinterpret_cast<aria_keypoint*>(P.d_kps[ds]) + cap, P.d_desc[ds] + cap * 32, P.d_cnt[ds] + 1,
                                         P.kp_cap), "aria_orb_extract_batch_device");
        const std::uint8_t* cur = P.d_desc[ds] + cap * 32;
        const std::uint8_t* prv = P.d_desc[ds];
        const int* ncur = P.d_cnt[ds] + 1;
        const int* nprv = P.d_cnt[ds];
        if (cfg_.legacy_order)       // query = previous, train = current (src/euroc_eval.cpp:168-169)
            ck(aria_matcher_match_batch_device(P.mat, prv, nprv, cur, ncur, (int)n, (std::int64_t)(cap * 32), cfg_.ratio_threshold,
                                               reinterpret_cast<aria_match*>(P.d_matches[ds]), P.d_nm[ds], P.kp_cap), "aria_matcher_match_batch_device");
        else                         // query = current, train = previous (docs/milestones/H12_CLEAN_ARCHITECTURE.md:601)
            ck(aria_matcher_match_batch_device(P.mat, cur, ncur, prv, nprv, (int)n, (std::int64_t)(cap * 32), cfg_.ratio_threshold,
                                               reinterpret_cast<aria_match*>(P.d_matches[ds]), P.d_nm[ds], P.kp_cap), "aria_matcher_match_batch_device");
        ck(aria_event_record(dev, P.ev[ds][3], P.s_c), "aria_event_record");
        // result stream: everything the sink needs, into the pinned result slot (last read when chunk c - 2 was delivered)
        ck(aria_stream_wait_event(dev, P.s_out, P.ev[ds][3]), "aria_stream_wait_event");
        ck(aria_event_record(dev, P.ev[ds][4], P.s_out), "aria_event_record");
        std::uint8_t* R = P.h_res[ds];
        ck(aria_copy_d2h_async(dev, P.s_out, R + P.off_cnt, P.d_cnt[ds], 4 * (n + 1)), "aria_copy_d2h_async");
        ck(aria_copy_d2h_async(dev, P.s_out, R + P.off_nm, P.d_nm[ds], 4 * n), "aria_copy_d2h_async");
        ck(aria_copy_d2h_async(dev, P.s_out, R + P.off_kps, P.d_kps[ds] + cap * sizeof(aria_keypoint), n * cap * sizeof(aria_keypoint)), "aria_copy_d2h_async");
        ck(aria_copy_d2h_async(dev, P.s_out, R + P.off_desc, P.d_desc[ds] + cap * 32, n * cap * 32), "aria_copy_d2h_async");
        ck(aria_copy_d2h_async(dev, P.s_out, R + P.off_mat, P.d_matches[ds], n * cap * sizeof(aria_match)), "aria_copy_d2h_async");
        ck(aria_event_record(dev, P.ev[ds][5], P.s_out), "aria_event_record");
        stats_.d2h_bytes += (double)(4 * (2 * n + 1) + n * cap * (sizeof(aria_keypoint) + 32 + sizeof(aria_match)));
        if (c > 0) drain(c - 1);
    }
    drain(n_chunks - 1);
    stop = true;
    cv.notify_all();
    producer.join();
    stats_.decode_s = decode_s;
    stats_.wall_s = now_s() - t_wall0;
}

}  // namespace aria::pipeline
