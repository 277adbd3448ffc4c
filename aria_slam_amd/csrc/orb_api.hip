// C-ABI of the extractor (include/aria_orb_hip.h): handle lifetime, scratch allocation, host-buffer and
// device-resident entry points. Replaces the body of the reference's OrbCudaExtractor
// (src/adapters/gpu/OrbCudaExtractor.cpp) -- there the work is cv::cuda::ORB calls, here it is the kernels in
// orb_kernels.hip. No CPU fallback exists: without a HIP device every entry point fails with ARIA_E_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"
#include "orb_kernels.h"
#include "orb_device.h"

namespace aria {
char* last_hip_error_buf() {
    static thread_local char buf[256] = {0};
    return buf;
}
}  // namespace aria

using namespace aria;

struct aria_orb_s {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int max_w = 0, max_h = 0, max_features = 0, max_batch = 1, tie_mode = 1, cand_cap_scale = 0, level_size_mode = 0;
    int band_qpct0 = 10;            // survivor-queue size of the band kernel at level 0 (% of a workgroup's pixels); self-tuned
    long long slow_blocks = 0;      // band-kernel workgroups that overflowed their queue since the last reset
    int last_n = -1;                // keypoints of the last completed single-frame extraction (still in the pinned buffers)
    int rows_needed = 0;            // rows the largest frame of the last checked batch call needed when kp_cap was too small

    Plan plan{};            // plan of the most recent (width, height)
    bool plan_valid = false;
    std::vector<uint32_t> tab_host;
    std::vector<int> bands_host;
    DeviceScratch D{};
    int kp_cap = 0;         // rows of the internal single-frame output buffers (grows when a tie storm needs more)
    int plan_rows = 0;      // sum over levels of (quota + tie slack): what aria_orb_kp_capacity() reports

    // single-frame host path
    uint8_t* d_img = nullptr;
    uint8_t* h_img = nullptr;          // pinned
    // outputs of the single-frame path: ONE device block and one pinned block, [16-int header][kp_cap keypoint records]
    // [kp_cap descriptor rows], so a frame comes back with one copy. Header: [0] count, [4] error bits, [5] slow-path
    // blocks, [6] rows needed (D.err points at header word 4 of the device block).
    uint8_t* d_out = nullptr;
    uint8_t* h_out = nullptr;          // pinned
    aria_keypoint* d_kps = nullptr;    // views into d_out / h_out
    uint8_t* d_desc = nullptr;
    int* d_count = nullptr;
    int* d_err_single = nullptr;       // deferred error words of the single-frame path: header words 4.. of d_out
    int* d_err_batch = nullptr;        // ... of the batch entry point: their own allocation, so that a single-frame call
                                       // between a batch call and aria_orb_check cannot wipe the batch's deferred bits
    aria_keypoint* h_kps = nullptr;
    uint8_t* h_desc = nullptr;
    int* h_count = nullptr;            // = header of h_out
    bool pending = false;
    // recorded behind the work of every single-frame extraction: the host waits for THIS, not for the whole stream, so that
    // work a caller queues behind an asynchronous extraction on a shared stream (aria_matcher_match_device_async) does not
    // delay aria_orb_sync
    hipEvent_t ev_done = nullptr;

    FrameSrc last_src{};
    bool have_last = false;
    Profiler prof;
    // single-frame latency path: the whole enqueue sequence (H2D, resize chain, 8 parallel FAST/blur branches, select,
    // describe, D2H) captured once per (plan, buffers) into a hipGraph and replayed
    // ARIA_HOST_TIMING=1: host-side time of the single-frame entry points by segment (printed when the handle is destroyed)
    double ht[4] = {0, 0, 0, 0};   // staging copy, enqueue/graph launch, wait, copy out
    long ht_n = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_w = 0, graph_h = 0, graph_nf = 0, graph_cap = 0, graph_q = 0;
    bool graph_failed = false;      // capture or instantiate failed once: stay on eager launches
    LaunchCtx ctx;          // per-handle launch state (function attributes of this device, side streams, stamp buffers)
};

namespace {

inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

void drop_graph(aria_orb_s* h) {
    if (h->graph_exec) hipGraphExecDestroy(h->graph_exec);
    if (h->graph) hipGraphDestroy(h->graph);
    h->graph_exec = nullptr;
    h->graph = nullptr;
}

void free_scratch(aria_orb_s* h) {
    drop_graph(h);              // the captured sequence points into the buffers freed below
    hipFree(h->D.raw); hipFree(h->D.blur); hipFree(h->D.cand);     // (D.cand_cnt lives behind D.ovf)
    hipFree(h->D.sel); hipFree(h->D.sel_cnt); hipFree(h->D.tab); hipFree(h->D.pyr_bands);    // (D.err lives inside d_out)
    hipFree(h->D.ovf); hipFree(h->D.ovf_items); hipFree(h->D.ovf_keys); hipFree(h->D.osel);
    hipFree(h->d_img); hipFree(h->d_out); hipFree(h->d_err_batch);
    h->d_err_batch = nullptr; h->d_err_single = nullptr;
    if (h->h_img) hipHostFree(h->h_img);
    if (h->h_out) hipHostFree(h->h_out);
    h->d_out = nullptr; h->h_out = nullptr;
    h->D = DeviceScratch{};
    h->d_img = nullptr; h->d_kps = nullptr; h->d_desc = nullptr; h->d_count = nullptr;
    h->h_img = nullptr; h->h_kps = nullptr; h->h_desc = nullptr; h->h_count = nullptr;
}

// Size everything for (max_w, max_h, max_features, max_batch). Level sizes are monotone in width/height, so a
// plan for any smaller image fits.
// Self-tuning of the band kernel's survivor queue: blocks that overflowed it took the (correct but ~20x slower) dense
// rescoring path; give the next calls a larger queue (at most 50 % of a workgroup's pixels).
void note_slow_blocks(aria_orb_s* h, int n) {
    if (n <= 0) return;
    h->slow_blocks += n;
    if (h->band_qpct0 < 50) h->band_qpct0 = std::min(50, h->band_qpct0 + 10);
}

int alloc_io(aria_orb_s* h, int rows);

int alloc_scratch(aria_orb_s* h) {
    Plan mp;
    int64_t tabn = plan_tab_entries(h->max_w, h->max_h) + 64;
    h->tab_host.assign((size_t)tabn, 0);
    int used = 0;
    int rc = build_plan(h->max_w, h->max_h, h->max_features, h->cand_cap_scale, h->tie_mode, &mp, h->tab_host.data(),
                        (int)tabn, &used, h->level_size_mode);
    if (rc != ARIA_OK) return rc;
    const size_t B = (size_t)h->max_batch;
    h->kp_cap = mp.sel_frame_entries;
    h->plan_rows = mp.sel_frame_entries;
    ARIA_HIP(hipMalloc(&h->D.raw, std::max<size_t>(mp.raw_frame_bytes * B, 256)));
    ARIA_HIP(hipMalloc(&h->D.blur, std::max<size_t>(mp.blur_frame_bytes * B, 256)));
    ARIA_HIP(hipMalloc(&h->D.cand, sizeof(uint32_t) * (size_t)mp.cand_frame_entries * B));
    // [4 tie-storm arena counters][8 candidate counters per frame]: zeroed together by one memset per pass
    ARIA_HIP(hipMalloc(&h->D.ovf, sizeof(int) * (4 + kLevels * B)));
    h->D.cand_cnt = h->D.ovf + 4;
    ARIA_HIP(hipMalloc(&h->D.sel, sizeof(uint4) * (size_t)mp.sel_frame_entries * B));
    ARIA_HIP(hipMalloc(&h->D.sel_cnt, sizeof(int) * kLevels * B));
    ARIA_HIP(hipMalloc(&h->D.tab, sizeof(uint32_t) * (size_t)tabn));
    h->bands_host.assign((size_t)((h->max_h + 7) / 8) * kLevels * 4 + 64, 0);
    ARIA_HIP(hipMalloc(&h->D.pyr_bands, sizeof(int) * h->bands_host.size()));
    // D.err ([0] deferred error bits, [1] band-kernel slow-path blocks, [2] rows needed by the largest frame that did not
    // fit kp_cap): part of the single-frame output block for that path (alloc_io below), d_err_batch for batches
    // tie-storm arenas: room for the worst case of four whole frames per pass (every FAST candidate of every level tied)
    {
        long long per_frame = 0;
        for (int l = 0; l < kLevels; l++) { long long np = 1; while (np < mp.lv[l].cand_cap) np <<= 1; per_frame += np; }
        h->D.ovf_keys_cap = 4 * per_frame;
        h->D.osel_cap = (int)std::min<long long>(4 * (long long)mp.cand_frame_entries + 64, 1ll << 24);
        ARIA_HIP(hipMalloc(&h->D.ovf_items, sizeof(int2) * kOvfItems));
        ARIA_HIP(hipMalloc(&h->D.ovf_keys, sizeof(unsigned long long) * (size_t)h->D.ovf_keys_cap));
        ARIA_HIP(hipMalloc(&h->D.osel, sizeof(uint4) * (size_t)h->D.osel_cap));
    }
    const size_t img_bytes = (size_t)align_up(h->max_w, 16) * h->max_h;
    ARIA_HIP(hipMalloc(&h->d_img, img_bytes));
    ARIA_HIP(hipHostMalloc(&h->h_img, img_bytes));
    rc = alloc_io(h, h->kp_cap);
    if (rc != ARIA_OK) return rc;
    ARIA_HIP(hipMalloc(&h->d_err_batch, 4 * sizeof(int)));
    ARIA_HIP(memset_on(h->stream, h->d_err_batch, 0, 4 * sizeof(int)));
    h->plan_valid = false;
    return ARIA_OK;
}

int ensure_plan(aria_orb_s* h, int w, int ht) {
    if (w > h->max_w || ht > h->max_h) return ARIA_E_TOO_LARGE;
    if (h->plan_valid && h->plan.width == w && h->plan.height == ht && h->plan.nfeatures == h->max_features) {
        h->plan.band_qpct0 = h->band_qpct0;
        return ARIA_OK;
    }
    int used = 0;
    int rc = build_plan(w, ht, h->max_features, h->cand_cap_scale, h->tie_mode, &h->plan, h->tab_host.data(),
                        (int)h->tab_host.size(), &used, h->level_size_mode);
    if (rc != ARIA_OK) return rc;
    h->plan.band_qpct0 = h->band_qpct0;
    if (used > 0)
        ARIA_HIP(hipMemcpyAsync(h->D.tab, h->tab_host.data(), sizeof(uint32_t) * (size_t)used, hipMemcpyHostToDevice,
                                h->stream));
    const int nb_ints = build_pyramid_bands(&h->plan, h->tab_host.data(), h->bands_host.data(), (int)h->bands_host.size());
    if (nb_ints < 0) return ARIA_E_INVALID;
    ARIA_HIP(hipMemcpyAsync(h->D.pyr_bands, h->bands_host.data(), sizeof(int) * (size_t)nb_ints, hipMemcpyHostToDevice,
                            h->stream));
    h->plan_valid = true;
    return ARIA_OK;
}

int launch_single(aria_orb_s* h);
static bool host_timing() { static const bool on = aria_getenv("ARIA_HOST_TIMING") != nullptr; return on; }

int enqueue_single(aria_orb_s* h, const uint8_t* image, int width, int height, int stride) {
    if (!image || stride < width) return ARIA_E_INVALID;
    int rc = ensure_plan(h, width, height);
    if (rc != ARIA_OK) return rc;
    const int pitch = align_up(width, 16);
    const bool timing = host_timing();
    const auto t0 = timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    if (stride == pitch) std::memcpy(h->h_img, image, (size_t)pitch * (height - 1) + (size_t)width);   // one call, not one per row
    else for (int y = 0; y < height; y++) std::memcpy(h->h_img + (size_t)y * pitch, image + (size_t)y * stride, (size_t)width);
    FrameSrc S{h->d_img, (int64_t)pitch * height, pitch, 1, (pitch % 16 == 0 && ((int64_t)pitch * height) % 16 == 0) ? 1 : 0};
    h->last_src = S;
    h->have_last = true;
    if (!timing) {
        rc = launch_single(h);
        if (rc == ARIA_OK) ARIA_HIP(hipEventRecord(h->ev_done, h->stream));
        return rc;
    }
    const auto t1 = std::chrono::steady_clock::now();
    rc = launch_single(h);
    if (rc == ARIA_OK) ARIA_HIP(hipEventRecord(h->ev_done, h->stream));
    const auto t2 = std::chrono::steady_clock::now();
    h->ht[0] += std::chrono::duration<double, std::micro>(t1 - t0).count();
    h->ht[1] += std::chrono::duration<double, std::micro>(t2 - t1).count();
    return rc;
}

// Internal single-frame output buffers (rows = kp_cap). A tie storm can return more keypoints than the plan's slots
// (OpenCV keeps every tie): the buffers then grow to what the frame needs and the frame is run again.
int alloc_io(aria_orb_s* h, int rows) {
    rows = (rows + 3) & ~3;
    hipFree(h->d_out);
    if (h->h_out) hipHostFree(h->h_out);
    h->d_out = nullptr; h->h_out = nullptr;
    h->kp_cap = rows;
    const size_t bytes = 64 + (size_t)rows * (sizeof(aria_keypoint) + 32);
    ARIA_HIP(hipMalloc(&h->d_out, bytes));
    ARIA_HIP(hipHostMalloc(&h->h_out, bytes));
    ARIA_HIP(memset_on(h->stream, h->d_out, 0, 64));
    std::memset(h->h_out, 0, 64);
    h->d_count = reinterpret_cast<int*>(h->d_out);
    h->d_err_single = reinterpret_cast<int*>(h->d_out) + 4;
    h->D.err = h->d_err_single;
    h->d_kps = reinterpret_cast<aria_keypoint*>(h->d_out + 64);
    h->d_desc = h->d_out + 64 + (size_t)rows * sizeof(aria_keypoint);
    h->h_count = reinterpret_cast<int*>(h->h_out);
    h->h_kps = reinterpret_cast<aria_keypoint*>(h->h_out + 64);
    h->h_desc = h->h_out + 64 + (size_t)rows * sizeof(aria_keypoint);
    return ARIA_OK;
}

int enqueue_single_ops(aria_orb_s* h);

// Replays (capturing it first when the plan or the buffers changed) the single-frame sequence as a hipGraph. Eager
// launches when profiling brackets are on (they synchronise the stream) or ARIA_SINGLE_GRAPH=0.
int launch_single(aria_orb_s* h) {
    static const bool want_graph = [] { const char* e = aria_getenv("ARIA_SINGLE_GRAPH"); return !(e && e[0] == '0'); }();
    const bool diag = env_config().stamp_level >= 0 || env_config().sel_stamps || env_config().desc_stamps;
    static const bool eager_latency = [] { const char* e = aria_getenv("ARIA_SINGLE_GRAPH"); return e && e[0] == 'e'; }();
    if (eager_latency && !h->prof.enabled && band_side_streams(h->ctx) == ARIA_OK) {
        h->ctx.schedule = 1;            // ARIA_SINGLE_GRAPH=eager: the latency schedule's launches without the graph
        return enqueue_single_ops(h);
    }
    if (diag && aria_getenv("ARIA_DIAG_LATENCY") && band_side_streams(h->ctx) == ARIA_OK) {
        h->ctx.schedule = 1;            // phase stamps of the latency schedule's kernels: same launches, eagerly
        return enqueue_single_ops(h);
    }
    if (!want_graph || h->graph_failed || h->prof.enabled || diag) { h->ctx.schedule = 0; return enqueue_single_ops(h); }
    if (band_side_streams(h->ctx) != ARIA_OK) { h->graph_failed = true; h->ctx.schedule = 0; return enqueue_single_ops(h); }
    const bool stale = !h->graph_exec || h->graph_w != h->plan.width || h->graph_h != h->plan.height ||
                       h->graph_nf != h->max_features || h->graph_cap != h->kp_cap || h->graph_q != h->plan.band_qpct0;
    if (stale) {
        drop_graph(h);
        h->ctx.schedule = 1;
        // whatever is queued on the stream (plan tables of ensure_plan) must be done before the capture starts
        ARIA_HIP(hipStreamSynchronize(h->stream));
        hipError_t e = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal);
        int rc = ARIA_OK;
        if (e == hipSuccess) {
            rc = enqueue_single_ops(h);
            e = hipStreamEndCapture(h->stream, &h->graph);
        }
        if (e == hipSuccess && rc == ARIA_OK) e = hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0);
        if (e != hipSuccess || rc != ARIA_OK || !h->graph_exec) {
            drop_graph(h);
            (void)hipGetLastError();
            h->graph_failed = true;
            h->ctx.schedule = 0;
            return enqueue_single_ops(h);
        }
        h->graph_w = h->plan.width; h->graph_h = h->plan.height; h->graph_nf = h->max_features; h->graph_cap = h->kp_cap;
        h->graph_q = h->plan.band_qpct0;
    }
    ARIA_HIP(hipGraphLaunch(h->graph_exec, h->stream));
    return ARIA_OK;
}

int enqueue_single_ops(aria_orb_s* h) {
    const size_t img_bytes = (size_t)h->last_src.row_stride * h->plan.height;
    h->ctx.hdr = reinterpret_cast<int*>(h->d_out);               // count + the deferred error words of this frame: zeroed by the pass
    h->D.err = h->d_err_single;
    h->ctx.host_img = h->h_img;
    if (!latency_zero_copy(h->plan, h->ctx, &h->prof))          // else the pyramid kernel pulls the frame from h_img itself
        ARIA_HIP(hipMemcpyAsync(h->d_img, h->h_img, img_bytes, hipMemcpyHostToDevice, h->stream));
    launch_extract_chunk(h->plan, h->last_src, h->D, 1, h->d_kps, h->d_desc, h->d_count, h->kp_cap, h->stream, &h->prof, h->ctx);
    h->ctx.hdr = nullptr;
    h->ctx.host_img = nullptr;
    ARIA_HIP(hipGetLastError());
    // the whole result in one copy: header (count, error words) + keypoints + descriptors
    ARIA_HIP(hipMemcpyAsync(h->h_out, h->d_out, 64 + (size_t)h->kp_cap * (sizeof(aria_keypoint) + 32), hipMemcpyDeviceToHost, h->stream));
    return ARIA_OK;
}

int finish_single(aria_orb_s* h, aria_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
    const bool timing = host_timing();
    const auto t0 = timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    ARIA_HIP(hipEventSynchronize(h->ev_done));
    const auto t1 = timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    struct Acc {
        aria_orb_s* h; bool on; std::chrono::steady_clock::time_point a, b;
        ~Acc() {
            if (!on) return;
            const auto c = std::chrono::steady_clock::now();
            h->ht[2] += std::chrono::duration<double, std::micro>(b - a).count();
            h->ht[3] += std::chrono::duration<double, std::micro>(c - b).count();
            h->ht_n++;
        }
    } acc{h, timing, t0, t1};
    if ((h->h_count[4] & ERRBIT_KPCAP) && h->h_count[6] > h->kp_cap) {
        // the frame has more keypoints than the internal buffers hold (ties): grow them and run the frame again
        // (its image is still in the handle's device copy)
        const int need = (h->h_count[6] + 63) & ~63;
        int rc = alloc_io(h, need);
        if (rc != ARIA_OK) return rc;
        rc = launch_single(h);
        if (rc != ARIA_OK) return rc;
        ARIA_HIP(hipStreamSynchronize(h->stream));
    }
    const int errbits = h->h_count[4];
    note_slow_blocks(h, h->h_count[5]);
    int st = errbits_to_status(errbits & ~ERRBIT_KPCAP);
    if (st != ARIA_OK) { if (n_out) *n_out = 0; return st; }
    const int n = (errbits & ERRBIT_KPCAP) ? std::max(h->h_count[6], h->h_count[0]) : h->h_count[0];
    h->last_n = n;
    if (n_out) *n_out = n;
    if (n > cap) return ARIA_E_OUTPUT_TOO_SMALL;
    if (n > 0) {
        if (!kps || !desc) return ARIA_E_INVALID;
        std::memcpy(kps, h->h_kps, sizeof(aria_keypoint) * (size_t)n);
        std::memcpy(desc, h->h_desc, 32 * (size_t)n);
    }
    return ARIA_OK;
}

}  // namespace

extern "C" {

const char* aria_status_string(int s) {
    switch (s) {
        case ARIA_OK: return "ok";
        case ARIA_E_INVALID: return "invalid argument";
        case ARIA_E_NO_DEVICE: return "no usable HIP device";
        case ARIA_E_HIP: return "HIP runtime call failed";
        case ARIA_E_KERNEL: return "kernel launch failed or faulted";
        case ARIA_E_OOM: return "out of memory";
        case ARIA_E_TOO_LARGE: return "image larger than the handle was created for";
        case ARIA_E_OUTPUT_TOO_SMALL: return "output capacity too small";
        case ARIA_E_OVERFLOW: return "internal candidate/sort buffer overflow";
        case ARIA_E_BUSY: return "an asynchronous extract is already pending";
        case ARIA_E_NOT_PENDING: return "no asynchronous extract pending";
        default: return "unknown status";
    }
}

int aria_abi_version(void) { return ARIA_ORB_HIP_ABI_VERSION; }
const char* aria_last_hip_error(void) { return last_hip_error_buf(); }

void aria_orb_default_config(aria_orb_config* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->struct_size = (int)sizeof(aria_orb_config);
    c->device = 0;
    c->stream = nullptr;
    c->max_width = 640;
    c->max_height = 480;
    c->max_features = 1000;   // reference include/adapters/gpu/OrbCudaExtractor.hpp:12
    c->max_batch = 1;
    c->blur_tie_mode = 1;
    c->cand_cap_scale = 0;
    c->level_size_mode = 0;
}

int aria_orb_create(const aria_orb_config* c, aria_orb_t* out) {
    if (!c || !out || c->struct_size != (int)sizeof(aria_orb_config)) return ARIA_E_INVALID;
    if (c->max_width < 16 || c->max_height < 16 || c->max_width > kMaxDim || c->max_height > kMaxDim ||
        c->max_features < 0 || c->max_features > 65536 || c->max_batch < 1)
        return ARIA_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    ARIA_HIP(hipGetDeviceCount(&ndev));
    if (c->device < 0 || c->device >= ndev) {
        std::snprintf(last_hip_error_buf(), 256, "device %d not present (%d devices)", c->device, ndev);
        return ARIA_E_NO_DEVICE;
    }
    ARIA_HIP(hipSetDevice(c->device));
    aria_orb_s* h = new (std::nothrow) aria_orb_s();
    if (!h) return ARIA_E_OOM;
    h->device = c->device;
    h->max_w = c->max_width;
    h->max_h = c->max_height;
    h->max_features = c->max_features;
    h->max_batch = c->max_batch;
    if (c->blur_tie_mode < 0 || c->blur_tie_mode > 3 || c->level_size_mode < 0 || c->level_size_mode > 1) { delete h; return ARIA_E_INVALID; }
    h->tie_mode = c->blur_tie_mode;
    h->level_size_mode = c->level_size_mode;
    h->cand_cap_scale = c->cand_cap_scale > 0 ? c->cand_cap_scale : 0;
    if (c->stream) {
        h->stream = (hipStream_t)c->stream;
    } else {
        hipError_t e = create_stream(&h->stream);
        if (e != hipSuccess) { delete h; return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
        h->owns_stream = true;
    }
    int rc = h->ctx.init(c->device);
    if (rc == ARIA_OK && hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming) != hipSuccess) rc = ARIA_E_HIP;
    if (rc == ARIA_OK) rc = alloc_scratch(h);
    if (rc != ARIA_OK) { aria_orb_destroy(h); return rc; }
    *out = h;
    return ARIA_OK;
}

void aria_orb_destroy(aria_orb_t h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->ht_n > 0)
        fprintf(stderr, "[aria host timing] %ld single-frame calls, us per call: staging copy %.1f, enqueue %.1f, wait %.1f, copy out %.1f\n",
                h->ht_n, h->ht[0] / h->ht_n, h->ht[1] / h->ht_n, h->ht[2] / h->ht_n, h->ht[3] / h->ht_n);
    h->prof.release();
    if (h->ev_done) hipEventDestroy(h->ev_done);
    drop_graph(h);
    h->ctx.release();
    free_scratch(h);
    if (h->owns_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int aria_orb_set_max_features(aria_orb_t h, int n) {
    if (!h || n < 0 || n > 65536) return ARIA_E_INVALID;
    if (h->pending) return ARIA_E_BUSY;
    if (n == h->max_features) return ARIA_OK;
    ARIA_HIP(hipSetDevice(h->device));
    ARIA_HIP(hipStreamSynchronize(h->stream));
    const int old = h->max_features;
    free_scratch(h);
    h->max_features = n;
    int rc = alloc_scratch(h);
    if (rc != ARIA_OK) {
        free_scratch(h);
        h->max_features = old;
        alloc_scratch(h);
    }
    h->have_last = false;
    return rc;
}

int aria_orb_get_max_features(aria_orb_t h) { return h ? h->max_features : ARIA_E_INVALID; }
int aria_orb_kp_capacity(aria_orb_t h) { return h ? h->plan_rows : ARIA_E_INVALID; }
int aria_orb_rows_needed(aria_orb_t h) { return h ? h->rows_needed : ARIA_E_INVALID; }
void* aria_orb_stream(aria_orb_t h) { return h ? (void*)h->stream : nullptr; }

int aria_orb_extract(aria_orb_t h, const uint8_t* image, int width, int height, int stride,
                     aria_keypoint* keypoints, uint8_t* descriptors, int cap, int* n_out) {
    if (!h || cap < 0) return ARIA_E_INVALID;
    if (h->pending) return ARIA_E_BUSY;
    ARIA_HIP(hipSetDevice(h->device));
    int rc = enqueue_single(h, image, width, height, stride);
    if (rc != ARIA_OK) return rc;
    return finish_single(h, keypoints, descriptors, cap, n_out);
}

int aria_orb_extract_async(aria_orb_t h, const uint8_t* image, int width, int height, int stride) {
    if (!h) return ARIA_E_INVALID;
    if (h->pending) return ARIA_E_BUSY;
    ARIA_HIP(hipSetDevice(h->device));
    int rc = enqueue_single(h, image, width, height, stride);
    if (rc == ARIA_OK) h->pending = true;
    return rc;
}

int aria_orb_sync(aria_orb_t h, aria_keypoint* keypoints, uint8_t* descriptors, int cap, int* n_out) {
    if (!h || cap < 0) return ARIA_E_INVALID;
    if (!h->pending) return ARIA_E_NOT_PENDING;
    ARIA_HIP(hipSetDevice(h->device));
    h->pending = false;
    return finish_single(h, keypoints, descriptors, cap, n_out);
}

int aria_orb_fetch_last(aria_orb_t h, aria_keypoint* keypoints, uint8_t* descriptors, int cap, int* n_out) {
    if (!h || cap < 0) return ARIA_E_INVALID;
    if (h->pending) return ARIA_E_BUSY;
    if (h->last_n < 0) return ARIA_E_NOT_PENDING;
    if (n_out) *n_out = h->last_n;
    if (h->last_n > cap) return ARIA_E_OUTPUT_TOO_SMALL;
    if (h->last_n > 0) {
        if (!keypoints || !descriptors) return ARIA_E_INVALID;
        std::memcpy(keypoints, h->h_kps, sizeof(aria_keypoint) * (size_t)h->last_n);
        std::memcpy(descriptors, h->h_desc, 32 * (size_t)h->last_n);
    }
    return ARIA_OK;
}

// The role OrbCudaExtractor::getGpuDescriptors() plays (include/adapters/gpu/OrbCudaExtractor.hpp:34-35): where the
// single-frame result lies on the device.
int aria_orb_last_device(aria_orb_t h, const aria_keypoint** d_keypoints, const uint8_t** d_descriptors, const int** d_count,
                         int* n, int* rows) {
    if (!h) return ARIA_E_INVALID;
    if (!h->d_out) return ARIA_E_NOT_PENDING;
    if (d_keypoints) *d_keypoints = h->d_kps;
    if (d_descriptors) *d_descriptors = h->d_desc;
    if (d_count) *d_count = h->d_count;
    if (n) *n = h->pending ? -1 : h->last_n;
    if (rows) *rows = h->kp_cap;
    return ARIA_OK;
}

int aria_orb_extract_batch_device(aria_orb_t h, const uint8_t* d_images, int n_frames, int width, int height,
                                  int64_t frame_stride, int row_stride, aria_keypoint* d_keypoints,
                                  uint8_t* d_descriptors, int* d_counts, int kp_cap) {
    if (!h || !d_images || !d_keypoints || !d_descriptors || !d_counts || n_frames < 0 || kp_cap < 1 ||
        row_stride < width || frame_stride < (int64_t)row_stride * (height - 1) + width)
        return ARIA_E_INVALID;
    if (h->pending) return ARIA_E_BUSY;
    ARIA_HIP(hipSetDevice(h->device));
    int rc = ensure_plan(h, width, height);
    if (rc != ARIA_OK) return rc;
    const int aligned4 = (((uintptr_t)d_images | (uintptr_t)frame_stride | (uintptr_t)row_stride) & 3) == 0;
    const int aligned16 = (((uintptr_t)d_images | (uintptr_t)frame_stride | (uintptr_t)row_stride) & 15) == 0;
    h->ctx.schedule = 0;        // batches use the throughput schedule
    h->D.err = h->d_err_batch;
    for (int f0 = 0; f0 < n_frames; f0 += h->max_batch) {
        const int nf = std::min(h->max_batch, n_frames - f0);
        FrameSrc S{d_images + (int64_t)f0 * frame_stride, frame_stride, row_stride, aligned4, aligned16};
        if (f0 == 0) { h->last_src = S; h->have_last = true; }
        h->ctx.stage_events_armed = f0 + nf >= n_frames;      // aria_orb_set_stage_event: last pass only
        launch_extract_chunk(h->plan, S, h->D, nf, d_keypoints + (int64_t)f0 * kp_cap,
                             d_descriptors + (int64_t)f0 * kp_cap * 32, d_counts + f0, kp_cap, h->stream, &h->prof, h->ctx);
        h->ctx.stage_events_armed = false;
    }
    ARIA_HIP(hipGetLastError());
    return ARIA_OK;
}

int aria_orb_check(aria_orb_t h) {
    if (!h) return ARIA_E_INVALID;
    ARIA_HIP(hipSetDevice(h->device));
    ARIA_HIP(hipStreamSynchronize(h->stream));
    int two[4] = {0, 0, 0, 0};
    ARIA_HIP(memcpy_on(h->stream, two, h->d_err_batch, 4 * sizeof(int), hipMemcpyDeviceToHost));
    if (two[0] || two[1] || two[2]) ARIA_HIP(memset_on(h->stream, h->d_err_batch, 0, 4 * sizeof(int)));
    h->rows_needed = two[2];
    note_slow_blocks(h, two[1]);
    return errbits_to_status(two[0]);
}

long long aria_orb_slow_path_blocks(aria_orb_t h, int reset) {
    if (!h) return ARIA_E_INVALID;
    const long long n = h->slow_blocks;
    if (reset) h->slow_blocks = 0;
    return n;
}

const char* aria_orb_fast_blur_kernel(aria_orb_t h) { return h ? h->ctx.last_fast_blur : ""; }

int aria_orb_set_stage_event(aria_orb_t h, int stage, void* event) {
    if (!h || stage < 0 || stage >= ARIA_ORB_STAGES) return ARIA_E_INVALID;
    if (stage != STAGE_SELECT) return ARIA_E_INVALID;       // the one point a caller has asked for so far
    h->ctx.stage_event[stage] = static_cast<hipEvent_t>(event);
    return ARIA_OK;
}

int aria_orb_set_profiling(aria_orb_t h, int enable) {
    if (!h) return ARIA_E_INVALID;
    h->prof.enabled = enable != 0;
    h->prof.stage_mask = (enable & 1) ? ~0u : ((unsigned)enable >> 1);    // 1: every stage; else bit (s + 1) selects stage s
    return ARIA_OK;
}

int aria_orb_get_profile(aria_orb_t h, int reset, double* stage_ms, int64_t* stage_launches, int64_t* frames) {
    if (!h) return ARIA_E_INVALID;
    ARIA_HIP(hipSetDevice(h->device));
    ARIA_HIP(hipStreamSynchronize(h->stream));
    h->prof.collect();
    for (int s = 0; s < STAGE_COUNT; s++) {
        if (stage_ms) stage_ms[s] = h->prof.ms[s];
        if (stage_launches) stage_launches[s] = h->prof.launches[s];
    }
    if (frames) *frames = h->prof.frames;
    if (reset) {
        for (int s = 0; s < STAGE_COUNT; s++) { h->prof.ms[s] = 0; h->prof.launches[s] = 0; }
        h->prof.frames = 0;
    }
    return ARIA_OK;
}

int aria_orb_level_info(int max_features, int width, int height, int level, int* lw, int* lh, int* quota, float* scale) {
    if (level < 0 || level >= kLevels) return ARIA_E_INVALID;
    Plan p;
    std::vector<uint32_t> tab((size_t)plan_tab_entries(width, height) + 64);
    int used = 0;
    int rc = build_plan(width, height, max_features, 0, 1, &p, tab.data(), (int)tab.size(), &used);
    if (rc != ARIA_OK) return rc;
    if (lw) *lw = p.lv[level].w;
    if (lh) *lh = p.lv[level].h;
    if (quota) *quota = p.lv[level].quota;
    if (scale) *scale = p.lv[level].scale;
    return ARIA_OK;
}

int aria_orb_resize_table(int width, int height, int level, int axis, uint32_t* out, int cap) {
    if (level < 1 || level >= kLevels || !out || (axis != 0 && axis != 1)) return ARIA_E_INVALID;
    Plan p;
    std::vector<uint32_t> tab((size_t)plan_tab_entries(width, height) + 64);
    int used = 0;
    int rc = build_plan(width, height, 1000, 0, 1, &p, tab.data(), (int)tab.size(), &used);
    if (rc != ARIA_OK) return rc;
    const int n = axis == 0 ? p.lv[level].w : p.lv[level].h;
    if (n > cap) return ARIA_E_OUTPUT_TOO_SMALL;
    std::memcpy(out, tab.data() + (axis == 0 ? p.lv[level].xtab : p.lv[level].ytab), sizeof(uint32_t) * (size_t)n);
    return n;
}

int aria_orb_debug_read_level(aria_orb_t h, int level, int blurred, uint8_t* host_out) {
    if (!h || !host_out || level < 0 || level >= kLevels || !h->plan_valid || !h->have_last) return ARIA_E_INVALID;
    ARIA_HIP(hipSetDevice(h->device));
    ARIA_HIP(hipStreamSynchronize(h->stream));
    const LevelGeom& g = h->plan.lv[level];
    const uint8_t* src;
    size_t spitch;
    if (blurred && h->ctx.last_blur_q4) {
        // the batch path keeps blurred levels in Q4 order (orb_device.h): fetch the row quads, put the rows back in order
        const int hq = (g.h + 3) & ~3;
        std::vector<uint8_t> tmp((size_t)g.pitch * hq);
        ARIA_HIP(memcpy_on(h->stream, tmp.data(), h->D.blur + g.blur_off, tmp.size(), hipMemcpyDeviceToHost));
        for (int y = 0; y < g.h; y++)
            for (int x = 0; x < g.w; x++) host_out[(size_t)y * g.w + x] = tmp[(size_t)q4_offset(x, y, g.pitch)];
        return ARIA_OK;
    }
    if (blurred) { src = h->D.blur + g.blur_off; spitch = (size_t)g.pitch; }
    else if (level == 0) { src = h->last_src.img; spitch = (size_t)h->last_src.row_stride; }
    else { src = h->D.raw + g.raw_off; spitch = (size_t)g.pitch; }
    ARIA_HIP(hipMemcpy2DAsync(host_out, (size_t)g.w, src, spitch, (size_t)g.w, (size_t)g.h, hipMemcpyDeviceToHost, h->stream));
    ARIA_HIP(hipStreamSynchronize(h->stream));
    return ARIA_OK;
}

int aria_orb_pyramid_bands(int width, int height, int* out, int cap, int* band_rows, int* lds_bytes) {
    if (!out) return ARIA_E_INVALID;
    Plan p;
    std::vector<uint32_t> tab((size_t)plan_tab_entries(width, height) + 64);
    int used = 0;
    int rc = build_plan(width, height, 1000, 0, 1, &p, tab.data(), (int)tab.size(), &used);
    if (rc != ARIA_OK) return rc;
    const int n = build_pyramid_bands(&p, tab.data(), out, cap);
    if (n < 0) return ARIA_E_OUTPUT_TOO_SMALL;
    if (band_rows) *band_rows = p.pyr_bh;
    if (lds_bytes) *lds_bytes = p.pyr_lds_bytes;
    return n;
}

int aria_orb_algorithmic_bytes(int width, int height, int n_keypoints, int64_t* b_extract, int64_t* b_fused) {
    Plan p;
    std::vector<uint32_t> tab((size_t)plan_tab_entries(width, height) + 64);
    int used = 0;
    int rc = build_plan(width, height, 1000, 0, 1, &p, tab.data(), (int)tab.size(), &used);
    if (rc != ARIA_OK) return rc;
    const int64_t P = p.pixels_total;
    const int64_t p0 = (int64_t)p.lv[0].w * p.lv[0].h, p7 = (int64_t)p.lv[kLevels - 1].w * p.lv[kLevels - 1].h;
    if (b_extract) *b_extract = 5 * P - p0 - p7 + 56ll * n_keypoints;
    if (b_fused) *b_fused = 2 * P + 56ll * n_keypoints;
    return ARIA_OK;
}

}  // extern "C"
