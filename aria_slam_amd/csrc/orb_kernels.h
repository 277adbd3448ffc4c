// Kernel-launch interface between the C-ABI host code (orb_api.hip) and the kernels (orb_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <vector>

#include "aria_orb_hip.h"
#include "orb_plan.h"

namespace aria {

// Where level 0 lives: the caller's images, read in place.
struct FrameSrc {
    const uint8_t* img;     // frame 0 of the chunk
    int64_t frame_stride;   // bytes between frames
    int row_stride;         // bytes between rows
    int aligned4;           // base, frame_stride and row_stride all multiples of 4 -> dword loads allowed
    int aligned16;          // ... all multiples of 16 -> 16-byte loads allowed
};

// Per-chunk device scratch owned by the extractor handle (HBM layout: DESIGN.md "Data layout").
struct DeviceScratch {
    uint8_t* raw;        // [max_batch][raw_frame_bytes]   pyramid levels 1..7, un-blurred
    uint8_t* blur;       // [max_batch][blur_frame_bytes]  pyramid levels 0..7, blurred
    uint32_t* cand;      // [max_batch][cand_frame_entries] FAST candidates x:11|y:11|score:8
    int* cand_cnt;       // [max_batch][8]
    uint4* sel;          // [max_batch][sel_frame_entries]  (x | y<<16, harris bits, rank in the canonical order of
                         //   the level, -), stored per level in 64 x 64-px tile order (k_describe's cache locality)
    int* sel_cnt;        // [max_batch][8]
    uint32_t* tab;       // resize coefficient tables (ofs | c1 << 16)
    int* pyr_bands;      // [pyr_nbands][8][4] row ranges of the fused pyramid kernel
    int* err;            // deferred error bits
    // tie-storm fallback (select_ovf_item): a key arena for the global-memory sort of (frame, level) pairs whose ties
    // overflow k_select's LDS capacities, and an arena of selected keypoints beyond a level's regular slots
    int* ovf;            // [0] work-list items, [1] osel entries used, [2..3] key-arena entries used (64-bit)
    int2* ovf_items;     // [kOvfItems] (frame, level) pairs k_select hands to k_select_ovf
    unsigned long long* ovf_keys;
    long long ovf_keys_cap;
    uint4* osel;
    int osel_cap;
};

// Optional per-stage timing (bench.py's roofline figure). HIP event pairs bracket each stage on the launch stream;
// the stream is drained before the start event so the bracket holds exactly that stage's kernels (event markers
// placed between back-to-back dependent launches do not bracket kernel execution precisely on this stack: their
// intervals overlap and sum to more than the wall clock). Costs one stream sync per stage per pass (< 2 % at
// 1024-frame passes); off by default.
enum { STAGE_RESIZE = 0, STAGE_FAST_BLUR = 1, STAGE_SELECT = 2, STAGE_DESCRIBE = 3, STAGE_COUNT = 4 };
struct LaunchEvents { hipEvent_t start, stop; int stage; int launches; };
struct Profiler {
    bool enabled = false;
    unsigned stage_mask = ~0u;            // bit s: bracket stage s (every bracket drains the stream twice: ~20 us of idle GPU)
    bool open = false;                    // a bracket is open
    std::vector<LaunchEvents> pending;    // recorded, not yet read
    std::vector<hipEvent_t> pool;         // recycled events
    double ms[STAGE_COUNT] = {0, 0, 0, 0};
    int64_t launches[STAGE_COUNT] = {0, 0, 0, 0};
    int64_t frames = 0;
    LaunchEvents cur{};
    hipEvent_t get();
    void begin(int stage, hipStream_t st);   // drain the stream, record the start event
    void count() { cur.launches++; }
    void end(hipStream_t st);                // record the stop event
    void collect();                          // caller has synchronised the stream
    void release();
};

// Per-handle launch state. Everything a launcher needs besides the plan and the scratch lives here, so two handles
// (two host threads, two devices, two streams) never share mutable state: the kernels' function attributes are set
// for the handle's device when the handle is created, and the optional side streams / diagnostic stamp buffers
// belong to the handle. The environment is read once per process into an immutable EnvConfig.
struct EnvConfig {
    int ablate;            // ARIA_ABLATE (timing experiments: results invalid), honoured by diagnostic builds only
    int level_streams;     // ARIA_LEVEL_STREAMS=1: one side stream per level
    int stamp_level;       // ARIA_STAMPS=<level>: phase stamps of the band kernel, -1 = off
    int sel_stamps;        // ARIA_SEL_STAMPS=1
    int band_trim;         // ARIA_BAND_TRIM=0: no trimming of blur-only right-edge columns to whole waves (band_lanes)
    int band_xcd_map;      // ARIA_BAND_XCD_MAP=0: plain (strip, frame) grid order in the batch FAST/blur launches
    int select_bitonic;    // ARIA_SELECT_SORT=bitonic: k_select always takes its LDS bitonic sort (default: histogram bins + in-bin ranks)
    int desc_stamps;       // ARIA_DESC_STAMPS=1
    int fast_blur_impl;    // 2 = band kernel, all VALU (fast_blur_band.hip, default), 1 = band kernel with the blur on the
                           // matrix cores (band_mfma.hip, ARIA_FAST_BLUR_IMPL=mfma: same bits, 23 % fewer VALU instructions,
                           // same time -- DESIGN.md section 4), 0 = 64x32 LDS tiles (ARIA_FAST_BLUR_IMPL=tile)
    int fuse_resize;       // pyramid step fused into the FAST/blur launches (default 1)
    int pyr_impl;          // 1 = fused in-LDS pyramid (ARIA_PYRAMID_IMPL=fused)
    int rs_impl;           // stand-alone resize pass: 2 dot2 LDS bands, 1 shift/mad LDS bands, 0 direct gathers
    int band_budget_kb, band_qpct0, band_qstep;   // ARIA_BAND_*: <0 / 0 = plan defaults
    int batch_stream;      // batches take the streaming FAST/blur kernel (fast_blur_stream.hip) where the plan allows it
                           // (default 1; ARIA_FAST_BLUR_IMPL=band keeps the band kernel)
};
const EnvConfig& env_config();

struct LaunchCtx {
    int device = -1;
    // 0: batch schedule (pyramid fused into the FAST/blur launches, everything on one stream: best throughput);
    // 1: latency schedule of the single-frame host path: stand-alone resize chain, then the 8 levels' FAST/blur launches
    //    as parallel branches (side streams; captured into a hipGraph by orb_api.hip), because with one frame a level's
    //    launch is a handful of workgroups and the chain of 8 dependent launches is what the caller waits for
    int schedule = 0;
    // single-frame host path: the 16-int result header (count, deferred error words) that launch_extract_chunk has to zero
    // before the first kernel that writes it (nullptr: the caller's business). In the latency schedule the pyramid kernel
    // does it together with the pass counters, which saves two fill nodes of the graph.
    int* hdr = nullptr;
    // single-frame latency schedule: the frame as it lies in the handle's pinned staging buffer. The pyramid kernel then
    // reads level 0 from there (over the host link) and writes the device copy the later kernels use, so the graph has
    // no upload node (latency_zero_copy() says whether launch_extract_chunk will take that route).
    const uint8_t* host_img = nullptr;
    // caller's events recorded on the stream right before a stage of a pass (aria_orb_set_stage_event); armed by the batch
    // entry point for its last pass only
    hipEvent_t stage_event[4] = {};
    bool stage_events_armed = false;
    bool last_blur_q4 = false;         // the most recent pass left the blurred levels in Q4 order (orb_device.h): k_fast_blur_stream does
    const char* last_fast_blur = "";   // name of the FAST/blur kernel the most recent pass launched (aria_orb_fast_blur_kernel)
    hipStream_t side[kLevels] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kLevels] = {}, ev_lvl[kLevels] = {};
    unsigned long long* d_band_stamps = nullptr;
    unsigned long long* d_sel_stamps = nullptr;
    unsigned long long* d_desc_stamps = nullptr;
    int init(int device);      // current device = `device`; returns an aria_status
    void release();
};

// Launch `kernel`; counted into the open stage when profiling is on.
#define ARIA_LAUNCH(prof, kernel, grid, block, lds, st, ...)                       \
    do {                                                                           \
        if ((prof) && (prof)->open) (prof)->count();                               \
        hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);             \
    } while (0)

// band implementation of the FAST+NMS+blur stage (fast_blur_band.hip): 8 launches, one per level
// fuse_resize: every level's launch also writes the raw rows of the next level (the caller then skips the resize pass;
// the launches must stay in level order on one stream)
void launch_fast_blur_band(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st,
                           Profiler* prof, bool fuse_resize, LaunchCtx& ctx);
// round-2 band kernel (band_mfma.hip): same contract as launch_fast_blur_band
void launch_band2(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st, Profiler* prof,
                  bool fuse_resize, LaunchCtx& ctx);
int band2_set_attributes();
// streaming form of the FAST/blur stage for batches (fast_blur_stream.hip): 8 launches in level order on one stream, each
// also writing the raw rows of the next level; same contract and same bits as launch_fast_blur_band with fuse_resize
void launch_fast_blur_stream(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st,
                             Profiler* prof, LaunchCtx& ctx);
bool stream_eligible(const Plan& P, const FrameSrc& S);
int stream_set_attributes();
int blur_tail_start(int tie_mode, int w);
int band_set_attributes();          // hipFuncSetAttribute of the band kernels on the current device (aria_status)
int band_init_ctx(LaunchCtx& ctx);  // side streams / stamp buffer when the environment asks for them
int band_side_streams(LaunchCtx& ctx);   // create the per-level side streams + fork/join events (idempotent)

// stand-alone pyramid pass (pyramid_pass.hip) and the tile form of the FAST/blur stage (fast_blur_tile.hip)
// strips of all levels in one launch of the band kernel (single-frame latency schedule): first[l] = first blockIdx.x of level l
// xcd_map (per-level batch launches): 1 = the strips of a frame all go to one XCD (see k_fast_blur_band)
// lpr[l]: lanes per row of level l in the walk (band_lanes(): the host decides, the kernel follows)
struct BandAll { int first[kLevels + 1]; int qcap[kLevels]; int xcd_map; int lpr[kLevels]; };
bool launch_pyramid_fused(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st, Profiler* prof,
                          int* clr_a = nullptr, int n_a = 0, int* clr_b = nullptr, int n_b = 0,
                          const uint8_t* src_alt = nullptr, uint8_t* copy_dst = nullptr);
bool pyramid_fused_available(const Plan& P);
// true when a single-frame pass with this context takes the latency schedule AND its pyramid kernel can pull the frame
// from ctx.host_img itself (the caller then skips the upload)
bool latency_zero_copy(const Plan& P, const LaunchCtx& ctx, const Profiler* prof);
void launch_pyramid_level(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st, Profiler* prof, int l);
void launch_pyramid_and_band_latency(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st,
                                     Profiler* prof, LaunchCtx& ctx);
void launch_pyramid_pass(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st, Profiler* prof);
int pyramid_set_attributes();
void launch_fast_blur_tile(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames, hipStream_t st, Profiler* prof);

void launch_extract_chunk(const Plan& P, const FrameSrc& S, const DeviceScratch& D, int n_frames,
                          aria_keypoint* d_kps, uint8_t* d_desc, int* d_counts, int kp_cap, hipStream_t st,
                          Profiler* prof, LaunchCtx& ctx);

}  // namespace aria
