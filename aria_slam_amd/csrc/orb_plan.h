// Per-(width,height,nfeatures) execution plan shared by host code and kernels (passed by value as a kernel
// argument). Geometry follows CPU cv::ORB (OpenCV 4.9.0 orb.cpp detectAndCompute / computeKeyPoints) with the
// parameters the reference fixes at src/adapters/gpu/OrbCudaExtractor.cpp:35-45.
#pragma once
#include <cstdint>

namespace aria {

constexpr int kLevels = 8;            // OrbCudaExtractor.cpp:38
constexpr float kScaleFactor = 1.2f;  // :37
constexpr int kEdgeThreshold = 31;    // :39
constexpr int kPatchSize = 31;        // :43
constexpr int kFastThreshold = 20;    // :44
constexpr int kHalfPatch = 15;

// FAST + blur tile (one 256-thread workgroup): 64 x 32 output pixels, 4-px halo.
constexpr int kTileW = 64;
constexpr int kTileH = 32;
constexpr int kHalo = 4;
constexpr int kPatchW = kTileW + 2 * kHalo;  // 72
constexpr int kPatchH = kTileH + 2 * kHalo;  // 40

constexpr int kSortCapMin = 4096;    // LDS bitonic sort capacity of the per-(frame,level) selection kernel (entries of 8 B)
constexpr int kSortCapMax = 16384;   // 128 KB of the CU's 160 KB LDS
constexpr int kSelSlack = 64;    // extra output slots per level for ties at the Harris cut
constexpr int kOvfItems = 4096;  // (frame, level) pairs per pass whose ties overflow the LDS selection (k_select_ovf's work list)
constexpr int kPyrYSlice = 96;   // max rows a pyramid band computes per level
constexpr int kMaxDim = 2047;    // candidate packing: x:11 | y:11 | score:8

struct LevelGeom {
    int w, h;          // level size in pixels
    int pitch;         // row pitch of this level in the raw / blurred scratch (multiple of 16)
    int quota;         // nfeaturesPerLevel
    int cand_cap;      // FAST candidate list capacity (entries)
    int cand_off;      // entry offset of the level's list inside a frame's candidate block
    int sel_cap;       // selected-keypoint capacity (quota + kSelSlack)
    int sel_off;       // entry offset inside a frame's selection block
    int tiles_x;       // FAST/blur tiles per row
    int tile_base;     // first tile id of this level in the all-level tile enumeration
    int xtab, ytab;    // offsets (in uint32) of the resize coefficient tables (level >= 1)
    float scale;       // layerScale
    int xinv;          // offset (in uint32) of the streaming kernel's host table (level >= 1): entry g = the dword of THIS level's
                       // row that source dword g of the level above hosts in the fused pyramid step, or 0xFFFFFFFF
    int ytr;           // offset (in uint32) of the pyramid step's table BY SOURCE ROW (level >= 1; one entry per row of the level
                       // above): row t is the LOWER source row of output row dy -> dy | cy1 << 16 | 1 << 31, else 0
    int64_t raw_off;   // byte offset of the level inside a frame's raw-pyramid block (level 0: caller image)
    int64_t blur_off;  // byte offset inside a frame's blurred-pyramid block
};

struct Plan {
    LevelGeom lv[kLevels];
    int width, height;
    int nfeatures;
    int total_tiles;
    int cand_frame_entries;   // candidate entries per frame (all levels)
    int sel_frame_entries;    // selection entries per frame (all levels)
    int tie_mode;             // 0: column-filter ties round up everywhere; 1 / 2 / 3: to even for x < (w & ~3 / ~7 / ~15), up in the tail
    int fast_threshold;
    int sort_cap;             // power of two in [kSortCapMin, kSortCapMax]
    // band kernel tuning: survivor-queue size = min(50, band_qpct0 + band_qstep * level) % of a workgroup's pixels;
    // LDS budget (KB) above which a workgroup is not given a second strip
    int band_qpct0, band_qstep, band_budget_kb;
    // fused pyramid kernel: a workgroup owns pyr_bh level-0 rows and builds the matching rows of levels 1..7 in LDS
    int pyr_bh, pyr_nbands, pyr_lds_bytes;
    int pyr_off[kLevels];     // LDS byte offset of the level-l row buffer (level 0: the staged source band)
    int pyr_xtab_off;         // LDS byte offset of the copied x tables (uint32, indexed by lv[l].xtab - lv[1].xtab)
    int pyr_ytab_off;         // LDS byte offset of the per-level y-table slice (uint32, kPyrYSlice entries)
    int pyr_p0;               // LDS pitch of the staged level-0 band
    int pyr_xtab_n;           // entries of the x-table copy
    int stream_ok;            // the streaming FAST/blur kernel's pyramid-step ownership rule holds for this plan's tables
    int64_t raw_frame_bytes;
    int64_t blur_frame_bytes;
    int64_t pixels_total;     // P of BASELINE.md section 3
};

// Host-side construction (orb_plan.cpp). tab receives the packed resize coefficients (ofs | c1 << 16).
// tie_mode: blur_tie_mode of aria_orb_config (0..3); level_size_mode: 0 = cvRound(dim * (1.0f / scale)), 1 = cvRound(dim / scale)
int build_plan(int width, int height, int nfeatures, int cand_cap_scale, int tie_mode, Plan* plan,
               uint32_t* tab, int tab_capacity, int* tab_used, int level_size_mode = 0);
int64_t plan_tab_entries(int width, int height);
// Per band and level: {comp_lo, comp_n, own_lo, own_n} = rows the band computes / rows it also writes to HBM.
// Returns the number of ints written (pyr_nbands * kLevels * 4). Fills plan->pyr_*.
int build_pyramid_bands(Plan* plan, const uint32_t* tab, int* out, int out_capacity);

}  // namespace aria
