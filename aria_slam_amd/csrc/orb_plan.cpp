// Host logic of the extractor: pyramid geometry, per-level quotas, resize coefficient tables.
// Follows CPU cv::ORB (OpenCV 4.9.0 modules/features2d/src/orb.cpp, modules/imgproc/src/resize.cpp) for the
// configuration fixed at reference src/adapters/gpu/OrbCudaExtractor.cpp:35-45.
#include "orb_plan.h"

#include <algorithm>
#include <cmath>

#include "aria_orb_hip.h"

namespace aria {

namespace {

inline int round_half_even(float v) { return (int)std::lrintf(v); }
inline int round_half_even(double v) { return (int)std::lrint(v); }
inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

// orb.cpp getScale(): (float)pow(scaleFactor, level - firstLevel); scaleFactor is the double of 1.2f.
float layer_scale(int level) { return (float)std::pow((double)kScaleFactor, (double)level); }

// resize.cpp interpolationLinear<uchar>::getCoeffs, ufixedpoint16 (8 fractional bits).
void axis_coeffs(int ssize, int dsize, uint32_t* out) {
    double inv_scale = (double)dsize / (double)ssize;   // cv::resize: inv_scale_x = dsize.width / ssize.width
    double scale = 1.0 / inv_scale;
    for (int d = 0; d < dsize; d++) {
        double fval = scale * ((double)d + 0.5) - 0.5;
        int ival = (int)std::floor(fval);
        uint32_t ofs = 0, c1 = 0;
        if (ival >= 0 && ssize > 1) {
            if (ival < ssize - 1) {
                ofs = (uint32_t)ival;
                c1 = (uint32_t)round_half_even((fval - (double)ival) * 256.0);
            } else {
                ofs = (uint32_t)(ssize - 1);
            }
        }
        out[d] = ofs | (c1 << 16);
    }
}

}  // namespace

int64_t plan_tab_entries(int width, int height) {
    int64_t n = 0;
    for (int l = 1; l < kLevels; l++) {
        float inv = 1.0f / layer_scale(l);
        n += round_half_even((float)width * inv) + round_half_even((float)height * inv);
    }
    return n;
}

int build_plan(int width, int height, int nfeatures, int cand_cap_scale, int tie_mode, Plan* P,
               uint32_t* tab, int tab_capacity, int* tab_used) {
    if (width < 16 || height < 16 || width > kMaxDim || height > kMaxDim || nfeatures < 0) return ARIA_E_INVALID;
    *P = Plan{};
    P->width = width;
    P->height = height;
    P->nfeatures = nfeatures;
    P->tie_mode = tie_mode;
    P->fast_threshold = kFastThreshold;

    // orb.cpp computeKeyPoints: nfeaturesPerLevel
    int quota[kLevels];
    {
        float factor = (float)(1.0 / (double)kScaleFactor);
        float nd = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)kLevels));
        int sum = 0;
        for (int l = 0; l < kLevels - 1; l++) {
            quota[l] = round_half_even(nd);
            sum += quota[l];
            nd *= factor;
        }
        quota[kLevels - 1] = std::max(nfeatures - sum, 0);
    }

    int tabpos = 0, tile = 0, cand = 0, sel = 0;
    int64_t raw = 0, blur = 0, pix = 0;
    for (int l = 0; l < kLevels; l++) {
        LevelGeom& g = P->lv[l];
        g.scale = layer_scale(l);
        float inv = 1.0f / g.scale;   // orb.cpp detectAndCompute: Size sz(cvRound(cols*inv_scale), cvRound(rows*inv_scale))
        g.w = round_half_even((float)width * inv);
        g.h = round_half_even((float)height * inv);
        if (g.w < 8 || g.h < 8) return ARIA_E_INVALID;
        g.pitch = align_up(g.w, 16);
        g.quota = quota[l];
        // worst case after 3x3 strict-max suppression is one corner per 2x2 block of the kept region
        int region = std::max(g.w - 2 * kEdgeThreshold, 0) * std::max(g.h - 2 * kEdgeThreshold, 0);
        // default: the worst case itself, so the list can never overflow; cand_cap_scale > 0 trades that for memory
        const int worst = region / 4 + 64;
        g.cand_cap = cand_cap_scale <= 0 ? worst : std::max(256, std::min(worst, cand_cap_scale * std::max(quota[l], 1)));
        g.cand_off = cand;
        cand += g.cand_cap;
        g.sel_cap = quota[l] + kSelSlack;
        g.sel_off = sel;
        sel += g.sel_cap;
        g.tiles_x = (g.w + kTileW - 1) / kTileW;
        g.tile_base = tile;
        tile += g.tiles_x * ((g.h + kTileH - 1) / kTileH);
        g.raw_off = raw;
        if (l >= 1) raw += (int64_t)g.pitch * g.h;
        g.blur_off = blur;
        blur += (int64_t)g.pitch * g.h;
        pix += (int64_t)g.w * g.h;
        if (l >= 1) {
            const LevelGeom& s = P->lv[l - 1];
            if (tabpos + g.w + g.h > tab_capacity) return ARIA_E_INVALID;
            g.xtab = tabpos;
            axis_coeffs(s.w, g.w, tab + tabpos);
            tabpos += g.w;
            g.ytab = tabpos;
            axis_coeffs(s.h, g.h, tab + tabpos);
            tabpos += g.h;
        }
    }
    // LDS sort capacity of k_select: room for retainBest(2*quota) plus ties at the cut, power of two
    int sc = kSortCapMin;
    while (sc < 2 * quota[0] + 1024 && sc < kSortCapMax) sc <<= 1;
    if (2 * quota[0] + 64 > kSortCapMax) return ARIA_E_INVALID;   // nfeatures beyond what one workgroup can rank
    P->sort_cap = sc;
    P->total_tiles = tile;
    P->cand_frame_entries = cand;
    P->sel_frame_entries = sel;
    P->raw_frame_bytes = (raw + 255) / 256 * 256;
    P->blur_frame_bytes = (blur + 255) / 256 * 256;
    P->pixels_total = pix;
    if (tab_used) *tab_used = tabpos;
    return ARIA_OK;
}

}  // namespace aria
