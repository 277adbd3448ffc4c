// Host logic of the extractor: pyramid geometry, per-level quotas, resize coefficient tables.
// Follows CPU cv::ORB (OpenCV 4.9.0 modules/features2d/src/orb.cpp, modules/imgproc/src/resize.cpp) for the
// configuration fixed at reference src/adapters/gpu/OrbCudaExtractor.cpp:35-45.
#include "orb_plan.h"

#include <algorithm>
#include <cstdlib>
#include <cmath>

#include "aria_orb_hip.h"

// (see common.h: the product build reads no environment variable)
#ifdef ARIA_VARIANTS
static const char* aria_getenv(const char* n) { return std::getenv(n); }
#else
static const char* aria_getenv(const char*) { return nullptr; }
#endif

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
        const float inv_up = 1.0f / layer_scale(l - 1);
        n += (round_half_even((float)width * inv_up) + 3) / 4;      // xinv: one entry per dword of the level above
        n += round_half_even((float)height * inv_up);               // ytr: one entry per row of the level above
    }
    return n + 4 * kLevels;       // (level_size_mode 1 can make a level one pixel larger)
}

int build_plan(int width, int height, int nfeatures, int cand_cap_scale, int tie_mode, Plan* P,
               uint32_t* tab, int tab_capacity, int* tab_used, int level_size_mode) {
    if (width < 16 || height < 16 || width > kMaxDim || height > kMaxDim || nfeatures < 0) return ARIA_E_INVALID;
    *P = Plan{};
    P->width = width;
    P->height = height;
    P->nfeatures = nfeatures;
    P->tie_mode = tie_mode;
    P->fast_threshold = kFastThreshold;
    P->band_qpct0 = 10;
    P->band_qstep = 4;
    P->band_budget_kb = 24;

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
    P->stream_ok = 1;
    int64_t raw = 0, blur = 0, pix = 0;
    for (int l = 0; l < kLevels; l++) {
        LevelGeom& g = P->lv[l];
        g.scale = layer_scale(l);
        float inv = 1.0f / g.scale;   // orb.cpp detectAndCompute: Size sz(cvRound(cols*inv_scale), cvRound(rows*inv_scale))
        g.w = round_half_even((float)width * inv);
        g.h = round_half_even((float)height * inv);
        if (level_size_mode == 1) {     // the other reading of orb.cpp: one float division
            g.w = round_half_even((float)width / g.scale);
            g.h = round_half_even((float)height / g.scale);
        }
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
        g.sel_cap = (quota[l] + kSelSlack + 3) & ~3;   // multiple of 4: a describe wave's 4 keypoints share one level
        g.sel_off = sel;
        sel += g.sel_cap;
        g.tiles_x = (g.w + kTileW - 1) / kTileW;
        g.tile_base = tile;
        tile += g.tiles_x * ((g.h + kTileH - 1) / kTileH);
        g.raw_off = raw;
        if (l >= 1) raw += (int64_t)g.pitch * g.h;
        g.blur_off = blur;
        blur += (int64_t)g.pitch * ((g.h + 3) & ~3);     // whole row quads: the batch path stores the blurred level in Q4 order (orb_device.h)
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
            // k_fast_blur_stream's pyramid step: output dword gx (pixels 4 gx .. 4 gx + 3 of this level) is produced by the
            // lane holding the source dword that contains its ANCHOR column ox(4 gx + 1); that lane reads source columns from
            // its left neighbour's last two pixels to its right neighbour's last pixel. Table of hosted dwords per source
            // dword, and the check that every source column of every hosted dword lies in that window (it does for every
            // scale near 1.2: 4 output pixels span at most 6 source columns; a plan that fails keeps the band kernel).
            const int Ds = (s.w + 3) / 4, Dn = (g.w + 3) / 4;
            if (tabpos + Ds > tab_capacity) return ARIA_E_INVALID;
            g.xinv = tabpos;
            for (int k = 0; k < Ds; k++) tab[tabpos + k] = 0xFFFFFFFFu;
            const uint32_t* xt = tab + g.xtab;
            for (int gx = 0; gx < Dn; gx++) {
                const int a = (int)(xt[std::min(4 * gx + 1, g.w - 1)] & 0xFFFFu);
                // host = the source dword of the anchor; a partial last output dword (its anchor index is clamped, so it
                // can fall into its predecessor's source dword) takes a free neighbour that still sees all its columns
                int host = -1;
                const int tries[3] = {a >> 2, (a >> 2) + 1, (a >> 2) - 1};
                for (int k = 0; k < 3 && host < 0; k++) {
                    const int gs = tries[k];
                    if (gs < 0 || gs >= Ds || tab[tabpos + gs] != 0xFFFFFFFFu) continue;
                    bool ok = true;
                    for (int i = 0; i < 4; i++) {
                        const int ox = (int)(xt[std::min(4 * gx + i, g.w - 1)] & 0xFFFFu);
                        if (ox < 4 * gs - 2 || ox + 1 > 4 * gs + 7) ok = false;
                    }
                    if (ok) host = gs;
                }
                if (host < 0) P->stream_ok = 0;
                else {
                    tab[tabpos + host] = (uint32_t)gx;
                    // the walk blends from REGISTERS (own dword and both neighbours): pixel 0's two source bytes come out of
                    // (left neighbour, own dword), pixels 1..3's out of (own dword, right neighbour) -- static operands
                    // (pixels past the level's width -- a partial last dword -- are padding: the kernel blends zeros there)
                    for (int i = 0; i < 4 && 4 * gx + i < g.w; i++) {
                        const int d = (int)(xt[4 * gx + i] & 0xFFFFu) - 4 * host;
                        if (i == 0 ? (d < -4 || d > 2) : (d < 0 || d > 6)) P->stream_ok = 0;
                    }
                }
            }
            tabpos += Ds;
        }
    }
    // (behind every other table: k_pyramid copies the span of the x / y tables into LDS, which must not grow)
    for (int l = 1; l < kLevels; l++) {
        LevelGeom& g = P->lv[l];
        const LevelGeom& s = P->lv[l - 1];
        // the pyramid step by SOURCE row: row t of the level above is the lower source row of at most one output row, whose
        // upper source row is t - 1 (every scale near 1.2); an output row clamped to the last source row (oy = h - 1,
        // weight 0 on the row below) takes that row with cy = 256, i.e. all weight on the row itself
        if (tabpos + s.h > tab_capacity) return ARIA_E_INVALID;
        g.ytr = tabpos;
        for (int k = 0; k < s.h; k++) tab[tabpos + k] = 0u;
        {
            const uint32_t* yt = tab + g.ytab;
            for (int dy = 0; dy < g.h; dy++) {
                const int oy = (int)(yt[dy] & 0xFFFFu);
                uint32_t cy1 = yt[dy] >> 16;
                int rb = oy + 1;
                if (rb > s.h - 1) { rb = s.h - 1; if (cy1 != 0) P->stream_ok = 0; cy1 = 256u; }
                if (oy < 0 || oy > s.h - 1 || tab[tabpos + rb] != 0u) { P->stream_ok = 0; continue; }
                tab[tabpos + rb] = (uint32_t)dy | (cy1 << 16) | 0x80000000u;
            }
        }
        tabpos += s.h;
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

// Row ownership for the fused pyramid kernel. Level l+1 row dy is a blend of level l rows oy(dy), oy(dy)+1.
//   own_0     = the band's level-0 rows
//   own_{l+1} = { dy : oy_{l+1}(dy) in own_l }          (contiguous because oy is monotone; partitions level l+1)
//   comp_l    = own_l plus the extra trailing rows deeper levels of this band read (one more per level at most)
static int pyramid_bands_for(Plan* P, const uint32_t* tab, int bh, int* out, int out_capacity) {
    const int xt_lo = P->lv[1].xtab, xt_hi = P->lv[kLevels - 1].xtab + P->lv[kLevels - 1].w;
    const int p0 = (P->lv[0].w + 15) / 16 * 16;
    const int nb = (P->lv[0].h + bh - 1) / bh;
    if (nb * kLevels * 4 > out_capacity) return -1;
    int cap[kLevels] = {0};
    for (int b = 0; b < nb; b++) {
        int own_lo[kLevels], own_hi[kLevels], comp_hi[kLevels];
        own_lo[0] = b * bh;
        own_hi[0] = std::min(P->lv[0].h, (b + 1) * bh);
        for (int l = 1; l < kLevels; l++) {
            const uint32_t* yt = tab + P->lv[l].ytab;
            int lo = 0, hi = 0;
            while (lo < P->lv[l].h && (int)(yt[lo] & 0xFFFF) < own_lo[l - 1]) lo++;   // first dy with oy >= own_lo[l-1]
            hi = lo;
            while (hi < P->lv[l].h && (int)(yt[hi] & 0xFFFF) < own_hi[l - 1]) hi++;   // first dy with oy >= own_hi[l-1]
            own_lo[l] = lo;
            own_hi[l] = hi;
        }
        comp_hi[kLevels - 1] = own_hi[kLevels - 1];
        for (int l = kLevels - 2; l >= 0; l--) {
            int need = own_hi[l];
            if (comp_hi[l + 1] > own_lo[l + 1]) {
                const uint32_t* yt = tab + P->lv[l + 1].ytab;
                const int oy_last = (int)(yt[comp_hi[l + 1] - 1] & 0xFFFF);
                need = std::max(need, std::min(P->lv[l].h, oy_last + 2));
            }
            comp_hi[l] = need;
        }
        for (int l = 0; l < kLevels; l++) {
            int* o = out + (b * kLevels + l) * 4;
            o[0] = own_lo[l];
            o[1] = comp_hi[l] - own_lo[l];
            o[2] = own_lo[l];
            o[3] = own_hi[l] - own_lo[l];
            cap[l] = std::max(cap[l], o[1]);
            if (o[1] > kPyrYSlice) return -1;
        }
    }
    int off = 0;
    P->pyr_p0 = p0;
    for (int l = 0; l < kLevels; l++) {
        P->pyr_off[l] = off;
        off += (cap[l] + 1) * (l == 0 ? p0 : P->lv[l].pitch);   // +1 row of slack: dword windows may run past a row's end
    }
    P->pyr_xtab_off = off;
    P->pyr_xtab_n = xt_hi - xt_lo;
    off += 4 * (xt_hi - xt_lo);
    P->pyr_ytab_off = off;
    off += 4 * kPyrYSlice;
    P->pyr_bh = bh;
    P->pyr_nbands = nb;
    P->pyr_lds_bytes = off + 64;
    return nb * kLevels * 4;
}

int build_pyramid_bands(Plan* P, const uint32_t* tab, int* out, int out_capacity) {
    // k_pyramid's default user is the single-frame latency schedule (one frame = one launch on an otherwise idle chip), so
    // the bands are as thin as the table allows: 8 level-0 rows = 60 workgroups at 640x480, each a short chain of seven
    // dependent levels (measured per aria_orb_extract: 8 rows 116.5 us, 12 rows 119.5, 24 rows -- the largest that lets
    // three workgroups share a CU, the former choice -- 121-123). The halo (one extra source row per level, amplified by 1.2
    // per level) makes thin bands recompute a lot, which is why the batch path does not use this kernel at all.
    const int lds_max = 150 * 1024;
    if (const char* e = aria_getenv("ARIA_PYR_BH")) {
        const int n = pyramid_bands_for(P, tab, std::max(8, atoi(e) & ~7), out, out_capacity);
        if (n > 0 && P->pyr_lds_bytes <= lds_max) return n;
    }
    for (int bh = 8; bh <= 64; bh += 8) {
        const int n = pyramid_bands_for(P, tab, bh, out, out_capacity);
        if (n > 0 && P->pyr_lds_bytes <= lds_max) return n;
    }
    return pyramid_bands_for(P, tab, 8, out, out_capacity);
}

}  // namespace aria
