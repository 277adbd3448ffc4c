/*
 * orb_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).  *** PARITY UNPINNED ***
 * See orb_oracle.h for scope, provenance and who may call this.
 *
 * Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off matters: the float stages (Harris response, fastAtan2, pattern rotation) restate
 * OpenCV code compiled for the SSE3 baseline, i.e. separately rounded multiplies and adds, no FMA.
 *
 * "cv" citations are to OpenCV 4.9.0 (the version the reference pins: scripts/setup_machine.sh:192-194);
 * "ref" citations are to files under /root/reference.
 */
#include "orb_oracle.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <vector>

namespace {

// cvRound: round half to even (cv: core/fast_math.hpp cvRound -> _mm_cvtsd_si32 / lrint).
inline int cv_round(double v) { return (int)std::lrint(v); }
inline int cv_round(float v) { return (int)std::lrintf(v); }
inline int cv_floor(double v) { return (int)std::floor(v); }
inline int cv_ceil(double v) { return (int)std::ceil(v); }

const int kEdgeThreshold = 31;   // ref src/adapters/gpu/OrbCudaExtractor.cpp:39
const int kPatchSize = 31;       // ref :43
const int kHalfPatch = 15;
const int kHarrisBlock = 7;      // cv orb.cpp computeKeyPoints: HarrisResponses(..., 7, HARRIS_K)
const float kHarrisK = 0.04f;    // cv orb.cpp HARRIS_K

const int kPattern[1024] = {
#include "orb_pattern_31.inc"
};

// BORDER_REFLECT_101 index (cv: borderInterpolate).
inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * n - 2 - i;
    }
    return i;
}

// 16-point Bresenham ring of radius 3, in OpenCV's order (cv: fast_score.cpp makeOffsets, offsets16).
const int kRingX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int kRingY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// cv fast_score.cpp cornerScore<16>: largest threshold for which the pixel is still a FAST-9 corner,
// = max over the 16 nine-pixel arcs of min(v - ring) and of min(ring - v), minus 1.
int corner_score16(const uint8_t* p, int stride) {
    int v = p[0];
    int d[25];
    for (int k = 0; k < 25; k++) d[k] = v - p[kRingY[k & 15] * stride + kRingX[k & 15]];
    int best_dark = -1000, best_bright = 1000;  // q0, q1 in the SIMD form
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
        for (int j = 1; j < 9; j++) {
            mn = std::min(mn, d[k + j]);
            mx = std::max(mx, d[k + j]);
        }
        best_dark = std::max(best_dark, mn);
        best_bright = std::min(best_bright, mx);
    }
    return std::max(best_dark, -best_bright) - 1;
}

// cv fast.cpp FAST_t<16>: corner iff >= 9 contiguous ring pixels are all < v - t or all > v + t.
bool is_fast9_corner(const uint8_t* p, int stride, int t) {
    int v = p[0];
    {   // any 9 contiguous ring pixels contain at least 2 of the 4 compass pixels (0, 4, 8, 12): cheap reject,
        // same role as the early-outs in cv fast.cpp; it never changes the outcome.
        int c0 = p[3 * stride], c4 = p[3], c8 = p[-3 * stride], c12 = p[-3];
        int nd = (c0 < v - t) + (c4 < v - t) + (c8 < v - t) + (c12 < v - t);
        int nb = (c0 > v + t) + (c4 > v + t) + (c8 > v + t) + (c12 > v + t);
        if (nd < 2 && nb < 2) return false;
    }
    int ring[25];
    for (int k = 0; k < 25; k++) ring[k] = p[kRingY[k & 15] * stride + kRingX[k & 15]];
    int cd = 0, cb = 0;
    for (int k = 0; k < 25; k++) {
        if (ring[k] < v - t) { if (++cd > 8) return true; } else cd = 0;
        if (ring[k] > v + t) { if (++cb > 8) return true; } else cb = 0;
    }
    return false;
}

struct Cand { int x, y; float resp; };

// cv keypoint.cpp KeyPointsFilter::retainBest: keep every keypoint whose response is >= the n-th largest
// (nth_element + partition => ties at the boundary are all retained). Order afterwards is libstdc++-defined
// in OpenCV; here the relative order is left untouched (callers canonicalise).
void retain_best(std::vector<Cand>& v, int n_points) {
    if (n_points >= 0 && v.size() > (size_t)n_points) {
        if (n_points == 0) { v.clear(); return; }
        std::vector<float> r(v.size());
        for (size_t i = 0; i < v.size(); i++) r[i] = v[i].resp;
        std::nth_element(r.begin(), r.begin() + (n_points - 1), r.end(), std::greater<float>());
        float ambiguous = r[n_points - 1];
        std::vector<Cand> out;
        out.reserve(v.size());
        for (const Cand& c : v) if (c.resp >= ambiguous) out.push_back(c);
        v.swap(out);
    }
}

}  // namespace

extern "C" {

void orc_default_params(orc_params* p) {
    p->nfeatures = 1000;       // ref include/adapters/gpu/OrbCudaExtractor.hpp:12
    p->scale_factor = 1.2f;    // ref src/adapters/gpu/OrbCudaExtractor.cpp:37
    p->nlevels = 8;            // :38
    p->fast_threshold = 20;    // :44
    p->blur_tie_mode = 1;
    p->level_size_mode = 0;
}

const int* orc_bit_pattern_31(void) { return kPattern; }

/* cv orb.cpp: static inline float getScale(int level, int firstLevel, double scaleFactor)
 *   { return (float)std::pow(scaleFactor, (double)(level - firstLevel)); }
 * scaleFactor is ORB_Impl's double member initialised from the float argument 1.2f. */
float orc_layer_scale(const orc_params* p, int level) {
    return (float)std::pow((double)p->scale_factor, (double)level);
}

/* cv orb.cpp detectAndCompute: float inv_scale = 1.0f / scale;
 *                              Size sz(cvRound(image.cols * inv_scale), cvRound(image.rows * inv_scale)); */
void orc_level_size(const orc_params* p, int w, int h, int level, int* lw, int* lh) {
    float scale = orc_layer_scale(p, level);
    float inv_scale = 1.0f / scale;
    if (p->level_size_mode == 1) {      // the other reading of orb.cpp: one float division instead of reciprocal + multiply
        *lw = cv_round((float)w / scale);
        *lh = cv_round((float)h / scale);
        return;
    }
    *lw = cv_round((float)w * inv_scale);
    *lh = cv_round((float)h * inv_scale);
}

/* cv orb.cpp computeKeyPoints: geometric split of nfeatures over the levels. */
void orc_feature_quotas(const orc_params* p, int* quota) {
    int nlevels = p->nlevels;
    float factor = (float)(1.0 / (double)p->scale_factor);
    float nd = p->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        quota[l] = cv_round(nd);
        sum += quota[l];
        nd *= factor;
    }
    quota[nlevels - 1] = std::max(p->nfeatures - sum, 0);
}

/* cv resize.cpp interpolationLinear<uchar>::getCoeffs with ufixedpoint16 (8 fractional bits):
 *   scale = 1 / inv_scale (inv_scale = (double)dsize/ssize, computed in cv::resize),
 *   fval = scale*(d + 0.5) - 0.5, ival = floor(fval),
 *   coeffs[1] = cvRound((fval - ival) * 256), coeffs[0] = 256 - coeffs[1];
 *   left of the source -> leftmost pixel, at/after the last source pixel -> last pixel.   */
void orc_resize_coeffs(int ssize, int dsize, int* ofs, int* c1) {
    double inv_scale = (double)dsize / (double)ssize;
    double scale = 1.0 / inv_scale;
    for (int d = 0; d < dsize; d++) {
        double fval = scale * ((double)d + 0.5) - 0.5;
        int ival = cv_floor(fval);
        if (ival >= 0 && ssize > 1) {
            if (ival < ssize - 1) {
                ofs[d] = ival;
                c1[d] = cv_round((fval - (double)ival) * 256.0);
            } else {
                ofs[d] = ssize - 1;
                c1[d] = 0;
            }
        } else {
            ofs[d] = 0;
            c1[d] = 0;
        }
    }
}

/* cv resize.cpp resize_bitExact<uchar, interpolationLinear<uchar>>:
 *   horizontal: H = c0*p[o] + c1*p[o+1]                 (ufixedpoint16, exact, <= 255*256)
 *   vertical:   out = (cy0*H0 + cy1*H1 + 32768) >> 16   (ufixedpoint32 -> uint8, round half up)       */
void orc_resize_linear_exact(const uint8_t* src, int sw, int sh, int sstride,
                             uint8_t* dst, int dw, int dh, int dstride) {
    std::vector<int> xo(dw), xc(dw), yo(dh), yc(dh);
    orc_resize_coeffs(sw, dw, xo.data(), xc.data());
    orc_resize_coeffs(sh, dh, yo.data(), yc.data());
    for (int dy = 0; dy < dh; dy++) {
        const uint8_t* r0 = src + (size_t)yo[dy] * sstride;
        const uint8_t* r1 = src + (size_t)std::min(yo[dy] + 1, sh - 1) * sstride;
        uint32_t cy1 = (uint32_t)yc[dy], cy0 = 256u - cy1;
        for (int dx = 0; dx < dw; dx++) {
            int o0 = xo[dx], o1 = std::min(o0 + 1, sw - 1);
            uint32_t cx1 = (uint32_t)xc[dx], cx0 = 256u - cx1;
            uint32_t h0 = cx0 * r0[o0] + cx1 * r0[o1];
            uint32_t h1 = cx0 * r1[o0] + cx1 * r1[o1];
            uint32_t v = (cy0 * h0 + cy1 * h1 + 32768u) >> 16;
            dst[(size_t)dy * dstride + dx] = (uint8_t)std::min(v, 255u);
        }
    }
}

/* cv fast.cpp FAST_t<16>: scores exist for x in [3, w-4], y in [3, h-4]; stored as uchar. */
void orc_fast_score_map(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score) {
    std::memset(score, 0, (size_t)w * h);
    threshold = std::min(std::max(threshold, 0), 255);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* p = img + (size_t)y * stride + x;
            if (is_fast9_corner(p, stride, threshold))
                score[(size_t)y * w + x] = (uint8_t)corner_score16(p, stride);
        }
}

/* cv fast.cpp FAST_t<16> with nonmax_suppression: keep a corner iff its score is strictly greater than the
 * scores of its 8 neighbours (non-corners score 0). Raster order, as OpenCV emits them. */
int orc_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold,
                    int* xs, int* ys, int* scores, int cap) {
    std::vector<uint8_t> s((size_t)w * h);
    orc_fast_score_map(img, w, h, stride, threshold, s.data());
    threshold = std::min(std::max(threshold, 0), 255);
    int n = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* c = &s[(size_t)y * w + x];
            // a corner whose score is 0 (only possible with threshold 0) can never be strictly greater
            if (!is_fast9_corner(img + (size_t)y * stride + x, stride, threshold)) continue;
            int sc = c[0];
            if (sc > c[1] && sc > c[-1] && sc > c[-w - 1] && sc > c[-w] && sc > c[-w + 1] &&
                sc > c[w - 1] && sc > c[w] && sc > c[w + 1]) {
                if (n < cap) { xs[n] = x; ys[n] = y; scores[n] = sc; }
                n++;
            }
        }
    return n;
}

/* cv smooth.dispatch.cpp getGaussianKernel(7, 2.0, CV_32F) followed by
 * cv filter.dispatch.cpp createSeparableLinearFilter's 8-bit path: kernel.convertTo(CV_32S, 256).       */
void orc_gaussian_kernel7_fixed(int* k7) {
    const int n = 7;
    const double sigma = 2.0;
    double scale2x = -0.5 / (sigma * sigma);
    double t[7], sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        t[i] = std::exp(scale2x * x * x);
        sum += t[i];
    }
    for (int i = 0; i < n; i++) {
        float cf = (float)(t[i] / sum);
        k7[i] = cv_round((double)cf * 256.0);
    }
}

/* The blur ORB applies before sampling descriptors (cv orb.cpp detectAndCompute:
 *   GaussianBlur(workingMat, workingMat, Size(7,7), 2, 2, BORDER_REFLECT_101), workingMat an ROI of the
 *   pyramid buffer whose 32-px surround already holds the BORDER_REFLECT_101 extension of the level).
 * Because workingMat is a submatrix and BORDER_ISOLATED is not set, cv::GaussianBlur skips its bit-exact
 * ufixedpoint16 path and runs sepFilter2D; for 8U->8U with smooth symmetric kernels that engine uses
 * integer kernels scaled by 2^8 per pass (sum = 18+34+49+55+49+34+18 = 257, NOT renormalised), int32
 * accumulation, and one final rounding by 2^16:
 *   - vector body (cv filter.simd.hpp SymmColumnVec_32s8u): float(sum)/65536 -> v_round (ties to even);
 *     exact for every sum < 2^24; larger sums saturate to 255 either way. Vector steps go down to 4
 *     lanes, so the body covers columns x < (w & ~3);
 *   - scalar tail (FixedPtCastEx<int,uchar>): (sum + 32768) >> 16 (ties up).
 * tie_mode 1 reproduces that split, tie_mode 0 uses ties-up everywhere; tie_mode 2 / 3 put the end of the
 * vector body at w & ~7 / w & ~15 (a dispatch whose narrowest vector step is 8 / 16 lanes).            */
void orc_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, int tie_mode) {
    int k[7];
    orc_gaussian_kernel7_fixed(k);
    std::vector<int32_t> rows((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int32_t s = 0;
            for (int i = -3; i <= 3; i++) s += k[i + 3] * src[(size_t)y * sstride + reflect101(x + i, w)];
            rows[(size_t)y * w + x] = s;
        }
    int body = tie_mode == 0 ? 0 : tie_mode == 2 ? (w & ~7) : tie_mode == 3 ? (w & ~15) : (w & ~3);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int64_t s = 0;
            for (int j = -3; j <= 3; j++) s += (int64_t)k[j + 3] * rows[(size_t)reflect101(y + j, h) * w + x];
            int64_t q = s >> 16, r = s & 0xFFFF;
            if (r > 32768) q += 1;
            else if (r == 32768) q += (x < body) ? (q & 1) : 1;
            dst[(size_t)y * dstride + x] = (uint8_t)std::min<int64_t>(q, 255);
        }
}

/* cv orb.cpp HarrisResponses(img, layerinfo, pts, blockSize=7, harris_k=0.04f). (x,y) = integer centre. */
float orc_harris_response(const uint8_t* img, int stride, int x, int y) {
    const int blockSize = kHarrisBlock, r = blockSize / 2;
    float scale = 1.f / ((1 << 2) * blockSize * 255.f);
    float scale_sq_sq = scale * scale * scale * scale;
    const uint8_t* ptr0 = img + (std::ptrdiff_t)(y - r) * stride + (x - r);
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < blockSize; i++)
        for (int j = 0; j < blockSize; j++) {
            const uint8_t* ptr = ptr0 + (std::ptrdiff_t)i * stride + j;
            int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-stride + 1] - ptr[-stride - 1]) + (ptr[stride + 1] - ptr[stride - 1]);
            int Iy = (ptr[stride] - ptr[-stride]) * 2 + (ptr[stride - 1] - ptr[-stride - 1]) + (ptr[stride + 1] - ptr[-stride + 1]);
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    return ((float)a * b - (float)c * c - kHarrisK * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

/* cv orb.cpp computeKeyPoints: end-of-row table of the circular patch. */
void orc_umax(int* umax) {
    int half = kHalfPatch;
    std::vector<int> u(half + 2, 0);
    int v, v0, vmax = cv_floor(half * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(half * std::sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v) u[v] = cv_round(std::sqrt((double)half * half - v * v));
    for (v = half, v0 = 0; v >= vmin; --v) {
        while (u[v0] == u[v0 + 1]) ++v0;
        u[v] = v0;
        ++v0;
    }
    for (v = 0; v <= half; v++) umax[v] = u[v];
}

/* cv orb.cpp ICAngles: first-order moments over the radius-15 disc. */
void orc_ic_moments(const uint8_t* img, int stride, int x, int y, int* m01_out, int* m10_out) {
    int umax[kHalfPatch + 1];
    orc_umax(umax);
    const uint8_t* center = img + (std::ptrdiff_t)y * stride + x;
    int m_01 = 0, m_10 = 0;
    for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m_10 += u * center[u];
    for (int v = 1; v <= kHalfPatch; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    *m01_out = m_01;
    *m10_out = m_10;
}

/* cv mathfuncs_core.dispatch.cpp fastAtan2 -> atan_f32 (scalar form, compiled without FMA). */
float orc_fast_atan2(float y, float x) {
    static const float atan2_p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    static const float atan2_p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    static const float atan2_p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    static const float atan2_p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = std::abs(x), ay = std::abs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

float orc_ic_angle(const uint8_t* img, int stride, int x, int y) {
    int m01, m10;
    orc_ic_moments(img, stride, x, y, &m01, &m10);
    return orc_fast_atan2((float)m01, (float)m10);
}

/* cv orb.cpp computeOrbDescriptors evaluates `(float)cos(angle)`, `(float)sin(angle)` through libm, whose
 * last double ulp differs between libm builds -- OpenCV's own result is platform-defined at that level.
 * The restatement pins one double-precision algorithm (Cody-Waite reduction by pi/2 + the classic fdlibm
 * kernel polynomials, |error| < 1e-15) built from IEEE add/mul/floor only, so that host and gfx950 agree
 * by construction. It matches (float)libm_cos((double)x) wherever the double result is not within
 * ~1e-15 relative of a float rounding boundary (tests/test_oracle.py checks a large sample). */
void orc_sincos(double x, double* s_out, double* c_out) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;  // first 33 bits of pi/2
    const double pio2_lo = 6.07710050650619224932e-11;  // pi/2 - pio2_hi
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double fn = std::floor(x * invpio2 + 0.5);
    double r = (x - fn * pio2_hi) - fn * pio2_lo;
    double z = r * r;
    double sp = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    double sn = r + (z * r) * (S1 + z * sp);
    double cp = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double cs = 1.0 - (0.5 * z - z * cp);
    int q = (int)fn & 3;
    double s, c;
    switch (q) {
        case 0: s = sn; c = cs; break;
        case 1: s = cs; c = -sn; break;
        case 2: s = -sn; c = -cs; break;
        default: s = -cs; c = sn; break;
    }
    *s_out = s;
    *c_out = c;
}

/* cv orb.cpp computeOrbDescriptors, WTA_K == 2 branch. */
void orc_brief_descriptor(const uint8_t* blurred, int stride, int x, int y, float angle_deg, uint8_t* desc) {
    float angle = angle_deg;
    angle *= (float)(3.1415926535897932384626433832795 / 180.f);
    double sd, cd;
    orc_sincos((double)angle, &sd, &cd);
    float a = (float)cd, b = (float)sd;
    const uint8_t* center = blurred + (std::ptrdiff_t)y * stride + x;
    const int* pattern = kPattern;
    for (int i = 0; i < 32; ++i, pattern += 32) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            float px0 = (float)pattern[4 * j + 0], py0 = (float)pattern[4 * j + 1];
            float px1 = (float)pattern[4 * j + 2], py1 = (float)pattern[4 * j + 3];
            float x0 = px0 * a - py0 * b, y0 = px0 * b + py0 * a;
            float x1 = px1 * a - py1 * b, y1 = px1 * b + py1 * a;
            int t0 = center[cv_round(y0) * stride + cv_round(x0)];
            int t1 = center[cv_round(y1) * stride + cv_round(x1)];
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

/* One level of cv orb.cpp computeKeyPoints (see header). */
int orc_detect_level(const uint8_t* img, int w, int h, int stride, int quota, int fast_threshold,
                     int* xs, int* ys, float* harris, int cap, int* n_fast, int* n_after_first_retain) {
    std::vector<Cand> kps;
    {
        int n = orc_fast_detect(img, w, h, stride, fast_threshold, nullptr, nullptr, nullptr, 0);
        std::vector<int> fx(n), fy(n), fs(n);
        orc_fast_detect(img, w, h, stride, fast_threshold, fx.data(), fy.data(), fs.data(), n);
        // cv keypoint.cpp runByImageBorder(keypoints, img.size(), edgeThreshold = 31)
        if (!(h <= kEdgeThreshold * 2 || w <= kEdgeThreshold * 2)) {
            for (int i = 0; i < n; i++)
                if (fx[i] >= kEdgeThreshold && fx[i] < w - kEdgeThreshold &&
                    fy[i] >= kEdgeThreshold && fy[i] < h - kEdgeThreshold)
                    kps.push_back({fx[i], fy[i], (float)fs[i]});
        }
    }
    if (n_fast) *n_fast = (int)kps.size();
    retain_best(kps, 2 * quota);  // HARRIS_SCORE: 2 * featuresNum
    if (n_after_first_retain) *n_after_first_retain = (int)kps.size();
    for (Cand& c : kps) c.resp = orc_harris_response(img, stride, c.x, c.y);
    retain_best(kps, quota);
    std::sort(kps.begin(), kps.end(), [](const Cand& p, const Cand& q) {
        if (p.resp != q.resp) return p.resp > q.resp;
        if (p.y != q.y) return p.y < q.y;
        return p.x < q.x;
    });
    int n = (int)kps.size();
    for (int i = 0; i < n && i < cap; i++) { xs[i] = kps[i].x; ys[i] = kps[i].y; harris[i] = kps[i].resp; }
    return n;
}

int64_t orc_pyramid_layout(const orc_params* p, int w, int h, int* lw, int* lh, int64_t* offs) {
    int64_t total = 0;
    for (int l = 0; l < p->nlevels; l++) {
        orc_level_size(p, w, h, l, &lw[l], &lh[l]);
        offs[l] = total;
        total += (int64_t)lw[l] * lh[l];
    }
    return total;
}

/* cv orb.cpp detectAndCompute pyramid loop: level 0 = the image; level l = resize(level l-1,
 * INTER_LINEAR_EXACT) (prevImg is updated to currImg for every level > firstLevel). No blur here. */
void orc_build_pyramid(const uint8_t* img, int w, int h, int stride, const orc_params* p, uint8_t* out) {
    int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS];
    int64_t offs[ORC_MAX_LEVELS];
    orc_pyramid_layout(p, w, h, lw, lh, offs);
    for (int y = 0; y < h; y++) std::memcpy(out + offs[0] + (size_t)y * lw[0], img + (size_t)y * stride, w);
    for (int l = 1; l < p->nlevels; l++)
        orc_resize_linear_exact(out + offs[l - 1], lw[l - 1], lh[l - 1], lw[l - 1], out + offs[l], lw[l], lh[l], lw[l]);
}

void orc_blur_pyramid(const uint8_t* pyr, int w, int h, const orc_params* p, uint8_t* out) {
    int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS];
    int64_t offs[ORC_MAX_LEVELS];
    orc_pyramid_layout(p, w, h, lw, lh, offs);
    for (int l = 0; l < p->nlevels; l++)
        orc_gaussian_blur7(pyr + offs[l], lw[l], lh[l], lw[l], out + offs[l], lw[l], p->blur_tie_mode);
}

/* cv orb.cpp ORB_Impl::detectAndCompute(image, noArray(), keypoints, descriptors, false)
 * = ref src/legacy/Frame.cpp:47. */
int orc_orb_extract(const uint8_t* img, int w, int h, int stride, const orc_params* p,
                    orc_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
    int nl = p->nlevels;
    if (nl < 1 || nl > ORC_MAX_LEVELS) return -2;
    int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS], quota[ORC_MAX_LEVELS];
    int64_t offs[ORC_MAX_LEVELS];
    int64_t total = orc_pyramid_layout(p, w, h, lw, lh, offs);
    std::vector<uint8_t> pyr((size_t)total), blur((size_t)total);
    orc_build_pyramid(img, w, h, stride, p, pyr.data());
    orc_feature_quotas(p, quota);

    struct K { int x, y, l; float resp, angle; };
    std::vector<K> all;
    for (int l = 0; l < nl; l++) {
        int capl = (lw[l] * lh[l]) / 4 + 16;
        std::vector<int> xs(capl), ys(capl);
        std::vector<float> hr(capl);
        int n = orc_detect_level(pyr.data() + offs[l], lw[l], lh[l], lw[l], quota[l], p->fast_threshold,
                                 xs.data(), ys.data(), hr.data(), capl, nullptr, nullptr);
        for (int i = 0; i < n; i++) all.push_back({xs[i], ys[i], l, hr[i], 0.f});
    }
    // ICAngles on the un-blurred pyramid
    for (K& k : all) k.angle = orc_ic_angle(pyr.data() + offs[k.l], lw[k.l], k.x, k.y);

    *n_out = (int)all.size();
    if ((int)all.size() > cap) return -1;

    // pre-descriptor blur of every level, then descriptors from the blurred level
    orc_blur_pyramid(pyr.data(), w, h, p, blur.data());
    for (size_t i = 0; i < all.size(); i++) {
        const K& k = all[i];
        float sf = orc_layer_scale(p, k.l);
        orc_keypoint o;
        // cv computeKeyPoints: octave = level; size = patchSize*sf; later pt *= layerScale[octave]
        o.x = (float)k.x * sf;
        o.y = (float)k.y * sf;
        o.size = kPatchSize * sf;
        o.angle = k.angle;
        o.response = k.resp;
        o.octave = k.l;
        kps[i] = o;
        // cv computeOrbDescriptors: centre = (cvRound(pt.x * (1/sf)), cvRound(pt.y * (1/sf)))
        float inv = 1.f / sf;
        int cx = cv_round(o.x * inv), cy = cv_round(o.y * inv);
        orc_brief_descriptor(blur.data() + offs[k.l], lw[k.l], cx, cy, k.angle, desc + i * 32);
    }
    return 0;
}

/* ---- matching ------------------------------------------------------------------------------------------ */

int orc_hamming256(const uint8_t* a, const uint8_t* b) {
    int d = 0;
    for (int i = 0; i < 32; i++) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
}

/* cv batch_distance.cpp BatchDistInvoker, K = 2: in-order scan, strict '<' insertion => on equal distance the
 * lower train index ranks first. */
void orc_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int* idx, int* dist) {
    for (int i = 0; i < nq; i++) {
        int d0 = INT_MAX, d1 = INT_MAX, i0 = -1, i1 = -1;
        uint64_t qa[4];
        std::memcpy(qa, q + (size_t)i * 32, 32);
        for (int j = 0; j < nt; j++) {
            uint64_t tb[4];
            std::memcpy(tb, t + (size_t)j * 32, 32);
            int d = __builtin_popcountll(qa[0] ^ tb[0]) + __builtin_popcountll(qa[1] ^ tb[1]) +
                    __builtin_popcountll(qa[2] ^ tb[2]) + __builtin_popcountll(qa[3] ^ tb[3]);
            if (d < d1) {
                if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
                else { d1 = d; i1 = j; }
            }
        }
        idx[2 * i] = i0; idx[2 * i + 1] = i1;
        dist[2 * i] = d0; dist[2 * i + 1] = d1;
    }
}

/* ref src/adapters/gpu/CudaMatcher.cpp:28-68. */
int orc_match_ratio(const uint8_t* q, int nq, const uint8_t* t, int nt, float ratio, orc_match* out) {
    if (nq <= 0 || nt <= 0) return 0;  // :35-37
    std::vector<int> idx(2 * (size_t)nq), dist(2 * (size_t)nq);
    orc_knn2(q, nq, t, nt, idx.data(), dist.data());
    int n = 0;
    for (int i = 0; i < nq; i++) {
        bool keep;
        if (ratio == 0.0f) keep = idx[2 * i] >= 0;  // include/interfaces/IMatcher.hpp:18 "0.0 = disabled"
        else keep = idx[2 * i + 1] >= 0 && (float)dist[2 * i] < ratio * (float)dist[2 * i + 1];  // :60
        if (keep) { out[n].query_idx = i; out[n].train_idx = idx[2 * i]; out[n].distance = (float)dist[2 * i]; n++; }
    }
    return n;
}

/* ref src/legacy/LoopClosure.cpp:90-95 (double literal 0.7). */
int orc_count_good_matches_f64(const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio) {
    if (nq <= 0 || nt <= 0) return 0;
    std::vector<int> idx(2 * (size_t)nq), dist(2 * (size_t)nq);
    orc_knn2(q, nq, t, nt, idx.data(), dist.data());
    int good = 0;
    for (int i = 0; i < nq; i++)
        if (idx[2 * i + 1] >= 0 && (double)(float)dist[2 * i] < ratio * (double)(float)dist[2 * i + 1]) good++;
    return good;
}

/* ref src/legacy/LoopClosure.cpp:72-114 findCandidates. */
int orc_loop_candidates(const uint8_t* q, int nq, int64_t query_id,
                        const uint8_t* db, const int* kf_counts, const int64_t* kf_ids, int n_kf,
                        int min_frames_between, int* cand_idx, double* cand_score) {
    std::vector<std::pair<int, double>> cands;
    if (nq <= 0) return 0;                                           // :75
    size_t off = 0;
    for (int i = 0; i < n_kf; i++) {
        const uint8_t* t = db + off * 32;
        off += (size_t)kf_counts[i];
        if (query_id - kf_ids[i] < min_frames_between) continue;     // :81
        if (kf_counts[i] <= 0) continue;                             // :83
        int good = orc_count_good_matches_f64(q, nq, t, kf_counts[i], 0.7);  // :86-95
        double score = (double)good / std::max(1, nq);               // :98
        if (score > 0.1) cands.push_back({i, score});                // :99
    }
    // :105-106 std::sort by score descending (unstable in the reference; ties broken here by DB index)
    std::sort(cands.begin(), cands.end(), [](const auto& a, const auto& b) {
        if (a.second != b.second) return a.second > b.second;
        return a.first < b.first;
    });
    if (cands.size() > 5) cands.resize(5);                           // :109-111
    for (size_t i = 0; i < cands.size(); i++) { cand_idx[i] = cands[i].first; cand_score[i] = cands[i].second; }
    return (int)cands.size();
}

/* ref src/main.cpp:42-50 isInDynamicObject + :164-175: a ratio-test survivor is dropped when either endpoint lies in a box
 * of a dynamic class (the caller passes only those boxes). mode 0 = the legacy executable's test, cv::Rect::contains of the
 * keypoint converted to an integer point (cv::Point_<int>(Point2f) rounds half to even): x1 <= cvRound(x) < x2, same for y;
 * mode 1 = core::Detection::contains (include/core/Types.hpp:109-111): closed float intervals. boxes = (x1, y1, x2, y2).
 * kps are 24-byte KeyPoint records (x, y first). Returns the number kept (written to out, order preserved); *filtered =
 * main.cpp's filtered_count. */
static bool orc_in_box(float x, float y, const float* boxes, int nb, int mode) {
    for (int b = 0; b < nb; b++) {
        const float* r = boxes + 4 * b;
        if (mode == 0) {
            const int px = (int)std::nearbyint(x), py = (int)std::nearbyint(y);
            if ((int)r[0] <= px && px < (int)r[2] && (int)r[1] <= py && py < (int)r[3]) return true;
        } else if (x >= r[0] && x <= r[2] && y >= r[1] && y <= r[3]) {
            return true;
        }
    }
    return false;
}
int orc_filter_dynamic_matches(const void* kps_q, const void* kps_t, const void* matches, int n, const float* boxes, int nb,
                               int mode, void* out, int* filtered) {
    const float* kq = static_cast<const float*>(kps_q);
    const float* kt = static_cast<const float*>(kps_t);
    const int* m = static_cast<const int*>(matches);
    int* o = static_cast<int*>(out);
    int kept = 0, dropped = 0;
    for (int i = 0; i < n; i++) {
        const int qi = m[3 * i], ti = m[3 * i + 1];
        if (orc_in_box(kq[6 * qi], kq[6 * qi + 1], boxes, nb, mode) || orc_in_box(kt[6 * ti], kt[6 * ti + 1], boxes, nb, mode)) {
            dropped++;
        } else {
            o[3 * kept] = m[3 * i]; o[3 * kept + 1] = m[3 * i + 1]; o[3 * kept + 2] = m[3 * i + 2];
            kept++;
        }
    }
    if (filtered) *filtered = dropped;
    return kept;
}

}  // extern "C"
