/*
 * orb_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A dependency-free C++17 restatement of the arithmetic behind the reference's hot path:
 *   cv::ORB::detectAndCompute  (call site: /root/reference src/legacy/Frame.cpp:45-49, parameters fixed at
 *                               src/adapters/gpu/OrbCudaExtractor.cpp:35-45)
 *   cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) + Lowe ratio
 *                              (call sites: src/adapters/gpu/CudaMatcher.cpp:28-68,
 *                               src/legacy/LoopClosure.cpp:72-114)
 * The arithmetic lives in a third-party, un-vendored dependency: OpenCV 4.9.0 (pinned by git tag only at
 * scripts/setup_machine.sh:192-194). OpenCV is absent from this container and from the GPU box, and the
 * reference holds no tests, golden vectors or fixtures for this path, so:
 *
 *      ***  PARITY UNPINNED  ***
 *
 * Every function below restates the published OpenCV 4.9.0 algorithm from knowledge of its source
 * (modules/features2d/src/{orb,fast,fast_score,keypoint}.cpp, modules/imgproc/src/{resize,smooth.dispatch,
 * filter.simd}.cpp, modules/core/src/mathfuncs_core.*), anchored on the reference call sites above. What the
 * tests can prove is "HIP path == this restatement, bit for bit"; "== real OpenCV" is not yet verified.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this library.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 16

/* Same 24-byte record as aria::core::KeyPoint (reference include/core/Types.hpp:9-15). */
typedef struct { float x, y, size, angle, response; int octave; } orc_keypoint;
/* Same 12-byte record as aria::core::Match (reference include/core/Types.hpp:97-101). */
typedef struct { int query_idx, train_idx; float distance; } orc_match;

typedef struct {
    int   nfeatures;      /* OrbCudaExtractor.hpp:12 default 1000 */
    float scale_factor;   /* 1.2f  OrbCudaExtractor.cpp:37 */
    int   nlevels;        /* 8     :38 */
    int   fast_threshold; /* 20    :44 */
    /* edgeThreshold 31, firstLevel 0, WTA_K 2, HARRIS_SCORE, patchSize 31 are fixed (:39-43). */
    int   blur_tie_mode;  /* Where the column filter's vector body (ties to even) ends and its scalar tail (ties up) begins --
                             it depends on the SIMD width OpenCV's dispatcher picks for SymmColumnVec_32s8u:
                             1 = body x < (w & ~3) (loops that go down to 4 lanes: the default here), 2 = x < (w & ~7),
                             3 = x < (w & ~15), 0 = ties up everywhere (a build without SIMD). */
    int   level_size_mode;/* 0 = cvRound(dim * (1.0f / scale)) (default: what orb.cpp is restated as here),
                             1 = cvRound(dim / scale) (SURVEY.md A.1's wording). tools/level_size_sweep.py lists where they differ. */
} orc_params;

void orc_default_params(orc_params* p);

/* ---- geometry ---------------------------------------------------------------------------------------- */
/* layerScale[l] = (float)pow((double)scaleFactor, l)                     [orb.cpp getScale]              */
float orc_layer_scale(const orc_params* p, int level);
/* level size = (cvRound(cols * (1.f/scale)), cvRound(rows * (1.f/scale)))  [orb.cpp detectAndCompute]; level_size_mode 1:
 * cvRound(cols / scale)                                                                                  */
void  orc_level_size(const orc_params* p, int w, int h, int level, int* lw, int* lh);
/* per-level feature quotas                                               [orb.cpp computeKeyPoints]      */
void  orc_feature_quotas(const orc_params* p, int* quota /*[nlevels]*/);

/* ---- stages ------------------------------------------------------------------------------------------ */
/* resize(src, dst, INTER_LINEAR_EXACT), 8-bit single channel             [resize.cpp resize_bitExact]    */
void orc_resize_linear_exact(const uint8_t* src, int sw, int sh, int sstride,
                             uint8_t* dst, int dw, int dh, int dstride);
/* coefficient table of the above for one axis: ofs[d], c1[d] in 1/256 units (c0 = 256 - c1)              */
void orc_resize_coeffs(int ssize, int dsize, int* ofs, int* c1);

/* FAST-9/16 corner score (before non-max suppression) for every pixel; 0 where not a corner.            */
void orc_fast_score_map(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score /*w*h*/);
/* FAST-9/16 with 3x3 non-max suppression; raster-ordered list. Returns the count (may exceed cap;
 * only the first cap entries are written).                               [fast.cpp FAST_t<16>]           */
int  orc_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold,
                     int* xs, int* ys, int* scores, int cap);

/* GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) as OpenCV executes it on an ORB pyramid ROI
 * (8-bit fixed-point separable filter, 8 fractional bits per pass)       [smooth.dispatch.cpp, filter.simd.hpp] */
void orc_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, int tie_mode);
void orc_gaussian_kernel7_fixed(int* k7 /*[7]*/);

/* Harris response, blockSize 7, k = 0.04                                 [orb.cpp HarrisResponses]       */
float orc_harris_response(const uint8_t* img, int stride, int x, int y);
/* intensity-centroid moments and angle                                   [orb.cpp ICAngles]              */
void  orc_ic_moments(const uint8_t* img, int stride, int x, int y, int* m01, int* m10);
float orc_ic_angle(const uint8_t* img, int stride, int x, int y);
float orc_fast_atan2(float y, float x);                               /* [mathfuncs_core fastAtan2]      */
void  orc_umax(int* umax /*[16]*/);
/* deterministic double-precision sin/cos used for the descriptor rotation (see .cpp for the rationale)  */
void  orc_sincos(double x, double* s, double* c);
/* rBRIEF, WTA_K = 2; img = blurred level; (x,y) = integer centre in the level; angle in degrees        */
void  orc_brief_descriptor(const uint8_t* blurred, int stride, int x, int y, float angle_deg, uint8_t* desc32);
const int* orc_bit_pattern_31(void);

/* ---- whole pipeline ---------------------------------------------------------------------------------- */
/* Per-level detection exactly as computeKeyPoints does it for one level, exposed for stage-wise parity:
 * FAST+NMS -> border filter(31) -> retainBest(2*quota, FAST score) -> Harris -> retainBest(quota, Harris).
 * Output in the canonical order (response desc, y asc, x asc). Returns count (<= cap written). */
int orc_detect_level(const uint8_t* img, int w, int h, int stride, int quota, int fast_threshold,
                     int* xs, int* ys, float* harris, int cap,
                     int* n_fast /*after NMS+border*/, int* n_after_first_retain);

/* Full cv::ORB::detectAndCompute(image, noArray(), keypoints, descriptors).
 * Keypoint order is CANONICAL (OpenCV's own order is libstdc++-introselect-defined):
 * level ascending; within a level Harris response descending, then y ascending, then x ascending.
 * Returns 0 on success, -1 if cap is too small (n_out still holds the required count). */
int orc_orb_extract(const uint8_t* img, int w, int h, int stride, const orc_params* p,
                    orc_keypoint* kps, uint8_t* desc /*cap*32*/, int cap, int* n_out);

/* Pyramid access for stage-wise parity tests: total bytes and per-level offsets of a packed buffer
 * (levels stored tightly, stride = level width). */
int64_t orc_pyramid_layout(const orc_params* p, int w, int h, int* lw, int* lh, int64_t* offs);
void orc_build_pyramid(const uint8_t* img, int w, int h, int stride, const orc_params* p, uint8_t* out);
void orc_blur_pyramid(const uint8_t* pyr, int w, int h, const orc_params* p, uint8_t* out);

/* ---- matching ---------------------------------------------------------------------------------------- */
int  orc_hamming256(const uint8_t* a, const uint8_t* b);
/* BFMatcher(NORM_HAMMING).knnMatch(k=2): per query the two nearest train rows; ties -> lower train index.
 * idx[2*i+k], dist[2*i+k]; idx = -1 where fewer than k+1 train rows exist. [batch_distance.cpp]          */
void orc_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int* idx, int* dist);
/* CudaMatcher::match semantics (CudaMatcher.cpp:28-68): fp32 ratio test d0 < ratio*d1, output in query
 * order. ratio == 0 means "disabled" per IMatcher.hpp:18 (best match per query, needs only 1 train row). */
int  orc_match_ratio(const uint8_t* q, int nq, const uint8_t* t, int nt, float ratio, orc_match* out);
/* LoopClosure.cpp:86-98 good-match count with the double-precision ratio literal (0.7).                  */
int  orc_count_good_matches_f64(const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio);
/* LoopClosure.cpp:72-114 findCandidates over a keyframe DB laid out as concatenated descriptor blocks.   */
int  orc_filter_dynamic_matches(const void* kps_q, const void* kps_t, const void* matches, int n, const float* boxes, int nb,
                                int mode, void* out, int* filtered);
int  orc_loop_candidates(const uint8_t* q, int nq, int64_t query_id,
                         const uint8_t* db, const int* kf_counts, const int64_t* kf_ids, int n_kf,
                         int min_frames_between, int* cand_idx /*[5]*/, double* cand_score /*[5]*/);

#ifdef __cplusplus
}
#endif
#endif
