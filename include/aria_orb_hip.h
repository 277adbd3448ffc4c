/*
 * aria_orb_hip.h -- C-ABI of the MI355X-native ORB extractor + brute-force Hamming matcher.
 *
 * This is the drop-in boundary for aria-slam's feature front-end. Each entry point names
 * the reference interface it replaces; paths are relative to the reference tree (robertteleng/aria-slam).
 * Plain pointers and sizes only: no C++ types, no torch types, never throws. All functions return an
 * aria_status (0 = OK, negative = error); aria_status_string() explains a code.
 *
 * Conventions
 *  - Images are 8-bit grayscale, row-major (include/interfaces/IFeatureExtractor.hpp:14).
 *  - aria_keypoint / aria_match are byte-for-byte aria::core::KeyPoint / aria::core::Match
 *    (include/core/Types.hpp:9-15, :97-101); descriptors are N x 32 bytes, row-major (Types.hpp:25,29).
 *  - The ORB configuration is the one the reference hard-codes (src/adapters/gpu/OrbCudaExtractor.cpp:35-45):
 *    scaleFactor 1.2f, 8 levels, edgeThreshold 31, firstLevel 0, WTA_K 2, HARRIS_SCORE, patchSize 31,
 *    fastThreshold 20; only nfeatures is settable, as in the reference (setMaxFeatures, :212-216).
 *  - Results follow CPU cv::ORB::detectAndCompute (src/legacy/Frame.cpp:45-49), not cv::cuda::ORB.
 *    Keypoint order is canonical: level ascending; within a level response (Harris) descending, then y, then
 *    x ascending. A level may return more than its quota when responses tie at the cut (OpenCV keeps ties),
 *    so a frame can yield slightly more than max_features keypoints; size outputs with aria_orb_kp_capacity().
 *  - A handle is single-owner (not thread-safe), like the reference adapters
 *    (include/adapters/gpu/OrbCudaExtractor.hpp:38-50). Independent handles on different devices/streams
 *    may run concurrently. The library never returns memory it owns; callers allocate every output.
 *  - "device" pointers are HIP device pointers on the handle's device; "stream" is a hipStream_t passed as
 *    void* (borrowed; NULL = the handle creates and owns one: OrbCudaExtractor.cpp:24-29,48-52).
 */
#ifndef ARIA_ORB_HIP_H
#define ARIA_ORB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARIA_ORB_HIP_ABI_VERSION 4

typedef enum {
    ARIA_OK = 0,
    ARIA_E_INVALID = -1,          /* bad argument (null pointer, size out of range, bad struct_size)        */
    ARIA_E_NO_DEVICE = -2,        /* no usable HIP device (none present, bad ordinal, driver/runtime not initialised) */
    ARIA_E_OOM = -3,              /* device or host allocation failed                                       */
    ARIA_E_TOO_LARGE = -4,        /* image larger than the handle was created for                           */
    ARIA_E_OUTPUT_TOO_SMALL = -5, /* caller's keypoint/match capacity is smaller than the result            */
    ARIA_E_OVERFLOW = -6,         /* an internal buffer overflowed; results are NOT valid. Either a FAST candidate
                                     list (only possible with cand_cap_scale > 0: raise it or use 0), or the
                                     tie-storm arenas: selection ties beyond the on-chip capacities are handled
                                     in global memory (k_select_ovf), whose arenas hold the worst case of four
                                     whole frames per internal pass -- a batch with more all-tied frames than
                                     that needs a smaller max_batch.                                         */
    ARIA_E_BUSY = -7,             /* extract_async called while another async extract is pending            */
    ARIA_E_NOT_PENDING = -8,      /* sync called with nothing pending (treated as a no-op by the adapter)   */
    ARIA_E_HIP = -9,              /* a HIP runtime call failed (bad pointer, bad stream, ...): aria_last_hip_error() */
    ARIA_E_KERNEL = -10           /* a kernel launch failed or faulted on the device: aria_last_hip_error()  */
} aria_status;

/* == aria::core::KeyPoint (include/core/Types.hpp:9-15), 24 bytes */
typedef struct { float x, y, size, angle, response; int octave; } aria_keypoint;
/* == aria::core::Match (include/core/Types.hpp:97-101), 12 bytes */
typedef struct { int query_idx, train_idx; float distance; } aria_match;

typedef struct aria_orb_s* aria_orb_t;
typedef struct aria_matcher_s* aria_matcher_t;

/* Replaces the constructor arguments of OrbCudaExtractor (OrbCudaExtractor.cpp:21-46) and
 * FactoryConfig{cuda_device, max_features} (include/factory/PipelineFactory.hpp:16-28). */
typedef struct {
    int   struct_size;     /* = sizeof(aria_orb_config)                                                     */
    int   device;          /* HIP device ordinal                                                            */
    void* stream;          /* borrowed hipStream_t, or NULL = the handle creates (and owns) a stream. The legacy
                            * default stream has handle 0 and therefore cannot be borrowed: two handles that must run
                            * in order (extract, then match on its descriptors) need one real stream between them,
                            * or a sync (aria_orb_check) in between.                                            */
    int   max_width;       /* largest image the handle must accept                                          */
    int   max_height;
    int   max_features;    /* nfeatures (OrbCudaExtractor.hpp:12 default 1000)                              */
    int   max_batch;       /* frames processed per internal pass of the batch entry point (>= 1)            */
    int   blur_tie_mode;   /* rounding of exact ties in the 7x7 blur's column filter, which in OpenCV depends on the SIMD
                            * width its dispatcher picks: 1 (default) ties to even for columns x < (w & ~3) and up in the
                            * scalar tail; 2 / 3: the vector body ends at w & ~7 / w & ~15; 0: ties up everywhere       */
    int   cand_cap_scale;  /* 0 (default): FAST candidate lists sized for the worst case, cannot overflow;
                              > 0: cap each level's list at cand_cap_scale * quota entries to save HBM          */
    int   level_size_mode; /* pyramid level size: 0 (default) cvRound(dim * (1.0f / scale)), 1 cvRound(dim / scale) -- the
                              two readings of cv::ORB's sizing; they differ for few sizes (tools/level_size_sweep.py,
                              profiles/level_size_sweep.txt: none of the BASELINE sizes)                            */
} aria_orb_config;

const char* aria_status_string(int status);
int         aria_abi_version(void);
/* Last HIP runtime error text seen by this thread's most recent failing call ("" if none). */
const char* aria_last_hip_error(void);

void aria_orb_default_config(aria_orb_config* cfg);
int  aria_orb_create(const aria_orb_config* cfg, aria_orb_t* out);
void aria_orb_destroy(aria_orb_t h);

/* IFeatureExtractor::setMaxFeatures / getMaxFeatures (include/interfaces/IFeatureExtractor.hpp:38-39,
 * OrbCudaExtractor.cpp:212-216). All other ORB parameters keep the reference's values. */
int aria_orb_set_max_features(aria_orb_t h, int n);
int aria_orb_get_max_features(aria_orb_t h);
/* Rows to allocate per frame for keypoints/descriptors: sum over levels of (quota + 64 rows of tie slack). OpenCV's
 * retainBest keeps EVERY keypoint tying with the last kept one, so a tie storm (checkerboards, synthetic
 * patterns) can return more: the host entry points then report ARIA_E_OUTPUT_TOO_SMALL with *n_out = rows required
 * (retry with that capacity, or call aria_orb_fetch_last), the batch entry point makes aria_orb_check return
 * ARIA_E_OUTPUT_TOO_SMALL and aria_orb_rows_needed() tell the rows its largest frame needs. */
int aria_orb_kp_capacity(aria_orb_t h);
int aria_orb_rows_needed(aria_orb_t h);
/* Copies the result of the last completed aria_orb_extract / aria_orb_sync out again (it stays in the handle until the
 * next extraction): what a caller uses after ARIA_E_OUTPUT_TOO_SMALL instead of extracting again. */
int aria_orb_fetch_last(aria_orb_t h, aria_keypoint* keypoints, uint8_t* descriptors, int cap, int* n_out);

/* IFeatureExtractor::extract (IFeatureExtractor.hpp:18-23; OrbCudaExtractor.cpp:64-128). Host buffers.
 * image is borrowed and never written; stride = bytes between rows (width for the reference's packed Mat,
 * OrbCudaExtractor.cpp:72). Writes up to cap keypoints (24 B each) and cap*32 descriptor bytes; *n_out = count.
 * Returns ARIA_E_OUTPUT_TOO_SMALL (with *n_out = required) if cap is too small. Blocks until done. */
int aria_orb_extract(aria_orb_t h, const uint8_t* image, int width, int height, int stride,
                     aria_keypoint* keypoints, uint8_t* descriptors, int cap, int* n_out);

/* IFeatureExtractor::extractAsync + sync (IFeatureExtractor.hpp:27-35; OrbCudaExtractor.cpp:136-210).
 * One pending operation per handle, as in the reference. The image buffer must stay valid until
 * aria_orb_sync returns (the upload is asynchronous, :158). */
int aria_orb_extract_async(aria_orb_t h, const uint8_t* image, int width, int height, int stride);
int aria_orb_sync(aria_orb_t h, aria_keypoint* keypoints, uint8_t* descriptors, int cap, int* n_out);

/* OrbCudaExtractor::getGpuDescriptors() (include/adapters/gpu/OrbCudaExtractor.hpp:34-35, "get descriptors without
 * download (for GPU matching)"): device pointers to the result of the single-frame entry points, for a matcher on the same
 * device (aria_matcher_match_device*). The block belongs to the handle: the pointers stay the same from call to call
 * (until a tie storm makes the handle grow it, or aria_orb_set_max_features) and the CONTENT is that of the most recent
 * aria_orb_extract / aria_orb_extract_async, complete once that call / aria_orb_sync has returned -- or, for work queued
 * behind it on the handle's stream, as soon as the extraction's kernels have run. *d_count is the device copy of the
 * keypoint count (clamped to *rows = the block's row capacity), *n the host copy (-1 while an async extract is pending,
 * or before the first extraction). Any output pointer may be NULL. */
int aria_orb_last_device(aria_orb_t h, const aria_keypoint** d_keypoints, const uint8_t** d_descriptors, const int** d_count,
                         int* n, int* rows);

/* Device-resident batch form: the role OrbCudaExtractor::getGpuDescriptors() was meant to play
 * (include/adapters/gpu/OrbCudaExtractor.hpp:35) -- results stay in HBM for the matcher.
 *   d_images     : n_frames images, frame f at d_images + f*frame_stride, rows row_stride bytes apart
 *   d_keypoints  : n_frames * kp_cap records (frame f at index f*kp_cap)
 *   d_descriptors: n_frames * kp_cap * 32 bytes
 *   d_counts     : n_frames ints, keypoints found per frame
 * Enqueued on the handle's stream; returns without synchronising. Call aria_orb_check() after a stream
 * sync to learn whether any frame overflowed an internal buffer or kp_cap. */
int aria_orb_extract_batch_device(aria_orb_t h, const uint8_t* d_images, int n_frames, int width, int height,
                                  int64_t frame_stride, int row_stride,
                                  aria_keypoint* d_keypoints, uint8_t* d_descriptors, int* d_counts, int kp_cap);
/* Synchronises the handle's stream and returns ARIA_OK or the first deferred error (overflow flags). */
int aria_orb_check(aria_orb_t h);
void* aria_orb_stream(aria_orb_t h);
/* Diagnostics: FAST+blur workgroups whose survivor queue overflowed (corner-dense image regions) and that therefore
 * took the slower dense-rescoring path -- results are identical either way. Counted when aria_orb_extract /
 * aria_orb_sync / aria_orb_check read the device state; the handle enlarges the queue for later calls by itself. */
long long aria_orb_slow_path_blocks(aria_orb_t h, int reset);

/* Diagnostics: name of the FAST/blur kernel the handle's most recent pass launched ("k_fast_blur_stream" for batches whose
 * plan and source alignment allow the streaming kernel, "k_fast_blur_band" for the single-frame schedule and the rest). */
const char* aria_orb_fast_blur_kernel(aria_orb_t h);

/* Per-stage device timing for bench.py's roofline figure (no reference counterpart): when enabled, HIP events
 * are recorded on the handle's stream around each stage of every internal pass.
 * Stages: 0 pyramid resize (7 launches per pass), 1 FAST+NMS+blur, 2 select (retainBest+Harris), 3 describe
 * (IC angle + rBRIEF). get_profile synchronises the stream; times are summed milliseconds since the last reset.
 * `enable`: 0 off; 1 every stage; any other even value brackets only the stages s whose bit (s + 1) is set (each
 * bracket drains the stream before and after the stage, ~20 us of idle GPU, so a throughput run times only what it
 * reports). The matcher's aria_matcher_set_profiling takes the same encoding over its 2 stages. */
#define ARIA_ORB_STAGES 4
int aria_orb_set_profiling(aria_orb_t h, int enable);
int aria_orb_get_profile(aria_orb_t h, int reset, double* stage_ms /*[4]*/, int64_t* stage_launches /*[4]*/,
                         int64_t* frames);

/* Stream-ordering hook for callers that pipeline other work beside a batch extraction (no reference counterpart):
 * `event` (a hipEvent_t, or NULL to clear) is recorded on the handle's stream right BEFORE stage `stage` (numbering as
 * above) of the last internal pass of every aria_orb_extract_batch_device call. A second stream that waits on it starts
 * when the FAST/blur launches are done -- bench.py lets the matcher of the previous step run beside select + describe
 * rather than beside the FAST/blur kernel. Only stage 2 (select) is accepted so far (ARIA_E_INVALID otherwise). The
 * event stays owned by the caller and must outlive the calls that record it. */
int aria_orb_set_stage_event(aria_orb_t h, int stage, void* event);

/* Introspection for parity tests and benchmarks (no reference counterpart).
 * Host-only geometry (no handle, no GPU needed): size, quota and scale of pyramid level `level` in [0, 8) for
 * a width x height image, as CPU cv::ORB lays it out; and the fixed-point INTER_LINEAR_EXACT coefficient table
 * of level >= 1 along axis 0 (x) or 1 (y): entry d = source offset | (weight_of_next_pixel_in_1/256 << 16).
 * aria_orb_resize_table returns the number of entries written (or a negative status). */
int aria_orb_level_info(int max_features, int width, int height, int level, int* lw, int* lh, int* quota, float* scale);
int aria_orb_resize_table(int width, int height, int level, int axis, uint32_t* out, int cap);
/* Row schedule of the fused pyramid kernel (host-only, for tests): per band of level-0 rows and per level, four ints
 * {first row computed, rows computed, first row owned, rows owned}. Returns the number of ints written. */
int aria_orb_pyramid_bands(int width, int height, int* out, int cap, int* band_rows, int* lds_bytes);
/* Algorithmic bytes of one frame: b_extract = 5P - p0 - p7 + 56N, b_fused = 2P + 56N. */
int aria_orb_algorithmic_bytes(int width, int height, int n_keypoints, int64_t* b_extract, int64_t* b_fused);
/* Copies level `level` (raw or blurred) of frame 0 of the most recent call to host memory (lw*lh bytes, packed). */
int aria_orb_debug_read_level(aria_orb_t h, int level, int blurred, uint8_t* host_out);

/* ---- matcher: replaces CudaMatcher (include/adapters/gpu/CudaMatcher.hpp, src/adapters/gpu/CudaMatcher.cpp) -- */
typedef struct {
    int   struct_size;
    int   device;
    void* stream;      /* borrowed hipStream_t or NULL (CudaMatcher.cpp:9-17,22-26) */
    int   max_query;   /* rows per descriptor set the host entry points must accept */
    int   max_train;
} aria_matcher_config;

void aria_matcher_default_config(aria_matcher_config* cfg);
int  aria_matcher_create(const aria_matcher_config* cfg, aria_matcher_t* out);
void aria_matcher_destroy(aria_matcher_t m);

/* IMatcher::match (include/interfaces/IMatcher.hpp:19-24; CudaMatcher.cpp:28-68). Host buffers.
 * Brute-force Hamming kNN (k = 2) of every query row against every train row (ties: lower train index
 * first, as CPU cv::BFMatcher), then Lowe's test d0 < ratio*d1 in fp32 (CudaMatcher.cpp:60). Matches are
 * written in query order. nq == 0 or nt == 0 -> *n_out = 0 (CudaMatcher.cpp:35-37). nt == 1 -> no matches
 * (knn.size() >= 2 fails, :60). ratio == 0 means "ratio test disabled" as IMatcher.hpp:18 documents:
 * the best match of every query is returned (the reference adapter would return nothing).
 * Frame-to-frame use: the handle keeps the query set of the previous call on the device; when train_desc holds the
 * same bytes (frame i matched against frame i-1) it is not uploaded again. Purely an optimisation -- the result never
 * depends on it -- but it means a handle serves one caller at a time, like every other entry point here. */
int aria_matcher_match(aria_matcher_t m, const uint8_t* query_desc, int nq, const uint8_t* train_desc, int nt,
               float ratio, aria_match* matches, int cap, int* n_out);

/* CudaMatcher::matchGpu (include/adapters/gpu/CudaMatcher.hpp:22-28, "GPU-to-GPU matching, zero-copy when used with
 * OrbCudaExtractor"): the descriptor sets are device pointers (e.g. aria_orb_last_device), the matches come back to the
 * host; same results as aria_matcher_match on the same rows. The handle keeps ONE descriptor set resident between calls
 * (device copy): a NULL d_query or d_train stands for that resident set, whose row count must then equal nq / nt. After
 * the call the resident set is the one passed as a non-NULL pointer (the query when both are given) -- frame i against
 * frame i-1 is match_device(d_cur, n_cur, NULL, n_prev) in SlamPipeline's order (query = current) or
 * match_device(NULL, n_prev, d_cur, n_cur) in the legacy executables' order (query = previous, src/euroc_eval.cpp:168-169).
 * aria_matcher_retain_device makes a set resident without matching (first frame); aria_matcher_resident_rows tells its
 * rows (-1: none). aria_matcher_match (host buffers) also replaces the resident set, by its query. */
int aria_matcher_match_device(aria_matcher_t m, const uint8_t* d_query, int nq, const uint8_t* d_train, int nt, float ratio,
                              aria_match* matches, int cap, int* n_out);
int aria_matcher_retain_device(aria_matcher_t m, const uint8_t* d_desc, int n);
int aria_matcher_resident_rows(aria_matcher_t m);
/* Pipelined form for a caller whose extractor and matcher share one stream (extractAsync ... sync, the shape of
 * SlamPipeline's async variant, docs/milestones/H12_CLEAN_ARCHITECTURE.md:711-716): the match of the NEW set (d_new, row count
 * read on the device from *d_n_new <= n_new_max, e.g. aria_orb_last_device's d_count) against the resident set is queued
 * on the matcher's stream without waiting -- behind the extraction when both handles were given the same stream -- and
 * aria_matcher_finish synchronises, is told the new set's row count (which the caller knows by then) and returns the
 * matches. new_is_query: 1 = query is the new set, 0 = query is the resident set. One pending operation per handle
 * (ARIA_E_BUSY); aria_matcher_finish without one returns ARIA_E_NOT_PENDING. The kernels read the new set where it lies
 * and the copy that keeps it resident is queued BEHIND the result copy (aria_matcher_finish does not wait for it): whatever
 * overwrites d_new next must be queued on the same stream -- true for the extractor handle that shares it. */
int aria_matcher_match_device_async(aria_matcher_t m, const uint8_t* d_new, const int* d_n_new, int n_new_max, int new_is_query,
                                    float ratio);
int aria_matcher_finish(aria_matcher_t m, int n_new, aria_match* matches, int cap, int* n_out);

/* Raw kNN-2 (cv::BFMatcher::knnMatch(k=2) itself), host buffers: idx/dist hold 2 ints per query
 * (nearest, second nearest); idx = -1 / dist = INT_MAX where the train set is too small. */
int aria_matcher_knn2(aria_matcher_t m, const uint8_t* query_desc, int nq, const uint8_t* train_desc, int nt,
                    int* idx, int* dist);

/* Device-resident batch form: the role CudaMatcher::matchGpu was declared for (CudaMatcher.hpp:23-28).
 * Pair p matches query block (d_query + p*desc_stride bytes, d_nq[p] rows) against train block
 * (d_train + p*desc_stride, d_nt[p] rows); writes up to match_cap matches at d_matches + p*match_cap and the
 * count at d_nmatches[p]. Enqueued on the matcher's stream; no synchronisation. */
int aria_matcher_match_batch_device(aria_matcher_t m, const uint8_t* d_query, const int* d_nq,
                            const uint8_t* d_train, const int* d_nt, int n_pairs, int64_t desc_stride,
                            float ratio, aria_match* d_matches, int* d_nmatches, int match_cap);

/* ---- dynamic-object filter: src/main.cpp:42-50 isInDynamicObject + :164-175, the hook
 * SlamPipeline::filterDynamicKeypoints was declared for (include/pipeline/SlamPipeline.hpp:96-99). The detector is out of
 * scope; its boxes are an input (the caller passes the boxes of dynamic classes, main.cpp:29-40). Step 1, between describe
 * and match: flags[f*kp_cap + i] = 1 when keypoint i of frame f lies in one of frame f's boxes (frame f's boxes at
 * d_boxes + f*box_cap, d_nboxes[f] of them). mode 0 = the legacy test, cv::Rect::contains of the keypoint rounded to an
 * integer point (half to even): x1 <= round(x) < x2, y1 <= round(y) < y2; mode 1 = core::Detection::contains
 * (include/core/Types.hpp:109-111): closed float intervals. Step 2: the batched matcher drops every ratio-test survivor
 * with a flagged endpoint and counts them (main.cpp's filtered_count) -- kNN-2 itself still sees every keypoint, as in the
 * reference. flag_stride = flags per frame block (>= desc_stride / 32); pair p's query / train flags at
 * d_qflags / d_tflags + p*flag_stride. */
typedef struct { float x1, y1, x2, y2; } aria_box;
int aria_flag_keypoints_device(void* stream, const aria_keypoint* d_keypoints, const int* d_counts, int n_frames, int kp_cap,
                               const aria_box* d_boxes, const int* d_nboxes, int box_cap, int mode, uint8_t* d_flags);
/* The reference's semantics for pairs of consecutive frames. main.cpp:164-175 tests BOTH endpoints of a match against the
 * detections of the CURRENT frame, so the train side of pair (f + 1, f) -- frame f's keypoints -- must be flagged against
 * frame f + 1's boxes, not its own (which is what the call above gives when one flag array serves as query flags of one pair
 * and train flags of the next). box_frame_offset = +1 produces exactly those train flags into a second array (frames whose
 * f + offset does not exist get no flag); offset 0 is the call above. Batched use with the reference's semantics:
 *   aria_flag_keypoints_device(...,          d_qflags);          // query flags: frame f vs boxes f
 *   aria_flag_keypoints_shifted_device(..., +1, d_tflags);       // train flags: frame f vs boxes f + 1
 *   aria_matcher_match_batch_filtered_device(m, desc + stride, cnt + 1, desc, cnt, B - 1, stride, ratio,
 *                                            d_qflags + kp_cap, d_tflags, kp_cap, ...);   // pair p: query f = p + 1, train f = p */
int aria_flag_keypoints_shifted_device(void* stream, const aria_keypoint* d_keypoints, const int* d_counts, int n_frames, int kp_cap,
                                       const aria_box* d_boxes, const int* d_nboxes, int box_cap, int mode, int box_frame_offset,
                                       uint8_t* d_flags);
int aria_matcher_match_batch_filtered_device(aria_matcher_t m, const uint8_t* d_query, const int* d_nq, const uint8_t* d_train,
                                             const int* d_nt, int n_pairs, int64_t desc_stride, float ratio,
                                             const uint8_t* d_qflags, const uint8_t* d_tflags, int64_t flag_stride,
                                             aria_match* d_matches, int* d_nmatches, int match_cap, int* d_nfiltered);

/* Loop-closure candidate scan, the semantic behind IMatcher::matchMultiple (IMatcher.hpp:27-37) as the
 * legacy code uses it (src/legacy/LoopClosure.cpp:72-114): one query descriptor set against n_kf keyframe
 * blocks resident in HBM (block k at d_db + k*desc_stride bytes, d_kf_counts[k] rows). For every keyframe:
 * kNN-2, ratio test in double (d0 < ratio*d1, LoopClosure.cpp:92), d_good[k] = number of passing queries.
 * Scoring/top-5 (LoopClosure.cpp:98-111) is host logic in the adapter. */
int aria_matcher_match_db_device(aria_matcher_t m, const uint8_t* d_query, int nq, const uint8_t* d_db,
                         const int* d_kf_counts, int n_kf, int64_t desc_stride, double ratio, int* d_good);
/* IMatcher::matchMultiple (include/interfaces/IMatcher.hpp:27-37) in one batch: the query is uploaded once, every
 * candidate's descriptors go to a device staging area back to back, ONE kNN-2 launch covers all candidates, one
 * download. Candidate c's matches are written at matches + c*cap_per_cand, its count at n_out[c]; per candidate the
 * result equals aria_matcher_match(query, candidate c). Host buffers. */
int aria_matcher_match_multi(aria_matcher_t m, const uint8_t* query_desc, int nq, const uint8_t* const* train_descs,
                             const int* nts, int n_cand, float ratio, aria_match* matches, int cap_per_cand, int* n_out);
/* The scan of LoopClosureDetector::findCandidates (src/legacy/LoopClosure.cpp:79-96) over host-resident candidates:
 * good[c] = number of queries whose two nearest neighbours in candidate c pass d0 < ratio*d1 in double. One launch. */
int aria_matcher_count_good_multi(aria_matcher_t m, const uint8_t* query_desc, int nq, const uint8_t* const* train_descs,
                                  const int* nts, int n_cand, double ratio, int* good);

/* ---- HBM-resident keyframe descriptor database: the deque of LoopClosureDetector (src/legacy/LoopClosure.cpp:24-31;
 * docs/milestones/H14_GPU_LOOPCLOSURE_AUDIT.md designs exactly this). Fixed-capacity slots of `rows` descriptors;
 * adding beyond `capacity` drops the oldest keyframe (pop_front, :28-30). Index i below = position in the deque,
 * oldest first. The scan is one kernel launch over the whole database. */
typedef struct aria_kfdb_s* aria_kfdb_t;
int  aria_kfdb_create(int device, void* stream, int capacity, int rows, aria_kfdb_t* out);
void aria_kfdb_destroy(aria_kfdb_t db);
int  aria_kfdb_size(aria_kfdb_t db);
int  aria_kfdb_add(aria_kfdb_t db, long long id, const uint8_t* desc_host, int n);
int  aria_kfdb_add_device(aria_kfdb_t db, long long id, const uint8_t* d_desc, int n);
int  aria_kfdb_info(aria_kfdb_t db, int index, long long* id, int* count);
int  aria_kfdb_fetch(aria_kfdb_t db, int index, uint8_t* desc_host, int cap_rows, int* n_out);
/* The ratio-test match list of the query against keyframe `index` (LoopClosure.cpp:120-131, the list verifyGeometry starts
 * from), matched where the keyframe lies in HBM; same result as aria_matcher_match(query, fetched keyframe). Host query. */
int  aria_kfdb_match(aria_kfdb_t db, aria_matcher_t m, int index, const uint8_t* query_desc, int nq, float ratio,
                     aria_match* matches, int cap, int* n_out);
/* good[i] for every keyframe i (ratio test in double, LoopClosure.cpp:92); *n_out = keyframes. Host query. */
int  aria_kfdb_scan(aria_kfdb_t db, aria_matcher_t m, const uint8_t* query_desc, int nq, double ratio, int* good, int cap,
                    int* n_out);

/* Same for the matcher: stage 0 = kNN-2 kernel, stage 1 = ratio test + ordered compaction. */
#define ARIA_MATCHER_STAGES 2
int aria_matcher_set_profiling(aria_matcher_t m, int enable);
int aria_matcher_get_profile(aria_matcher_t m, int reset, double* stage_ms /*[2]*/, int64_t* stage_launches /*[2]*/,
                             int64_t* pairs);
/* Diagnostics (like aria_orb_fast_blur_kernel): kernel form of the handle's most recent batch / database kNN-2 launch --
 * "k_knn2_fp4" (FP4 matrix path, train sets <= 4096 rows), "k_knn2_mfma" (int8, wider train sets), or
 * "k_knn2_fp4|k_knn2_mfma" when both were launched behind the device-side gate (the FP4 one does the batch unless some pair's
 * train count exceeds 4096). bench.py keys the label and the peak of roofline.matcher on it. */
const char* aria_matcher_knn_kernel(aria_matcher_t m);
void* aria_matcher_stream(aria_matcher_t m);
int   aria_matcher_sync(aria_matcher_t m);

/* cudaStreamCreate / cudaStreamDestroy as the reference adapters use them (OrbCudaExtractor.cpp:28,50; CudaMatcher.cpp:16,24)
 * for hosts that do not link the HIP runtime: a stream to pass as aria_orb_config.stream AND aria_matcher_config.stream,
 * which orders the two handles' work (needed by aria_matcher_match_device_async). Destroy it after the handles. */
int aria_stream_create(int device, void** stream);
int aria_stream_destroy(int device, void* stream);

/* ---- device memory, staging copies and events (ABI 4) for a host that drives the BATCH entry points from the reference's
 * language without linking the HIP runtime (aria_slam_amd/host BatchFrontEnd, euroc_frontend --batch). What the reference
 * does through cv::cuda::GpuMat / the CUDA runtime on this path: device allocation + upload per frame
 * (src/legacy/Frame.cpp:19, src/adapters/gpu/OrbCudaExtractor.cpp:83, :158 upload on the stream), download of the results
 * (OrbCudaExtractor.cpp:102-103, :186-187), cudaStreamSynchronize (src/euroc_eval.cpp:153-154). Plain pointers and sizes;
 * every call returns an aria_status (HIP failures as ARIA_E_HIP with the text in aria_last_hip_error()).
 * Pinned host memory is what makes the copies asynchronous (a pageable source is staged by the runtime, synchronously).
 * Events order two streams (the copy stream fills chunk c + 1 while the compute stream works on chunk c) and time them. */
int aria_device_count(int* n);
int aria_device_alloc(int device, size_t bytes, void** d_ptr);
int aria_device_free(int device, void* d_ptr);
int aria_host_alloc_pinned(size_t bytes, void** h_ptr);
int aria_host_free_pinned(void* h_ptr);
int aria_copy_h2d_async(int device, void* stream, void* d_dst, const void* h_src, size_t bytes);
int aria_copy_d2h_async(int device, void* stream, void* h_dst, const void* d_src, size_t bytes);
int aria_copy_d2d_async(int device, void* stream, void* d_dst, const void* d_src, size_t bytes);
int aria_fill_async(int device, void* stream, void* d_dst, int byte_value, size_t bytes);
int aria_stream_synchronize(int device, void* stream);
int aria_event_create(int device, void** event);
int aria_event_destroy(int device, void* event);
int aria_event_record(int device, void* event, void* stream);
int aria_stream_wait_event(int device, void* stream, void* event);
int aria_event_synchronize(int device, void* event);
int aria_event_elapsed_ms(void* start_event, void* stop_event, float* ms);

/* ---- synthetic workload: integer-only generator, identical bytes everywhere ------------ */
int aria_synth_frame_pair(uint64_t seed, int width, int height, uint8_t* frame_a, uint8_t* frame_b);
int aria_synth_sequence(uint64_t seed0, int n_pairs, int width, int height, uint8_t* out, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
