"""Python mirror of the reference's extractor / matcher ports, bound to the C-ABI.

Names and argument meaning follow the reference interfaces so that the parity tests read like tests of the
reference adapters would:
  IFeatureExtractor  extract / extractAsync / sync / setMaxFeatures / getMaxFeatures
                     (reference include/interfaces/IFeatureExtractor.hpp:18-39,
                      src/adapters/gpu/OrbCudaExtractor.cpp:64-216)
  IMatcher           match / matchMultiple
                     (reference include/interfaces/IMatcher.hpp:19-37, src/adapters/gpu/CudaMatcher.cpp:28-68)
A "frame" here is a dict with the fields the extractor fills in aria::core::Frame (include/core/Types.hpp:18-31):
width, height, keypoints (structured array, KP_DTYPE), descriptors (N x 32 uint8).
The C++ adapters under aria_slam_amd/host are the drop-in classes; this module only serves tests and bench.py.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, MATCH_DTYPE, AriaError, check


def _ptr(x):
    """Device pointer of a torch tensor, or an int passed through."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    return x.data_ptr()


class OrbHipExtractor:
    """Replaces aria::adapters::gpu::OrbCudaExtractor (reference include/adapters/gpu/OrbCudaExtractor.hpp)."""

    def __init__(self, max_features=1000, stream=None, device=0, max_width=640, max_height=480, max_batch=1,
                 blur_tie_mode=1, cand_cap_scale=0, level_size_mode=0):
        self._L = _lib.load_library()
        cfg = _lib.OrbConfig()
        self._L.aria_orb_default_config(C.byref(cfg))
        cfg.device = device
        cfg.stream = stream
        cfg.max_width, cfg.max_height = max_width, max_height
        cfg.max_features = max_features
        cfg.max_batch = max_batch
        cfg.blur_tie_mode = blur_tie_mode
        cfg.cand_cap_scale = cand_cap_scale
        cfg.level_size_mode = level_size_mode
        h = C.c_void_p()
        check(self._L.aria_orb_create(C.byref(cfg), C.byref(h)), "aria_orb_create")
        self._h = h
        self._pending = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.aria_orb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- IFeatureExtractor ----
    def extract(self, image_data, width=None, height=None, frame=None):
        img = np.ascontiguousarray(image_data, np.uint8)
        if width is None:
            height, width = img.shape
        cap = self.kp_capacity()
        kps = np.empty(cap, KP_DTYPE)
        desc = np.empty((cap, 32), np.uint8)
        n = C.c_int()
        st = self._L.aria_orb_extract(self._h, img.ctypes.data, width, height, width, kps.ctypes.data,
                                      desc.ctypes.data, cap, C.byref(n))
        if st == _lib.ARIA_E_OUTPUT_TOO_SMALL:
            kps, desc = self._fetch_last(n.value)       # a tie storm returned more keypoints than the plan's rows
        else:
            check(st, "aria_orb_extract")
        return self._fill(frame, width, height, kps[:n.value].copy(), desc[:n.value].copy())

    def _fetch_last(self, rows):
        kps = np.empty(rows, KP_DTYPE)
        desc = np.empty((rows, 32), np.uint8)
        n = C.c_int()
        check(self._L.aria_orb_fetch_last(self._h, kps.ctypes.data, desc.ctypes.data, rows, C.byref(n)), "aria_orb_fetch_last")
        return kps, desc

    def extractAsync(self, image_data, width=None, height=None, frame=None):
        img = np.ascontiguousarray(image_data, np.uint8)
        if width is None:
            height, width = img.shape
        check(self._L.aria_orb_extract_async(self._h, img.ctypes.data, width, height, width), "aria_orb_extract_async")
        self._pending = (img, width, height, frame if frame is not None else {})
        return self._pending[3]

    def sync(self):
        if self._pending is None:   # OrbCudaExtractor.cpp:177 "if (!pending_frame_) return;"
            return None
        img, width, height, frame = self._pending
        self._pending = None
        cap = self.kp_capacity()
        kps = np.empty(cap, KP_DTYPE)
        desc = np.empty((cap, 32), np.uint8)
        n = C.c_int()
        st = self._L.aria_orb_sync(self._h, kps.ctypes.data, desc.ctypes.data, cap, C.byref(n))
        if st == _lib.ARIA_E_OUTPUT_TOO_SMALL:
            kps, desc = self._fetch_last(n.value)
        else:
            check(st, "aria_orb_sync")
        return self._fill(frame, width, height, kps[:n.value].copy(), desc[:n.value].copy())

    def setMaxFeatures(self, n):
        check(self._L.aria_orb_set_max_features(self._h, int(n)), "aria_orb_set_max_features")

    def getMaxFeatures(self):
        return self._L.aria_orb_get_max_features(self._h)

    # ---- OrbCudaExtractor::getGpuDescriptors (include/adapters/gpu/OrbCudaExtractor.hpp:34-35) ----
    def device_result(self):
        """(d_keypoints, d_descriptors, d_count, n, rows): raw device pointers of the single-frame result block, the host
        copy of the count (-1 while an async extract is pending) and the block's row capacity."""
        kp, ds, cnt = C.c_void_p(), C.c_void_p(), C.c_void_p()
        n, rows = C.c_int(), C.c_int()
        check(self._L.aria_orb_last_device(self._h, C.byref(kp), C.byref(ds), C.byref(cnt), C.byref(n), C.byref(rows)),
              "aria_orb_last_device")
        return kp.value, ds.value, cnt.value, n.value, rows.value

    # ---- device-resident batch form ----
    def kp_capacity(self):
        return self._L.aria_orb_kp_capacity(self._h)

    def extract_batch_device(self, d_images, n_frames, width, height, d_keypoints, d_descriptors, d_counts, kp_cap,
                             frame_stride=None, row_stride=None):
        row_stride = width if row_stride is None else row_stride
        frame_stride = row_stride * height if frame_stride is None else frame_stride
        check(self._L.aria_orb_extract_batch_device(self._h, _ptr(d_images), n_frames, width, height, frame_stride,
                                                    row_stride, _ptr(d_keypoints), _ptr(d_descriptors),
                                                    _ptr(d_counts), kp_cap), "aria_orb_extract_batch_device")

    def check(self):
        check(self._L.aria_orb_check(self._h), "aria_orb_check")

    def rows_needed(self):
        """Rows the largest frame of the last checked batch call needed when kp_cap was too small (else 0)."""
        return self._L.aria_orb_rows_needed(self._h)

    def fast_blur_kernel(self):
        """Name of the FAST/blur kernel the most recent pass launched."""
        return self._L.aria_orb_fast_blur_kernel(self._h).decode()

    def slow_path_blocks(self, reset=False):
        return self._L.aria_orb_slow_path_blocks(self._h, int(reset))

    STAGES = ("resize", "fast_blur", "select", "describe")

    def set_stage_event(self, stage, event_handle):
        """Record `event_handle` (a raw hipEvent_t, e.g. torch.cuda.Event().cuda_event; None clears) on the stream right
        before `stage` ("select") of the last pass of every batch call."""
        check(self._L.aria_orb_set_stage_event(self._h, {"select": 2}[stage], event_handle), "aria_orb_set_stage_event")

    def set_profiling(self, enable, stages=None):
        """stages: optional subset of STAGES to bracket (each bracket drains the stream twice); default all."""
        v = int(bool(enable))
        if enable and stages is not None:
            v = sum(1 << (self.STAGES.index(s) + 1) for s in stages)
        check(self._L.aria_orb_set_profiling(self._h, v), "aria_orb_set_profiling")

    def get_profile(self, reset=True):
        """{stage: (total_ms, launches)}, frames -- HIP-event times on the launch stream."""
        ms = (C.c_double * 4)()
        ln = (C.c_int64 * 4)()
        fr = C.c_int64()
        check(self._L.aria_orb_get_profile(self._h, int(reset), ms, ln, C.byref(fr)), "aria_orb_get_profile")
        return {s: (ms[i], ln[i]) for i, s in enumerate(self.STAGES)}, fr.value

    @property
    def stream(self):
        return self._L.aria_orb_stream(self._h)

    # ---- introspection for parity tests ----
    def level_info(self, width, height):
        return _lib.level_info(self.getMaxFeatures(), width, height)

    def debug_read_level(self, level, blurred, lw, lh):
        out = np.empty((lh, lw), np.uint8)
        check(self._L.aria_orb_debug_read_level(self._h, level, int(bool(blurred)), out.ctypes.data),
              "aria_orb_debug_read_level")
        return out

    def algorithmic_bytes(self, width, height, n_keypoints):
        return _lib.algorithmic_bytes(width, height, n_keypoints)

    @staticmethod
    def _fill(frame, width, height, kps, desc):
        frame = {} if frame is None else frame
        frame["width"], frame["height"] = width, height       # OrbCudaExtractor.cpp:109-110
        frame["keypoints"] = kps                               # :111-123 (cleared, then refilled)
        frame["descriptors"] = desc                            # :126-127
        return frame


class HipMatcher:
    """Replaces aria::adapters::gpu::CudaMatcher (reference include/adapters/gpu/CudaMatcher.hpp).
    stream: a real hipStream_t handle to borrow; None or 0 (torch's default stream) makes the handle create its own
    stream, which is NOT ordered against other handles' streams -- share one explicit stream, or sync in between."""

    def __init__(self, stream=None, device=0, max_query=4096, max_train=4096):
        self._L = _lib.load_library()
        cfg = _lib.MatcherConfig()
        self._L.aria_matcher_default_config(C.byref(cfg))
        cfg.device, cfg.stream, cfg.max_query, cfg.max_train = device, stream, max_query, max_train
        h = C.c_void_p()
        check(self._L.aria_matcher_create(C.byref(cfg), C.byref(h)), "aria_matcher_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.aria_matcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _desc(frame_or_array):
        d = frame_or_array["descriptors"] if isinstance(frame_or_array, dict) else frame_or_array
        return np.ascontiguousarray(d, np.uint8).reshape(-1, 32)

    # ---- IMatcher ----
    def match(self, query, train, matches=None, ratio_threshold=0.75):
        """Appends to `matches` (a list) like CudaMatcher::match appends to the vector (CudaMatcher.cpp:65)."""
        q, t = self._desc(query), self._desc(train)
        out = np.empty(max(len(q), 1), MATCH_DTYPE)
        n = C.c_int()
        check(self._L.aria_matcher_match(self._h, q.ctypes.data, len(q), t.ctypes.data, len(t),
                                         C.c_float(ratio_threshold), out.ctypes.data, len(out), C.byref(n)),
              "aria_matcher_match")
        res = out[:n.value].copy()
        if matches is not None:
            matches.extend(res.tolist())
        return res

    def matchMultiple(self, query, candidates, all_matches=None, ratio_threshold=0.75):
        """IMatcher.hpp:27-37: resize the outer list, then match against every candidate."""
        out = [self.match(query, c, None, ratio_threshold) for c in candidates]
        if all_matches is not None:
            del all_matches[len(candidates):]
            while len(all_matches) < len(candidates):
                all_matches.append([])
            for dst, src in zip(all_matches, out):
                dst.extend(src.tolist())
        return out

    def knn2(self, query, train):
        q, t = self._desc(query), self._desc(train)
        idx = np.empty((len(q), 2), np.int32)
        dist = np.empty((len(q), 2), np.int32)
        check(self._L.aria_matcher_knn2(self._h, q.ctypes.data, len(q), t.ctypes.data if len(t) else None, len(t),
                                        idx.ctypes.data, dist.ctypes.data), "aria_matcher_knn2")
        return idx, dist

    # ---- CudaMatcher::matchGpu (include/adapters/gpu/CudaMatcher.hpp:22-28): descriptors on the device ----
    def match_device(self, d_query, nq, d_train, nt, ratio_threshold=0.75):
        """d_query / d_train: raw device pointers (int) or None = the set kept resident from the previous call."""
        out = np.empty(max(nq, 1), MATCH_DTYPE)
        n = C.c_int()
        check(self._L.aria_matcher_match_device(self._h, _ptr(d_query), nq, _ptr(d_train), nt, C.c_float(ratio_threshold),
                                                out.ctypes.data, len(out), C.byref(n)), "aria_matcher_match_device")
        return out[:n.value].copy()

    def retain_device(self, d_desc, n):
        check(self._L.aria_matcher_retain_device(self._h, _ptr(d_desc), n), "aria_matcher_retain_device")

    def resident_rows(self):
        return self._L.aria_matcher_resident_rows(self._h)

    def match_device_async(self, d_new, d_n_new, n_new_max, new_is_query=True, ratio_threshold=0.75):
        check(self._L.aria_matcher_match_device_async(self._h, _ptr(d_new), _ptr(d_n_new), n_new_max, int(bool(new_is_query)),
                                                      C.c_float(ratio_threshold)), "aria_matcher_match_device_async")

    def finish(self, n_new, cap):
        out = np.empty(max(cap, 1), MATCH_DTYPE)
        n = C.c_int()
        check(self._L.aria_matcher_finish(self._h, n_new, out.ctypes.data, len(out), C.byref(n)), "aria_matcher_finish")
        return out[:n.value].copy()

    # ---- device-resident batch forms ----
    def match_batch_device(self, d_query, d_nq, d_train, d_nt, n_pairs, desc_stride, ratio, d_matches, d_nmatches,
                           match_cap):
        check(self._L.aria_matcher_match_batch_device(self._h, _ptr(d_query), _ptr(d_nq), _ptr(d_train), _ptr(d_nt),
                                                      n_pairs, desc_stride, C.c_float(ratio), _ptr(d_matches),
                                                      _ptr(d_nmatches), match_cap), "aria_matcher_match_batch_device")

    def match_batch_filtered_device(self, d_query, d_nq, d_train, d_nt, n_pairs, desc_stride, ratio, d_qflags, d_tflags,
                                    flag_stride, d_matches, d_nmatches, match_cap, d_nfiltered):
        """main.cpp:164-175 on the device: ratio-test survivors with an endpoint flagged by flag_keypoints_device are dropped."""
        check(self._L.aria_matcher_match_batch_filtered_device(
            self._h, _ptr(d_query), _ptr(d_nq), _ptr(d_train), _ptr(d_nt), n_pairs, desc_stride, ratio,
            _ptr(d_qflags), _ptr(d_tflags), flag_stride, _ptr(d_matches), _ptr(d_nmatches), match_cap,
            _ptr(d_nfiltered)), "aria_matcher_match_batch_filtered_device")

    def match_db_device(self, d_query, nq, d_db, d_kf_counts, n_kf, desc_stride, ratio, d_good):
        check(self._L.aria_matcher_match_db_device(self._h, _ptr(d_query), nq, _ptr(d_db), _ptr(d_kf_counts), n_kf,
                                                   desc_stride, C.c_double(ratio), _ptr(d_good)),
              "aria_matcher_match_db_device")

    def sync(self):
        check(self._L.aria_matcher_sync(self._h), "aria_matcher_sync")

    def knn_kernel(self):
        """Kernel form of the handle's most recent batch / database kNN-2 launch (aria_matcher_knn_kernel)."""
        return self._L.aria_matcher_knn_kernel(self._h).decode()

    STAGES = ("knn2", "ratio_compact")

    def set_profiling(self, enable, stages=None):
        v = int(bool(enable))
        if enable and stages is not None:
            v = sum(1 << (self.STAGES.index(s) + 1) for s in stages)
        check(self._L.aria_matcher_set_profiling(self._h, v), "aria_matcher_set_profiling")

    def get_profile(self, reset=True):
        ms = (C.c_double * 2)()
        ln = (C.c_int64 * 2)()
        pr = C.c_int64()
        check(self._L.aria_matcher_get_profile(self._h, int(reset), ms, ln, C.byref(pr)), "aria_matcher_get_profile")
        return {s: (ms[i], ln[i]) for i, s in enumerate(self.STAGES)}, pr.value

    @property
    def stream(self):
        return self._L.aria_matcher_stream(self._h)


def flag_keypoints_device(stream, d_keypoints, d_counts, n_frames, kp_cap, d_boxes, d_nboxes, box_cap, mode, d_flags,
                          box_frame_offset=0):
    """main.cpp:42-50 for every keypoint of every frame (boxes are an input; mode 0 legacy cv::Rect test, 1 Detection::contains).
    box_frame_offset = +1: frame f's keypoints against frame f + 1's boxes (the train flags of pair (f + 1, f) in the
    reference's semantics, main.cpp:164-175)."""
    L = _lib.load_library()
    check(L.aria_flag_keypoints_shifted_device(stream, _ptr(d_keypoints), _ptr(d_counts), n_frames, kp_cap, _ptr(d_boxes),
                                               _ptr(d_nboxes), box_cap, mode, box_frame_offset, _ptr(d_flags)),
          "aria_flag_keypoints_shifted_device")
