"""ctypes loader for libaria_orb_hip.so (the C-ABI declared in include/aria_orb_hip.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ARIA_ORB_HIP_LIBRARY: tests and profiling tools point this binding at the variants build (libaria_orb_hip_variants.so,
# `make -C aria_slam_amd/csrc variants`), the only library that knows the ARIA_* kernel / tuning switches. The product
# library itself reads no environment variable.
_SO = os.environ.get("ARIA_ORB_HIP_LIBRARY") or os.path.join(_HERE, "libaria_orb_hip.so")
_SO_VARIANTS = os.path.join(_HERE, "libaria_orb_hip_variants.so")

# byte-for-byte aria::core::KeyPoint / aria::core::Match (reference include/core/Types.hpp:9-15, :97-101)
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4")])
MATCH_DTYPE = np.dtype([("query_idx", "<i4"), ("train_idx", "<i4"), ("distance", "<f4")])

ARIA_OK = 0
ARIA_E_OUTPUT_TOO_SMALL = -5
ARIA_E_NOT_PENDING = -8

# every symbol include/aria_orb_hip.h declares
EXPORTS = [
    "aria_status_string", "aria_abi_version", "aria_last_hip_error",
    "aria_orb_default_config", "aria_orb_create", "aria_orb_destroy", "aria_orb_set_max_features",
    "aria_orb_get_max_features", "aria_orb_kp_capacity", "aria_orb_rows_needed", "aria_orb_fetch_last", "aria_orb_extract", "aria_orb_extract_async",
    "aria_orb_sync", "aria_orb_extract_batch_device", "aria_orb_check", "aria_orb_stream", "aria_orb_slow_path_blocks",
    "aria_orb_set_profiling", "aria_orb_set_stage_event", "aria_orb_get_profile", "aria_matcher_set_profiling", "aria_matcher_get_profile",
    "aria_orb_level_info", "aria_orb_resize_table", "aria_orb_pyramid_bands", "aria_orb_debug_read_level", "aria_orb_algorithmic_bytes",
    "aria_matcher_default_config", "aria_matcher_create", "aria_matcher_destroy", "aria_matcher_match",
    "aria_matcher_knn2", "aria_matcher_match_batch_device", "aria_matcher_match_db_device",
    "aria_matcher_stream", "aria_matcher_sync", "aria_synth_frame_pair", "aria_synth_sequence",
    "aria_flag_keypoints_device", "aria_matcher_match_batch_filtered_device", "aria_matcher_match_multi", "aria_matcher_count_good_multi", "aria_kfdb_create", "aria_kfdb_destroy", "aria_kfdb_size",
    "aria_kfdb_add", "aria_kfdb_add_device", "aria_kfdb_info", "aria_kfdb_fetch", "aria_kfdb_scan",
    "aria_orb_last_device", "aria_matcher_match_device", "aria_matcher_retain_device", "aria_matcher_resident_rows",
    "aria_matcher_match_device_async", "aria_matcher_finish", "aria_kfdb_match", "aria_stream_create", "aria_stream_destroy",
    "aria_orb_fast_blur_kernel", "aria_flag_keypoints_shifted_device",
    # ABI 4: device memory, staging copies and events for hosts that do not link the HIP runtime (host/ BatchFrontEnd)
    "aria_device_count", "aria_device_alloc", "aria_device_free", "aria_host_alloc_pinned", "aria_host_free_pinned",
    "aria_copy_h2d_async", "aria_copy_d2h_async", "aria_copy_d2d_async", "aria_fill_async", "aria_stream_synchronize",
    "aria_event_create", "aria_event_destroy", "aria_event_record", "aria_stream_wait_event", "aria_event_synchronize",
    "aria_event_elapsed_ms", "aria_matcher_knn_kernel",
]


class OrbConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int), ("device", C.c_int), ("stream", C.c_void_p), ("max_width", C.c_int),
                ("max_height", C.c_int), ("max_features", C.c_int), ("max_batch", C.c_int),
                ("blur_tie_mode", C.c_int), ("cand_cap_scale", C.c_int), ("level_size_mode", C.c_int)]


class MatcherConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int), ("device", C.c_int), ("stream", C.c_void_p), ("max_query", C.c_int),
                ("max_train", C.c_int)]


class AriaError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        detail = ""
        try:
            detail = load_library().aria_last_hip_error().decode()
        except Exception:
            pass
        super().__init__("%s failed: %s (%d)%s" % (where, status_string(status), status,
                                                   (" [" + detail + "]") if detail else ""))


def library_path():
    return _SO


def build_library(force=False):
    """Compile the gfx950 shared library in-tree with hipcc (cross-compiles without a GPU)."""
    if force and os.path.exists(_SO):
        os.remove(_SO)
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s"])
    return _SO


def build_variants_library():
    """Compile libaria_orb_hip_variants.so (-DARIA_VARIANTS: superseded kernels + ARIA_* switches) and return its path."""
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s", "variants"])
    return _SO_VARIANTS


_lib = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.

    PyTorch-ROCm wheels bundle their own libamdhip64.so; this library links the system one. If both copies end up in
    one process (library loaded first, torch imported later) the second runtime finds no device. So when a torch
    installation is present, map its copy first: the loader then satisfies our NEEDED libamdhip64.so.7 from it and a
    later `import torch` reuses it too. Without torch nothing happens and the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if not spec or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """Load the HIP library. Never falls back to anything else: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback." % _SO)
    _preload_torch_hip_runtime()
    L = C.CDLL(_SO)
    L.aria_status_string.restype = C.c_char_p
    L.aria_last_hip_error.restype = C.c_char_p
    L.aria_orb_stream.restype = C.c_void_p
    L.aria_matcher_stream.restype = C.c_void_p
    L.aria_orb_destroy.restype = None
    L.aria_matcher_destroy.restype = None
    L.aria_orb_destroy.argtypes = [C.c_void_p]
    L.aria_matcher_destroy.argtypes = [C.c_void_p]
    L.aria_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_int, C.POINTER(C.c_int)]
    L.aria_orb_extract_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.aria_orb_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.aria_orb_extract_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64,
                                                C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.aria_orb_check.argtypes = [C.c_void_p]
    L.aria_orb_slow_path_blocks.argtypes = [C.c_void_p, C.c_int]
    L.aria_orb_slow_path_blocks.restype = C.c_longlong
    L.aria_orb_stream.argtypes = [C.c_void_p]
    L.aria_orb_set_max_features.argtypes = [C.c_void_p, C.c_int]
    L.aria_orb_get_max_features.argtypes = [C.c_void_p]
    L.aria_orb_kp_capacity.argtypes = [C.c_void_p]
    L.aria_orb_rows_needed.argtypes = [C.c_void_p]
    L.aria_orb_fetch_last.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.aria_orb_level_info.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int), C.POINTER(C.c_float)]
    L.aria_orb_resize_table.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.aria_orb_debug_read_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.aria_orb_algorithmic_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.aria_matcher_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p,
                                     C.c_int, C.POINTER(C.c_int)]
    L.aria_orb_fast_blur_kernel.argtypes = [C.c_void_p]
    L.aria_orb_fast_blur_kernel.restype = C.c_char_p
    L.aria_matcher_knn_kernel.argtypes = [C.c_void_p]
    L.aria_matcher_knn_kernel.restype = C.c_char_p
    L.aria_orb_last_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.aria_matcher_match_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p,
                                            C.c_int, C.POINTER(C.c_int)]
    L.aria_matcher_retain_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.aria_matcher_resident_rows.argtypes = [C.c_void_p]
    L.aria_matcher_match_device_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
    L.aria_matcher_finish.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.aria_matcher_knn2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.aria_matcher_match_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                  C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_int]
    L.aria_matcher_match_db_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_int64, C.c_double, C.c_void_p]
    L.aria_flag_keypoints_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.c_void_p]
    L.aria_flag_keypoints_shifted_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                     C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.aria_matcher_match_batch_filtered_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                           C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                                           C.c_void_p, C.c_int, C.c_void_p]
    L.aria_orb_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.aria_orb_set_stage_event.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.aria_orb_get_profile.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int64)]
    L.aria_matcher_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.aria_matcher_get_profile.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_int64)]
    L.aria_matcher_stream.argtypes = [C.c_void_p]
    L.aria_matcher_sync.argtypes = [C.c_void_p]
    L.aria_synth_frame_pair.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.aria_synth_sequence.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    _lib = L
    return L


def status_string(status):
    return load_library().aria_status_string(int(status)).decode()


def abi_version():
    return load_library().aria_abi_version()


def check(status, where):
    if status != ARIA_OK:
        raise AriaError(status, where)


def level_info(max_features, width, height):
    """[(level_width, level_height, quota, scale)] x 8 -- host-only plan geometry."""
    L = load_library()
    out = []
    for l in range(8):
        lw, lh, q, s = C.c_int(), C.c_int(), C.c_int(), C.c_float()
        check(L.aria_orb_level_info(max_features, width, height, l, C.byref(lw), C.byref(lh), C.byref(q), C.byref(s)),
              "aria_orb_level_info")
        out.append((lw.value, lh.value, q.value, s.value))
    return out


def resize_table(width, height, level, axis):
    """(offsets, next-pixel weights in 1/256) of the INTER_LINEAR_EXACT table of `level` along `axis`."""
    buf = np.zeros(4096, np.uint32)
    n = load_library().aria_orb_resize_table(width, height, level, axis, buf.ctypes.data, len(buf))
    if n < 0:
        raise AriaError(n, "aria_orb_resize_table")
    return (buf[:n] & 0xFFFF).astype(np.int32), (buf[:n] >> 16).astype(np.int32)


def pyramid_bands(width, height):
    """(bands[nb, 8, 4] = {comp_lo, comp_n, own_lo, own_n}, band_rows, lds_bytes) of the fused pyramid kernel."""
    buf = np.zeros(300 * 32, np.int32)
    bh, lds = C.c_int(), C.c_int()
    L = load_library()
    L.aria_orb_pyramid_bands.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    n = L.aria_orb_pyramid_bands(width, height, buf.ctypes.data, len(buf), C.byref(bh), C.byref(lds))
    if n < 0:
        raise AriaError(n, "aria_orb_pyramid_bands")
    return buf[:n].reshape(-1, 8, 4).copy(), bh.value, lds.value


def algorithmic_bytes(width, height, n_keypoints):
    be, bf = C.c_int64(), C.c_int64()
    check(load_library().aria_orb_algorithmic_bytes(width, height, n_keypoints, C.byref(be), C.byref(bf)),
          "aria_orb_algorithmic_bytes")
    return be.value, bf.value


def synth_frame_pair(seed, width=640, height=480):
    """Synthetic frame pair of SURVEY.md 8(d): B shows A's content moved by (+3, +2) px."""
    a = np.empty((height, width), np.uint8)
    b = np.empty((height, width), np.uint8)
    check(load_library().aria_synth_frame_pair(seed, width, height, a.ctypes.data, b.ctypes.data), "aria_synth_frame_pair")
    return a, b


def synth_sequence(seed0, n_pairs, width=640, height=480, out=None, n_threads=None):
    """2*n_pairs frames A(seed0), B(seed0), A(seed0+1), ... as one (2*n_pairs, H, W) uint8 array."""
    if out is None:
        out = np.empty((2 * n_pairs, height, width), np.uint8)
    assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and out.size == 2 * n_pairs * height * width
    if n_threads is None:
        n_threads = max(1, min(32, len(os.sched_getaffinity(0))))
    check(load_library().aria_synth_sequence(seed0, n_pairs, width, height, out.ctypes.data, n_threads), "aria_synth_sequence")
    return out
