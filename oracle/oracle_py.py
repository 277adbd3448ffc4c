"""ctypes binding of the CPU oracle (oracle/liborb_oracle.so).  TEST INFRASTRUCTURE ONLY -- parity unpinned.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/orb_oracle.h). The product package aria_slam_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liborb_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4")])
MATCH_DTYPE = np.dtype([("query_idx", "<i4"), ("train_idx", "<i4"), ("distance", "<f4")])


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("fast_threshold", C.c_int), ("blur_tie_mode", C.c_int), ("level_size_mode", C.c_int)]


def build():
    """Compile the oracle if the shared object is missing or stale."""
    src = [os.path.join(_HERE, f) for f in ("orb_oracle.cpp", "orb_oracle.h", "orb_pattern_31.inc")]
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, ip, fp, dp = (C.POINTER(C.c_uint8), C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double))
        L.orc_layer_scale.restype = C.c_float
        L.orc_harris_response.restype = C.c_float
        L.orc_ic_angle.restype = C.c_float
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_sincos.argtypes = [C.c_double, dp, dp]
        L.orc_pyramid_layout.restype = C.c_int64
        L.orc_bit_pattern_31.restype = ip
        _lib = L
    return _lib


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def default_params(nfeatures=1000, blur_tie_mode=1, level_size_mode=0):
    p = Params()
    lib().orc_default_params(C.byref(p))
    p.nfeatures = nfeatures
    p.blur_tie_mode = blur_tie_mode
    p.level_size_mode = level_size_mode
    return p


def level_sizes(p, w, h):
    out = []
    for l in range(p.nlevels):
        lw, lh = C.c_int(), C.c_int()
        lib().orc_level_size(C.byref(p), w, h, l, C.byref(lw), C.byref(lh))
        out.append((lw.value, lh.value))
    return out


def layer_scales(p):
    return [lib().orc_layer_scale(C.byref(p), l) for l in range(p.nlevels)]


def feature_quotas(p):
    q = np.zeros(p.nlevels, np.int32)
    lib().orc_feature_quotas(C.byref(p), _i32(q))
    return q.tolist()


def resize_coeffs(ssize, dsize):
    ofs = np.zeros(dsize, np.int32)
    c1 = np.zeros(dsize, np.int32)
    lib().orc_resize_coeffs(ssize, dsize, _i32(ofs), _i32(c1))
    return ofs, c1


def resize_linear_exact(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw = src.shape
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_exact(_u8(src), sw, sh, sw, _u8(dst), dw, dh, dw)
    return dst


def fast_score_map(img, threshold=20):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    s = np.zeros((h, w), np.uint8)
    lib().orc_fast_score_map(_u8(img), w, h, w, threshold, _u8(s))
    return s


def fast_detect(img, threshold=20):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = w * h // 4 + 16
    xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib().orc_fast_detect(_u8(img), w, h, w, threshold, _i32(xs), _i32(ys), _i32(sc), cap)
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def gaussian_kernel7_fixed():
    k = np.zeros(7, np.int32)
    lib().orc_gaussian_kernel7_fixed(_i32(k))
    return k


def gaussian_blur7(img, tie_mode=1):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_gaussian_blur7(_u8(img), w, h, w, _u8(out), w, tie_mode)
    return out


def harris_response(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    return float(lib().orc_harris_response(_u8(img), img.shape[1], int(x), int(y)))


def ic_moments(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    m01, m10 = C.c_int(), C.c_int()
    lib().orc_ic_moments(_u8(img), img.shape[1], int(x), int(y), C.byref(m01), C.byref(m10))
    return m01.value, m10.value


def ic_angle(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    return float(lib().orc_ic_angle(_u8(img), img.shape[1], int(x), int(y)))


def fast_atan2(y, x):
    return float(lib().orc_fast_atan2(C.c_float(y), C.c_float(x)))


def umax():
    u = np.zeros(16, np.int32)
    lib().orc_umax(_i32(u))
    return u


def sincos(x):
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos(C.c_double(x), C.byref(s), C.byref(c))
    return s.value, c.value


def brief_descriptor(blurred, x, y, angle_deg):
    blurred = np.ascontiguousarray(blurred, np.uint8)
    d = np.zeros(32, np.uint8)
    lib().orc_brief_descriptor(_u8(blurred), blurred.shape[1], int(x), int(y), C.c_float(angle_deg), _u8(d))
    return d


def bit_pattern_31():
    return np.ctypeslib.as_array(lib().orc_bit_pattern_31(), shape=(1024,)).reshape(256, 4).copy()


def detect_level(img, quota, fast_threshold=20):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = w * h // 4 + 16
    xs, ys = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    hr = np.zeros(cap, np.float32)
    nf, nr = C.c_int(), C.c_int()
    n = lib().orc_detect_level(_u8(img), w, h, w, quota, fast_threshold, _i32(xs), _i32(ys),
                               hr.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(nf), C.byref(nr))
    return xs[:n].copy(), ys[:n].copy(), hr[:n].copy(), nf.value, nr.value


def pyramid_layout(p, w, h):
    lw = (C.c_int * 16)()
    lh = (C.c_int * 16)()
    offs = (C.c_int64 * 16)()
    total = lib().orc_pyramid_layout(C.byref(p), w, h, lw, lh, offs)
    n = p.nlevels
    return total, list(lw)[:n], list(lh)[:n], list(offs)[:n]


def build_pyramid(img, p):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    total, lw, lh, offs = pyramid_layout(p, w, h)
    out = np.zeros(total, np.uint8)
    lib().orc_build_pyramid(_u8(img), w, h, w, C.byref(p), _u8(out))
    return [out[offs[l]:offs[l] + lw[l] * lh[l]].reshape(lh[l], lw[l]) for l in range(p.nlevels)]


def blur_pyramid(levels, p, w, h):
    total, lw, lh, offs = pyramid_layout(p, w, h)
    flat = np.concatenate([np.ascontiguousarray(l, np.uint8).ravel() for l in levels])
    out = np.zeros(total, np.uint8)
    lib().orc_blur_pyramid(_u8(flat), w, h, C.byref(p), _u8(out))
    return [out[offs[l]:offs[l] + lw[l] * lh[l]].reshape(lh[l], lw[l]) for l in range(p.nlevels)]


def orb_extract(img, p, cap=None):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    if cap is None:
        cap = p.nfeatures * 2 + 256
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = C.c_int()
    rc = lib().orc_orb_extract(_u8(img), w, h, w, C.byref(p), kps.ctypes.data_as(C.c_void_p), _u8(desc), cap, C.byref(n))
    if rc != 0:
        raise RuntimeError("orc_orb_extract rc=%d need=%d" % (rc, n.value))
    return kps[:n.value].copy(), desc[:n.value].copy()


def knn2(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.zeros((len(q), 2), np.int32)
    dist = np.zeros((len(q), 2), np.int32)
    lib().orc_knn2(_u8(q), len(q), _u8(t), len(t), _i32(idx), _i32(dist))
    return idx, dist


def match_ratio(q, t, ratio=0.75):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    out = np.zeros(max(len(q), 1), MATCH_DTYPE)
    n = lib().orc_match_ratio(_u8(q), len(q), _u8(t), len(t), C.c_float(ratio), out.ctypes.data_as(C.c_void_p))
    return out[:n].copy()


def count_good_matches_f64(q, t, ratio=0.7):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    return lib().orc_count_good_matches_f64(_u8(q), len(q), _u8(t), len(t), C.c_double(ratio))


def loop_candidates(q, query_id, db_blocks, kf_ids, min_frames_between):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    counts = np.array([len(b) for b in db_blocks], np.int32)
    db = np.concatenate([np.ascontiguousarray(b, np.uint8).reshape(-1, 32) for b in db_blocks]) if db_blocks else np.zeros((0, 32), np.uint8)
    ids = np.array(kf_ids, np.int64)
    ci = np.zeros(5, np.int32)
    cs = np.zeros(5, np.float64)
    n = lib().orc_loop_candidates(_u8(q), len(q), C.c_int64(query_id), _u8(db), _i32(counts),
                                  ids.ctypes.data_as(C.POINTER(C.c_int64)), len(db_blocks), min_frames_between,
                                  _i32(ci), cs.ctypes.data_as(C.POINTER(C.c_double)))
    return ci[:n].copy(), cs[:n].copy()


def filter_dynamic_matches(kps_q, kps_t, matches, boxes, mode=0):
    """src/main.cpp:42-50, 164-175 (mode 0) / core::Detection::contains (mode 1). Returns (kept matches, filtered count)."""
    kq = np.ascontiguousarray(kps_q)
    kt = np.ascontiguousarray(kps_t)
    m = np.ascontiguousarray(matches)
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
    out = np.zeros(len(m), MATCH_DTYPE)
    filt = C.c_int()
    n = lib().orc_filter_dynamic_matches(kq.ctypes.data_as(C.c_void_p), kt.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p),
                                         len(m), b.ctypes.data_as(C.POINTER(C.c_float)), len(b), mode,
                                         out.ctypes.data_as(C.c_void_p), C.byref(filt))
    return out[:n].copy(), filt.value
