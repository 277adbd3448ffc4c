"""Pins the CPU oracle (oracle/orb_oracle.cpp) with hand-derivable known answers and independent recomputation.

The reference holds no tests, goldens or fixtures for this path and OpenCV is not installed, so
parity is UNPINNED against real OpenCV. What can be pinned here:
  * the published tables of the reference docs' own configurations (level sizes, quotas),
  * tables available from an independent source (scikit-image's copy of the rBRIEF pattern and OFAST_UMAX),
  * per-stage answers derivable by hand or recomputed with an independent numpy/Fraction implementation.
"""
import hashlib
import numpy as np
import pytest


# ---- geometry (computed there independently of this code) ---------------------
SURVEY_LEVELS = {
    (640, 480): [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)],
    (752, 480): [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)],
    (1408, 1408): [(1408, 1408), (1173, 1173), (978, 978), (815, 815), (679, 679), (566, 566), (472, 472), (393, 393)],
}
SURVEY_QUOTAS = {2000: [434, 362, 302, 251, 209, 175, 145, 122], 1000: [217, 181, 151, 126, 105, 87, 73, 60],
                 4000: [869, 724, 603, 503, 419, 349, 291, 242]}
SURVEY_P = {(640, 480): 950532, (752, 480): 1117367, (1408, 1408): 6137732}


@pytest.mark.parametrize("wh", list(SURVEY_LEVELS))
def test_level_sizes_match_survey_table(oracle, wh):
    p = oracle.default_params(1000)
    sizes = oracle.level_sizes(p, *wh)
    assert sizes == SURVEY_LEVELS[wh]
    assert sum(w * h for w, h in sizes) == SURVEY_P[wh]


@pytest.mark.parametrize("nf", list(SURVEY_QUOTAS))
def test_quotas_match_survey_table(oracle, nf):
    q = oracle.feature_quotas(oracle.default_params(nf))
    assert q == SURVEY_QUOTAS[nf] and sum(q) == nf


def test_layer_scale_is_float_pow_of_float_1p2(oracle):
    s = oracle.layer_scales(oracle.default_params())
    base = float(np.float32(1.2))
    for l, v in enumerate(s):
        assert np.float32(v) == np.float32(base ** l)
    assert s[0] == 1.0 and abs(s[7] - 3.5831808) < 1e-5


# ---- tables -----------------------------------------------------------------------------------------------
def test_umax_matches_skimage_ofast_umax(oracle):
    # scikit-image skimage/feature/orb.py OFAST_UMAX (independent source)
    assert oracle.umax().tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


def _pattern_sha_from_product():
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "aria_slam_amd", "csrc", "orb_pattern_31.inc")).read()
    txt = "\n".join(l for l in txt.splitlines() if not l.strip().startswith("//"))
    vals = np.array([int(v) for v in re.findall(r"-?\d+", txt)], np.int8)
    assert vals.size == 1024
    return hashlib.sha256(vals.tobytes()).hexdigest()


PATTERN_SHA = _pattern_sha_from_product()   # oracle copy and product copy must be the same 1024 numbers


def test_bit_pattern_table(oracle):
    pat = oracle.bit_pattern_31()
    assert pat.shape == (256, 4)
    assert pat[:5].tolist() == [[8, -3, 9, 5], [4, 2, 7, -12], [-11, 9, -8, 2], [7, -12, 12, -13], [2, -13, 2, 12]]
    assert pat[255].tolist() == [-1, -6, 0, -11]
    assert pat.min() == -13 and pat.max() == 12
    # sha256 of the 1024 values as int8, so the oracle and product copies can be compared without sharing a file
    assert hashlib.sha256(pat.astype(np.int8).tobytes()).hexdigest() == PATTERN_SHA


# ---- resize ------------------------------------------------------------------------------------------------
def _coeffs_python(ssize, dsize):
    """Independent restatement with Python floats (IEEE double, same operation order as the C++)."""
    inv = float(dsize) / float(ssize)
    scale = 1.0 / inv
    ofs, c1 = [], []
    for d in range(dsize):
        f = scale * (d + 0.5) - 0.5
        i = int(np.floor(f))
        if i >= 0 and ssize > 1:
            if i < ssize - 1:
                ofs.append(i)
                c1.append(int(np.rint((f - i) * 256.0)))
            else:
                ofs.append(ssize - 1)
                c1.append(0)
        else:
            ofs.append(0)
            c1.append(0)
    return np.array(ofs), np.array(c1)


@pytest.mark.parametrize("s,d", [(640, 533), (533, 444), (480, 400), (1408, 1173), (179, 149), (7, 19), (19, 7)])
def test_resize_coeffs(oracle, s, d):
    ofs, c1 = oracle.resize_coeffs(s, d)
    po, pc = _coeffs_python(s, d)
    assert np.array_equal(ofs, po) and np.array_equal(c1, pc)
    assert ofs.min() >= 0 and ofs.max() <= s - 1 and c1.min() >= 0 and c1.max() <= 256
    if (s, d) == (640, 533):
        # fval(0) = (1/(533/640))*0.5 - 0.5 = 0.10037...; 0.10037*256 = 25.69 -> 26
        assert ofs[0] == 0 and c1[0] == 26


def test_resize_exact_rational_reference(oracle):
    """out = (cy0*(cx0*p00+cx1*p01) + cy1*(cx0*p10+cx1*p11) + 2^15) >> 16, checked with exact integers."""
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    dw, dh = 44, 31
    got = oracle.resize_linear_exact(src, dw, dh)
    xo, xc = _coeffs_python(53, dw)
    yo, yc = _coeffs_python(37, dh)
    s = src.astype(np.int64)
    for dy in range(dh):
        for dx in range(dw):
            x0, x1 = xo[dx], min(xo[dx] + 1, 52)
            y0, y1 = yo[dy], min(yo[dy] + 1, 36)
            h0 = (256 - xc[dx]) * s[y0, x0] + xc[dx] * s[y0, x1]
            h1 = (256 - xc[dx]) * s[y1, x0] + xc[dx] * s[y1, x1]
            v = ((256 - yc[dy]) * h0 + yc[dy] * h1 + 32768) >> 16
            assert got[dy, dx] == min(v, 255)


def test_resize_constant_image_is_constant(oracle):
    for c in (0, 1, 127, 255):
        out = oracle.resize_linear_exact(np.full((480, 640), c, np.uint8), 533, 400)
        assert out.min() == c and out.max() == c


# ---- FAST ---------------------------------------------------------------------------------------------------
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _ring_image(center, ring_vals, size=15):
    img = np.full((size, size), center, np.uint8)
    c = size // 2
    for (dx, dy), v in zip(RING, ring_vals):
        img[c + dy, c + dx] = v
    return img, c


def test_fast_nine_contiguous_is_corner_eight_is_not(oracle):
    for start in range(16):
        vals = [100] * 16
        for j in range(9):
            vals[(start + j) % 16] = 160
        img, c = _ring_image(100, vals)
        s = oracle.fast_score_map(img, 20)
        # bright arc of +60: largest threshold keeping it a corner is 59
        assert s[c, c] == 59
        vals8 = [100] * 16
        for j in range(8):
            vals8[(start + j) % 16] = 160
        img8, _ = _ring_image(100, vals8)
        assert oracle.fast_score_map(img8, 20)[c, c] == 0


def test_fast_threshold_is_strict(oracle):
    # ring exactly threshold brighter is NOT a corner (x > v + t is strict); threshold + 1 is, with score == t
    vals = [120] * 9 + [100] * 7
    img, c = _ring_image(100, vals)
    assert oracle.fast_score_map(img, 20)[c, c] == 0
    vals = [121] * 9 + [100] * 7
    img, c = _ring_image(100, vals)
    assert oracle.fast_score_map(img, 20)[c, c] == 20
    vals = [79] * 9 + [100] * 7      # dark arc
    img, c = _ring_image(100, vals)
    assert oracle.fast_score_map(img, 20)[c, c] == 20


def test_fast_score_is_min_over_best_arc(oracle):
    vals = [200, 190, 180, 170, 160, 150, 140, 130, 125, 100, 100, 100, 100, 100, 100, 100]
    img, c = _ring_image(100, vals)
    assert oracle.fast_score_map(img, 20)[c, c] == 24   # min over the only 9-arc is 125-100 = 25 -> 24


def test_fast_nms_keeps_strict_maximum_only(oracle):
    img = np.full((40, 40), 100, np.uint8)
    for (dx, dy), v in zip(RING, [160] * 9 + [100] * 7):
        img[20 + dy, 20 + dx] = v
    xs, ys, sc = oracle.fast_detect(img, 20)
    smap = oracle.fast_score_map(img, 20)
    for x, y, s in zip(xs, ys, sc):
        nb = smap[y - 1:y + 2, x - 1:x + 2].astype(int).copy()
        nb[1, 1] = -1
        assert s == smap[y, x] and s > nb.max()
    assert (20, 20) in set(zip(xs.tolist(), ys.tolist()))
    # raster order
    order = list(zip(ys.tolist(), xs.tolist()))
    assert order == sorted(order)


def test_fast_skips_three_pixel_margin(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (30, 41), dtype=np.uint8)
    s = oracle.fast_score_map(img, 20)
    assert s[:3].max() == 0 and s[-3:].max() == 0 and s[:, :3].max() == 0 and s[:, -3:].max() == 0
    assert s.max() > 0


# ---- Gaussian blur -------------------------------------------------------------------------------------------
def _blur_sums_numpy(img):
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    p = np.pad(img.astype(np.int64), 3, mode="reflect")     # numpy 'reflect' == BORDER_REFLECT_101
    h, w = img.shape
    rows = sum(k[i] * p[:, i:i + w] for i in range(7))
    return sum(k[j] * rows[j:j + h, :] for j in range(7))


def test_gaussian_kernel_fixed_point(oracle):
    k = oracle.gaussian_kernel7_fixed()
    assert k.tolist() == [18, 34, 49, 55, 49, 34, 18] and k.sum() == 257
    # independent: exp(-x^2/8) normalised, float32, times 256, round half even
    t = np.exp(-0.5 * (np.arange(7) - 3.0) ** 2 / 4.0)
    assert np.rint((t / t.sum()).astype(np.float32).astype(np.float64) * 256).astype(int).tolist() == k.tolist()


@pytest.mark.parametrize("shape", [(480, 640), (134, 179), (61, 67)])
def test_gaussian_blur_against_numpy_and_tie_rule(oracle, shape):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    s = _blur_sums_numpy(img)
    q, r = s >> 16, s & 0xFFFF
    up = np.minimum(q + (r >= 32768), 255)                          # ties up everywhere
    h, w = shape
    body = (np.arange(w) < (w & ~3))[None, :]
    tie = r == 32768
    even = np.minimum(np.where(tie & body, q + (q & 1), q + (r >= 32768)), 255)
    assert np.array_equal(oracle.gaussian_blur7(img, 0), up)
    assert np.array_equal(oracle.gaussian_blur7(img, 1), even)


def test_gaussian_blur_tie_modes_differ_only_on_exact_ties(oracle):
    """Ties (sum == 2^15 mod 2^16) occur about once per 65536 pixels; the two rounding modes must differ exactly
    at ties whose quotient is even and whose column lies in the vector body x < (w & ~3)."""
    n_diff = 0
    for seed in range(3):
        rng = np.random.default_rng(100 + seed)
        img = rng.integers(0, 256, (480, 642), dtype=np.uint8)     # 642: two tail columns
        s = _blur_sums_numpy(img)
        q, r = s >> 16, s & 0xFFFF
        expect_diff = (r == 32768) & ((q & 1) == 0) & (np.arange(642) < 640)[None, :] & (q < 255)
        m0, m1 = oracle.gaussian_blur7(img, 0), oracle.gaussian_blur7(img, 1)
        assert np.array_equal(m0 != m1, expect_diff)
        assert np.all(m0[expect_diff].astype(int) - m1[expect_diff].astype(int) == 1)
        n_diff += int(expect_diff.sum())
    assert n_diff >= 1


def tie_stripes(h, w):
    """Image whose 7x7 blur hits EXACT ties in whole columns: rows are identical, period 7 along x with
    18 a + 34 b + 49 c + 55 d + 49 e + 34 f + 18 g = 32768 at one phase (257 * 32768 = 128.5 * 2^16)."""
    pat = np.array([127, 128, 128, 126, 128, 128, 128], np.uint8)
    return np.tile(pat[np.arange(w) % 7][None, :], (h, 1))


@pytest.mark.parametrize("w", [61, 645, 652, 655])
def test_gaussian_blur_vector_tail_modes(oracle, w):
    """blur_tie_mode 1 / 2 / 3 = vector body of the column filter ends at w & ~3 / w & ~7 / w & ~15: on an
    image with whole columns of exact ties the modes differ exactly in the tie columns between two body ends."""
    img = tie_stripes(40, w)
    s = _blur_sums_numpy(img)
    q, r = s >> 16, s & 0xFFFF
    tie = r == 32768
    assert tie[5:-5].all(axis=0).sum() >= w // 7 - 2              # about one column in seven ties, in every interior row
    outs = {}
    for mode, body_end in ((0, 0), (1, w & ~3), (2, w & ~7), (3, w & ~15)):
        body = (np.arange(w) < body_end)[None, :]
        want = np.minimum(np.where(tie & body, q + (q & 1), q + (r >= 32768)), 255)
        outs[mode] = oracle.gaussian_blur7(img, mode)
        assert np.array_equal(outs[mode], want), mode
    for a, b, lo, hi in ((1, 2, w & ~7, w & ~3), (2, 3, w & ~15, w & ~7), (0, 1, 0, w & ~3)):
        diff_cols = np.flatnonzero((outs[a] != outs[b]).any(axis=0))
        assert all(lo <= c < hi for c in diff_cols)
        if tie[:, lo:hi].any():
            assert len(diff_cols) >= 1


def test_level_size_modes_and_sweep_file(oracle):
    """level_size_mode 0 = cvRound(dim * (1.0f / scale)), 1 = cvRound(dim / scale): the oracle follows the switch, and the
    committed sweep (tools/level_size_sweep.py -> profiles/level_size_sweep.txt) lists exactly where they differ."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "profiles", "level_size_sweep.txt")).read()
    fresh = subprocess.run([sys.executable, os.path.join(root, "tools", "level_size_sweep.py")], capture_output=True, text=True).stdout
    assert fresh == txt
    assert "BASELINE.json sizes affected: none" in txt
    rows = [tuple(int(v) for v in l.split()) for l in txt.splitlines() if l and not l.startswith("#")]
    assert len(rows) >= 10
    p0, p1 = oracle.default_params(1000, 1, 0), oracle.default_params(1000, 1, 1)
    listed = {(d, l) for d, l, _, _ in rows}
    for dim, level, a, b in rows[:12] + rows[-12:]:
        assert oracle.level_sizes(p0, dim, 64 + level)[level][0] == a
        assert oracle.level_sizes(p1, dim, 64 + level)[level][0] == b
    for dim in (640, 480, 752, 1408, 333, 251, 100):              # sizes not in the list: both modes agree at every level
        for level in range(8):
            assert (dim, level) in listed or oracle.level_sizes(p0, dim, dim)[level] == oracle.level_sizes(p1, dim, dim)[level]


def test_gaussian_blur_gain_257_over_256_saturates(oracle):
    assert oracle.gaussian_blur7(np.full((40, 40), 255, np.uint8)).min() == 255
    assert oracle.gaussian_blur7(np.full((40, 40), 100, np.uint8)).tolist()[0][0] == (100 * 257 * 257 + 32768) >> 16


# ---- Harris, IC angle, fastAtan2 ----------------------------------------------------------------------------
def test_harris_on_vertical_step_edge(oracle):
    img = np.zeros((31, 31), np.uint8)
    img[:, 16:] = 100
    # block 7x7 centred (15,15): columns 12..18; Ix = 4*(I[x+1]-I[x-1]) -> 400 where the step is within reach
    ix = np.zeros(7, np.int64)
    for j, x in enumerate(range(12, 19)):
        ix[j] = 4 * (int(img[15, x + 1]) - int(img[15, x - 1]))
    a = int((ix ** 2).sum() * 7)
    f = np.float32
    scale = f(1.0) / (f(4 * 7) * f(255.0))
    s4 = scale * scale * scale * scale
    fa = f(a)
    expect = (fa * f(0) - f(0) * f(0) - f(0.04) * (fa + f(0)) * (fa + f(0))) * s4
    got = oracle.harris_response(img, 15, 15)
    assert np.float32(got) == np.float32(expect) and got < 0


def test_harris_on_perfect_corner_is_positive(oracle):
    img = np.zeros((31, 31), np.uint8)
    img[15:, 15:] = 200
    assert oracle.harris_response(img, 15, 15) > 0


def test_ic_moments_of_linear_ramps(oracle):
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    sum_u2 = sum(u * u for v in range(-15, 16) for u in range(-umax[abs(v)], umax[abs(v)] + 1))
    yy, xx = np.mgrid[0:41, 0:41]
    ramp_x = (100 + (xx - 20)).astype(np.uint8)
    m01, m10 = oracle.ic_moments(ramp_x, 20, 20)
    assert (m01, m10) == (0, sum_u2)
    assert oracle.ic_angle(ramp_x, 20, 20) == 0.0
    ramp_y = (100 + (yy - 20)).astype(np.uint8)
    m01, m10 = oracle.ic_moments(ramp_y, 20, 20)
    assert (m01, m10) == (sum_u2, 0)          # the disc is symmetric under x<->y
    assert abs(oracle.ic_angle(ramp_y, 20, 20) - 90.0) < 1e-4
    ramp_nx = (100 - (xx - 20)).astype(np.uint8)
    assert abs(oracle.ic_angle(ramp_nx, 20, 20) - 180.0) < 1e-4


def test_fast_atan2_properties(oracle):
    assert oracle.fast_atan2(0.0, 0.0) == 0.0
    assert oracle.fast_atan2(0.0, 5.0) == 0.0
    assert abs(oracle.fast_atan2(1.0, 1.0) - 45.0) < 0.01
    rng = np.random.default_rng(1)
    for _ in range(2000):
        y, x = rng.normal(size=2) * 1000
        got = oracle.fast_atan2(float(np.float32(y)), float(np.float32(x)))
        want = np.degrees(np.arctan2(y, x)) % 360.0
        d = abs(got - want)
        assert min(d, 360 - d) < 0.02 and 0.0 <= got <= 360.0


def test_fast_atan2_float32_step_by_step(oracle):
    f = np.float32
    k = f(180.0 / np.pi)
    p1, p3, p5, p7 = f(0.9997878412794807) * k, f(-0.3258083974640975) * k, f(0.1555786518463281) * k, f(-0.04432655554792128) * k
    eps = f(np.finfo(np.float64).eps)
    for y, x in [(3.0, 4.0), (-7.0, 2.0), (5.0, -1.0), (-2.0, -9.0), (123456.0, 7.0)]:
        ax, ay = f(abs(x)), f(abs(y))
        if ax >= ay:
            c = ay / (ax + eps)
            c2 = c * c
            a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
        else:
            c = ax / (ay + eps)
            c2 = c * c
            a = f(90.0) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
        if x < 0:
            a = f(180.0) - a
        if y < 0:
            a = f(360.0) - a
        assert f(oracle.fast_atan2(y, x)) == f(a)


def test_sincos_matches_libm_through_float(oracle):
    rng = np.random.default_rng(2)
    ang = np.concatenate([rng.uniform(0, 360, 20000), np.arange(0, 361, 0.25)]).astype(np.float32)
    rad = ang * np.float32(np.pi / np.float32(180.0))
    for r in rad:
        s, c = oracle.sincos(float(r))
        assert np.float32(s) == np.float32(np.sin(np.float64(r))) and np.float32(c) == np.float32(np.cos(np.float64(r)))
        assert abs(s - np.sin(np.float64(r))) < 1e-15 and abs(c - np.cos(np.float64(r))) < 1e-15


# ---- rBRIEF ---------------------------------------------------------------------------------------------------
def test_brief_on_x_ramp_at_angle_zero(oracle):
    pat = oracle.bit_pattern_31()
    yy, xx = np.mgrid[0:61, 0:61]
    img = (xx * 2).astype(np.uint8)
    d = oracle.brief_descriptor(img, 30, 30, 0.0)
    bits = np.unpackbits(d, bitorder="little")
    assert np.array_equal(bits, (pat[:, 0] < pat[:, 2]).astype(np.uint8))
    assert oracle.brief_descriptor(np.full((61, 61), 9, np.uint8), 30, 30, 123.0).max() == 0


def test_brief_rotation_by_90_degrees_swaps_axes(oracle):
    pat = oracle.bit_pattern_31()
    yy, xx = np.mgrid[0:61, 0:61]
    img = (yy * 2).astype(np.uint8)       # value grows with y
    d = oracle.brief_descriptor(img, 30, 30, 90.0)
    # at 90 deg: a = cos ~ -4.4e-8 (float of the deterministic cos), b = 1: y' = x*b + y*a rounds to x
    bits = np.unpackbits(d, bitorder="little")
    assert np.array_equal(bits, (pat[:, 0] < pat[:, 2]).astype(np.uint8))


# ---- matcher -------------------------------------------------------------------------------------------------
def test_hamming_and_knn_ties_prefer_lower_index(oracle):
    z = np.zeros((1, 32), np.uint8)
    o = np.full((1, 32), 255, np.uint8)
    idx, dist = oracle.knn2(z, np.concatenate([o, z, z, o]))
    assert idx.tolist() == [[1, 2]] and dist.tolist() == [[0, 0]]
    idx, dist = oracle.knn2(z, np.concatenate([o, o]))
    assert idx.tolist() == [[0, 1]] and dist.tolist() == [[256, 256]]
    idx, dist = oracle.knn2(z, o)
    assert idx.tolist() == [[0, -1]] and dist[0, 0] == 256


def test_match_ratio_semantics(oracle):
    rng = np.random.default_rng(3)
    t = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    q = t[[5, 17, 30]].copy()
    q[1, 0] ^= 0x0F
    m = oracle.match_ratio(q, t, 0.75)
    assert m["query_idx"].tolist() == [0, 1, 2] and m["train_idx"].tolist() == [5, 17, 30]
    assert m["distance"].tolist() == [0.0, 4.0, 0.0]
    assert len(oracle.match_ratio(q, t[:1], 0.75)) == 0          # CudaMatcher.cpp:60 needs two neighbours
    assert len(oracle.match_ratio(q, t[:0], 0.75)) == 0          # :35-37
    assert len(oracle.match_ratio(q, t[:1], 0.0)) == 3           # IMatcher.hpp:18 ratio 0 = disabled
    # d0 < ratio*d1 is strict: equal best and second-best never pass for ratio <= 1
    tt = np.concatenate([t[:1], t[:1]])
    assert len(oracle.match_ratio(t[:1], tt, 1.0)) == 0


def test_loop_candidates_semantics(oracle):
    rng = np.random.default_rng(4)
    q = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    blocks, ids = [], []
    for k in range(8):
        b = rng.integers(0, 256, (200, 32), dtype=np.uint8)
        n_copy = [0, 10, 30, 60, 100, 150, 200, 200][k]
        b[:n_copy] = q[:n_copy]
        blocks.append(b)
        ids.append(k * 10)
    ci, cs = oracle.loop_candidates(q, 1000, blocks, ids, 30)
    # score = good/200 must exceed 0.1; top 5 by score; ties keep DB order
    assert ci.tolist() == [6, 7, 5, 4, 3]
    assert all(cs[i] >= cs[i + 1] for i in range(len(cs) - 1)) and cs[0] == 1.0
    # min_frames_between excludes recent keyframes (LoopClosure.cpp:81)
    ci2, _ = oracle.loop_candidates(q, 75, blocks, ids, 30)
    assert 6 not in ci2.tolist() and 7 not in ci2.tolist() and 5 not in ci2.tolist()
