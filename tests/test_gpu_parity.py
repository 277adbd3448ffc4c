"""GPU parity tests: HIP path (through the C-ABI) vs the CPU oracle, bit for bit.

The oracle is a restatement of OpenCV 4.9.0's CPU ORB / BFMatcher (parity unpinned: the reference holds no
golden vectors and OpenCV is not installed here), so these tests prove "HIP == restatement".
Tolerances: none. Every comparison below is exact (integer, byte, index and float-bit equality).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _kp_bits(k):
    return k.view(np.uint8).reshape(len(k), -1)


@pytest.fixture(scope="module")
def ext2000(aria):
    e = aria.OrbHipExtractor(max_features=2000, max_width=752, max_height=480)
    yield e
    e.close()


@pytest.mark.parametrize("seed,w,h", [(1, 640, 480), (2, 640, 480), (7, 752, 480), (3, 333, 251)])
def test_pyramid_and_blur_levels_bit_exact(aria, oracle, ext2000, seed, w, h):
    a, _ = aria.synth_frame_pair(seed, w, h)
    ext2000.extract(a)
    p = oracle.default_params(2000)
    raw = oracle.build_pyramid(a, p)
    blur = oracle.blur_pyramid(raw, p, w, h)
    info = ext2000.level_info(w, h)
    for l in range(8):
        lw, lh = info[l][0], info[l][1]
        assert (lh, lw) == raw[l].shape
        g_raw = ext2000.debug_read_level(l, False, lw, lh)
        g_blur = ext2000.debug_read_level(l, True, lw, lh)
        assert np.array_equal(g_raw, raw[l]), "raw level %d differs at %d px" % (l, np.count_nonzero(g_raw != raw[l]))
        assert np.array_equal(g_blur, blur[l]), "blurred level %d differs at %d px" % (l, np.count_nonzero(g_blur != blur[l]))


@pytest.mark.parametrize("seed,w,h,nf", [(1, 640, 480, 2000), (2, 640, 480, 2000), (5, 640, 480, 500),
                                          (7, 752, 480, 1000), (3, 333, 251, 300)])
def test_extract_bit_exact(aria, oracle, seed, w, h, nf):
    a, b = aria.synth_frame_pair(seed, w, h)
    e = aria.OrbHipExtractor(max_features=nf, max_width=w, max_height=h)
    try:
        for img in (a, b):
            f = e.extract(img)
            p = oracle.default_params(nf)
            kps, desc = oracle.orb_extract(img, p)
            assert len(f["keypoints"]) == len(kps)
            assert np.array_equal(f["keypoints"]["octave"], kps["octave"])
            assert np.array_equal(_kp_bits(f["keypoints"]), _kp_bits(kps)), "keypoint records differ"
            assert np.array_equal(f["descriptors"], desc), "descriptors differ"
    finally:
        e.close()


def test_match_indices_identical(aria, oracle):
    a, b = aria.synth_frame_pair(1, 640, 480)
    e = aria.OrbHipExtractor(max_features=2000, max_width=640, max_height=480)
    m = aria.HipMatcher()
    try:
        fa, fb = e.extract(a), e.extract(b)
        idx, dist = m.knn2(fb, fa)
        oidx, odist = oracle.knn2(fb["descriptors"], fa["descriptors"])
        assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
        for ratio in (0.75, 0.7, 0.0, 1.0):
            got = m.match(fb, fa, None, ratio)
            want = oracle.match_ratio(fb["descriptors"], fa["descriptors"], ratio)
            assert got.tobytes() == want.tobytes(), "ratio %g" % ratio
        assert 200 <= len(m.match(fb, fa)) <= 2000
    finally:
        e.close()
        m.close()
