"""Golden fixtures (tests/golden, made by tests/golden/make_golden.py from this repo's oracle -- they pin the
oracle and the generator against drift; parity against real OpenCV stays unpinned)."""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "golden.json")))
CASES = [k for k in GOLD if isinstance(GOLD[k], dict)]


def sha(x):
    return hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()


def _parse(key):
    s, wh, n = key.split("_")
    w, h = wh.split("x")
    return int(s[1:]), int(w), int(h), int(n[1:])


@pytest.mark.parametrize("key", [k for k in CASES if "1408" not in k])
def test_oracle_reproduces_golden(aria, oracle, key):
    seed, w, h, nf = _parse(key)
    g = GOLD[key]
    a, b = aria.synth_frame_pair(seed, w, h)
    assert sha(a) == g["image_a"] and sha(b) == g["image_b"]
    p = oracle.default_params(nf)
    ka, da = oracle.orb_extract(a, p)
    kb, db = oracle.orb_extract(b, p)
    assert (len(ka), len(kb)) == (g["n_a"], g["n_b"])
    assert sha(ka) == g["kp_a"] and sha(da) == g["desc_a"] and sha(kb) == g["kp_b"] and sha(db) == g["desc_b"]
    m = oracle.match_ratio(db, da, 0.75)
    assert len(m) == g["n_matches"] and sha(m) == g["matches"]
    assert np.bincount(ka["octave"], minlength=8).tolist() == g["per_level_a"]
    assert da[0].tolist() == g["first_desc_a"]


@pytest.mark.gpu
@pytest.mark.parametrize("key", CASES)
def test_hip_reproduces_golden_without_oracle(aria, key):
    seed, w, h, nf = _parse(key)
    g = GOLD[key]
    a, b = aria.synth_frame_pair(seed, w, h)
    e = aria.OrbHipExtractor(max_features=nf, max_width=w, max_height=h)
    m = aria.HipMatcher(max_query=nf + 1024, max_train=nf + 1024)
    try:
        fa, fb = e.extract(a), e.extract(b)
        assert (len(fa["keypoints"]), len(fb["keypoints"])) == (g["n_a"], g["n_b"])
        assert sha(fa["keypoints"]) == g["kp_a"] and sha(fa["descriptors"]) == g["desc_a"]
        assert sha(fb["keypoints"]) == g["kp_b"] and sha(fb["descriptors"]) == g["desc_b"]
        got = m.match(fb, fa, None, 0.75)
        assert len(got) == g["n_matches"] and sha(got) == g["matches"]
    finally:
        e.close()
        m.close()


@pytest.mark.gpu
def test_hip_matches_committed_vectors(aria):
    z = np.load(os.path.join(HERE, "golden", "golden_seed1_640x480.npz"))
    a, b = aria.synth_frame_pair(1, 640, 480)
    e = aria.OrbHipExtractor(max_features=2000, max_width=640, max_height=480)
    try:
        fa = e.extract(a)
        assert fa["keypoints"].tobytes() == z["kp_a"].tobytes()
        assert np.array_equal(fa["descriptors"], z["desc_a"])
    finally:
        e.close()
