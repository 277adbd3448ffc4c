"""The C++ adapters (aria_slam_amd/host) implementing the reference's IFeatureExtractor / IMatcher ports."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "aria_slam_amd")
EXE = os.path.join(ROOT, "tests", "cpp", "adapter_selftest")


def _fnv(b):
    h = 1469598103934665603
    for x in bytes(b):
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


@pytest.fixture(scope="module")
def selftest(aria):
    subprocess.check_call(["make", "-C", os.path.join(PKG, "host"), "-s"])
    src = os.path.join(ROOT, "tests", "cpp", "adapter_selftest.cpp")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(PKG, "libaria_hip_adapters.so"))):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(PKG, "host", "include"), src, "-o", EXE, "-L" + PKG,
                               "-laria_hip_adapters", "-laria_orb_hip", "-lz", "-Wl,-rpath," + PKG])
    return EXE


def test_adapters_fail_loudly_without_gpu(selftest):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = subprocess.run([selftest, "nogpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK nogpu" in out.stdout and "no usable HIP device" in out.stdout


def test_shard_partition_equals_the_python_rule(selftest):
    """aria_hip/Shard.hpp (euroc_frontend --shards) and aria_slam_amd/shard.py (bench.py) cut a sequence the same way:
    contiguous ranges that cover it once, a one-frame halo in front of every non-empty range but the first."""
    from aria_slam_amd import shard
    out = subprocess.run([selftest, "shard"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0
    rows = [tuple(int(v) for v in l.split()) for l in out.stdout.splitlines()]
    assert len(rows) == 8 * 36
    for n, g, r, lo, hi, first in rows:
        p = shard.shard_plan(n, r, g)
        assert (lo, hi) == (p["lo"], p["hi"]) and first == p["extract"][0], (n, g, r)
    for n in (7, 100, 4096):
        for g in range(1, 9):
            cover = [i for nn, gg, r, lo, hi, first in rows if nn == n and gg == g for i in range(lo, hi)]
            assert cover == list(range(n))


@pytest.mark.gpu
def test_adapters_equal_python_binding_and_reference_conventions(aria, selftest):
    out = subprocess.run([selftest, "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "DONE" in out.stdout, out.stdout + out.stderr
    kv = {}
    for line in out.stdout.splitlines():
        t = line.split()
        kv[t[0]] = t[1:]
    a, b = aria.synth_frame_pair(1, 640, 480)
    e = aria.OrbHipExtractor(max_features=2000)
    m = aria.HipMatcher()
    fa, fb = e.extract(a), e.extract(b)
    assert kv["n_a"] == [str(len(fa["keypoints"])), "n_b", str(len(fb["keypoints"]))]
    assert kv["kp_a"] == [_fnv(fa["keypoints"].tobytes()), "desc_a", _fnv(fa["descriptors"].tobytes())]
    assert kv["kp_b"] == [_fnv(fb["keypoints"].tobytes()), "desc_b", _fnv(fb["descriptors"].tobytes())]
    assert kv["async_same"] == ["1"]
    mm = m.match(fb, fa)
    assert kv["n_matches"] == [str(len(mm)), "first_kept", "1"]            # append semantics (CudaMatcher.cpp:65)
    assert kv["matches"] == [_fnv(mm.tobytes())]
    assert kv["empty_untouched"] == ["1"]                                  # CudaMatcher.cpp:35-37
    assert kv["multi"] == ["3", str(len(mm)), str(len(m.match(fb, fb))), "0"]   # IMatcher.hpp:33 resize + loop
    assert kv["multi_hash"] == [_fnv(mm.tobytes()), _fnv(m.match(fb, fb).tobytes())]   # one batched launch == per-candidate match()
    e.setMaxFeatures(500)
    f5 = e.extract(a)
    assert kv["n_500"] == [str(len(f5["keypoints"])), "max", "500"]
    assert kv["kp_500"] == [_fnv(f5["keypoints"].tobytes()), "desc_500", _fnv(f5["descriptors"].tobytes())]
    assert kv["loop"][0] == "1" and float(kv["loop"][1]) > 0.1
    # HipLoopDetector (ILoopDetector over the HBM-resident database), capacity 3 after four insertions
    from oracle import oracle_py as O
    blocks = [fb["descriptors"], f5["descriptors"], fa["descriptors"]]
    ci, cs = O.loop_candidates(fb["descriptors"], 50, blocks, [1, 2, 3], 1)
    assert kv["ld_size"] == ["3", "oldest", "1", "ncand", str(len(ci))]
    got = [(int(t[1]), float(t[2])) for t in (l.split() for l in out.stdout.splitlines()) if t[0] == "ld_cand"]
    assert [g[0] for g in got] == [[1, 2, 3][i] for i in ci.tolist()] and [g[1] for g in got] == cs.tolist()
    assert kv["ld_loop"][0] == "1" and kv["ld_loop"][1] == "1" and int(kv["ld_loop"][2]) == len(m.match(fb, fb, None, 0.7)) and kv["ld_loop"][3] == "1"
    assert kv["ld_recent"] == ["0"]
    assert kv["factory"] == [str(len(fa["keypoints"])), str(len(fb["keypoints"])), str(len(mm)), "1", "1"]
    # device hand-off (getGpuDescriptors / matchGpu): equal to the host port calls, which the parity tests pin to the oracle
    from oracle import oracle_py as O2
    e.setMaxFeatures(2000)
    fa2, fb2 = e.extract(a), e.extract(b)
    want_ba = O2.match_ratio(fb2["descriptors"], fa2["descriptors"], 0.75)
    want_ab = O2.match_ratio(fa2["descriptors"], fb2["descriptors"], 0.75)
    want_aa = O2.match_ratio(fa2["descriptors"], fa2["descriptors"], 0.75)
    assert kv["dev_n"][:2] == [str(len(fa2["keypoints"])), str(len(fb2["keypoints"]))] and kv["dev_n"][5] == str(len(fb2["keypoints"]))
    # resident b as the query against a: the legacy order of the next frame (euroc_eval.cpp:168-169)
    want_b_a = O2.match_ratio(fb2["descriptors"], fa2["descriptors"], 0.75)
    assert kv["dev_match"] == [_fnv(want_ba.tobytes()), _fnv(want_b_a.tobytes()), _fnv(want_aa.tobytes())]
    assert kv["dev_ptr_stable"] == ["1"]
    for legacy in (0, 1):
        rows = [kv["fe_%d_%d" % (legacy, mode)] for mode in range(3)]
        assert [r[1] for r in rows] == ["0", "1", "2"]                     # host calls, device hand-off, queued behind extractAsync
        assert rows[0][2:] == rows[1][2:] == rows[2][2:], rows
        pairs = [(fb2, fa2), (fa2, fb2), (fb2, fa2)]                       # (current, previous) of frames 1..3
        exp = ["0:%s" % _fnv(b"")]
        for cur, prev in pairs:
            w = O2.match_ratio(prev["descriptors"], cur["descriptors"], 0.75) if legacy else O2.match_ratio(cur["descriptors"], prev["descriptors"], 0.75)
            exp.append("%d:%s" % (len(w), _fnv(w.tobytes())))
        assert rows[0][2:] == exp
    # setMaxFeatures(6000) after the first frame (above the matcher handle's 4096 rows): no throw, same matches in every mode,
    # and they are the oracle's for (2000-feature a) -> (6000-feature b, a, b)
    grow = [kv["fe_grow_%d" % mode] for mode in range(3)]
    assert grow[0] == grow[1] == grow[2], grow
    e.setMaxFeatures(6000)
    fa6, fb6 = e.extract(a), e.extract(b)
    assert len(fb6["keypoints"]) > 4096
    exp = ["%d:0:%s" % (len(fa2["keypoints"]), _fnv(b""))]
    for cur, prev in ((fb6, fa2), (fa6, fb6), (fb6, fa6)):
        w = O2.match_ratio(cur["descriptors"], prev["descriptors"], 0.75)
        exp.append("%d:%d:%s" % (len(cur["keypoints"]), len(w), _fnv(w.tobytes())))
    assert grow[0] == exp
    e.close()
    m.close()
