"""Alternative kernel paths (selected by environment variables, read once per process) must give the same bits:
the 64x32 tile FAST/blur kernel, direct-gather resize, the fused in-LDS pyramid, per-level side streams, the vector-ALU kNN-2, and --
by shrinking the survivor queue to 1 % -- the dense-rescoring slow path of the band kernel."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = [
    {},
    {"ARIA_FAST_BLUR_IMPL": "tile"},
    {"ARIA_FAST_BLUR_IMPL": "mfma"},                               # band kernel with the 7x7 blur on the matrix cores (band_mfma.hip)
    {"ARIA_RESIZE_FUSE": "0"},                                     # separate resize pass (dot2 LDS kernel)
    {"ARIA_RESIZE_IMPL": "direct"},
    {"ARIA_RESIZE_IMPL": "lds"},                                   # LDS-staged resize with shift/mad arithmetic
    {"ARIA_PYRAMID_IMPL": "fused"},
    {"ARIA_LEVEL_STREAMS": "1"},
    {"ARIA_BAND_QPCT0": "1", "ARIA_BAND_QPCT_STEP": "0"},          # survivor queue overflows -> slow path
    {"ARIA_BAND_BUDGET_KB": "160"},                                # several strips per workgroup
    {"ARIA_KNN_IMPL": "valu"},                                     # vector-ALU kNN-2 instead of the matrix-core one
    {"ARIA_SELECT_SORT": "bitonic"},
    {"ARIA_ZERO_COPY": "1"},                                       # single-frame: pyramid kernel reads the pinned host frame itself
    {"ARIA_BAND_XCD_MAP": "0"},                                    # plain (strip, frame) order of the batch FAST/blur workgroups                               # k_select's LDS bitonic sort (fallback of the bin sort)
]


@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()) or "default")
def test_variant_matches_golden(env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_check.py")], env=e, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("OK") == 3


@pytest.mark.parametrize("env", [{}, {"ARIA_KNN_NC": "2"}, {"ARIA_KNN_NC": "4"}, {"ARIA_KNN_IMPL": "valu"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()) or "default")
def test_knn2_kernels_against_brute_force(env):
    """Matrix-core kNN-2 in both workgroup shapes (256 / 512 queries), both key layouts (train sets above 4096
    descriptors take the 16-bit-index one) and the vector-ALU kernel, through the host, batched-device and
    keyframe-DB entry points, against a numpy brute force."""
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "knn_check.py")], env=e, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bad" not in out.stdout
