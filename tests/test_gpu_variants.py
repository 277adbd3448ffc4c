"""Alternative kernel paths must give the same bits: the band FAST/blur kernel in the batch path (the default there is the
streaming kernel), the 64x32 tile FAST/blur kernel, the matrix-core blur, direct-gather resize, the fused in-LDS pyramid,
per-level side streams, the vector-ALU kNN-2, and -- by shrinking the survivor queue to 1 % -- the dense-rescoring slow
path of the band kernel.

These kernels and the ARIA_* switches that select them are NOT in the product library (libaria_orb_hip.so reads no
environment variable): this module builds libaria_orb_hip_variants.so (`make -C aria_slam_amd/csrc variants`,
-DARIA_VARIANTS) itself and points the subprocesses' binding at it through ARIA_ORB_HIP_LIBRARY."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

@pytest.fixture(scope="module")
def variants_lib():
    sys.path.insert(0, ROOT)
    import aria_slam_amd
    return aria_slam_amd.build_variants_library()


VARIANTS = [
    {},
    {"ARIA_FAST_BLUR_IMPL": "band"},                               # batches on the band kernel instead of the streaming one
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
    {"ARIA_KNN_IMPL": "int8"},                                     # int8 matrix-core kNN-2 instead of the FP4 one
    {"ARIA_SELECT_SORT": "bitonic"},
    {"ARIA_ZERO_COPY": "1"},                                       # single-frame: pyramid kernel reads the pinned host frame itself
    {"ARIA_BAND_XCD_MAP": "0"},                                    # plain (strip, frame) order of the batch FAST/blur workgroups                               # k_select's LDS bitonic sort (fallback of the bin sort)
]


@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()) or "default")
def test_variant_matches_golden(env, variants_lib):
    e = dict(os.environ)
    e["ARIA_ORB_HIP_LIBRARY"] = variants_lib
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_check.py")], env=e, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("OK") == 6


@pytest.mark.parametrize("env", [{}, {"ARIA_KNN_NC": "2"}, {"ARIA_KNN_NC": "4"}, {"ARIA_KNN_IMPL": "valu"}, {"ARIA_KNN_IMPL": "int8"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()) or "default")
def test_knn2_kernels_against_brute_force(env, variants_lib):
    """Matrix-core kNN-2 in both workgroup shapes (256 / 512 queries), both key layouts (train sets above 4096
    descriptors take the 16-bit-index one) and the vector-ALU kernel, through the host, batched-device and
    keyframe-DB entry points, against a numpy brute force."""
    e = dict(os.environ)
    e["ARIA_ORB_HIP_LIBRARY"] = variants_lib
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "knn_check.py")], env=e, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bad" not in out.stdout


# Build options of the streaming kernel that the product does not take (aria_slam_amd/csrc/fast_blur_stream.hip): the
# role-split launch (pyramid waves beside FAST/blur waves), the unconstrained three-wave form
# (branching walk, wide LDS lists) and the pyramid step as a separate phase over the LDS ring. Each is a product-flags build with extra defines (tools/build_ab.sh), held to the golden
# digests through the batch entry point.
STREAM_BUILDS = {
    "split": ["-DARIA_STREAM_SPLIT=1", "-DARIA_SCORE_SEQ=1", "-DARIA_STREAM_PYR_INWALK=0"],
    "three_waves": ["-DARIA_STREAM_WAVES4=0", "-DARIA_STREAM_FLAT=0", "-DARIA_STREAM_COMPACT_LDS=0", "-DARIA_STREAM_PEND_LDS=0",
                    "-DARIA_STREAM_PYR_INWALK=0"],
    "pyramid_phase": ["-DARIA_STREAM_PYR_INWALK=0"],          # the pyramid step as a phase of its own over the LDS ring (until round 4)
}


@pytest.mark.parametrize("name", sorted(STREAM_BUILDS))
def test_stream_kernel_build_options_match_golden(name):
    lib = os.path.join(ROOT, "build", "ab", "libopt_%s.so" % name)
    csrc = os.path.join(ROOT, "aria_slam_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".cpp")))
    if not os.path.exists(lib) or os.path.getmtime(lib) < newest:       # (a library built in the CPU container travels with the snapshot)
        subprocess.check_call([os.path.join(ROOT, "tools", "build_ab.sh"), "opt_" + name] + STREAM_BUILDS[name], stdout=subprocess.DEVNULL)
    e = dict(os.environ)
    e["ARIA_ORB_HIP_LIBRARY"] = lib
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_check.py")], env=e, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("batch (k_fast_blur_stream) OK") == 3, out.stdout
