"""The C++ adapters compile against the reference's REAL port headers.

aria_slam_amd/host normally builds against aria_hip/compat.hpp, a layout-compatible restatement of the three ports and the
structs they use. A maintainer of the reference compiles the same sources with -DARIA_HIP_USE_REFERENCE_HEADERS and
-I<aria-slam>/include; this test does exactly that in the build container, where the reference tree is mounted read-only at
/root/reference. The reference's core/Types.hpp includes <Eigen/Dense>, which the container lacks:
tests/cpp/eigen_standin/Eigen/Dense is a test-only stand-in for the few Eigen names that header uses (clearly labelled, never
on the product's include path). Skipped where the reference tree is absent (the GPU box): nothing of it travels.
"""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/include"
HOST = os.path.join(ROOT, "aria_slam_amd", "host")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF_INC, "interfaces")),
                                reason="reference tree not present (only in the build container)")


def _flags():
    return ["-std=c++17", "-fPIC", "-Wall", "-Wextra", "-Werror", "-DARIA_HIP_USE_REFERENCE_HEADERS",
            "-I" + REF_INC, "-I" + os.path.join(ROOT, "tests", "cpp", "eigen_standin"),
            "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HOST, "include")]


def test_every_adapter_source_compiles_against_the_reference_headers(tmp_path):
    cxx = shutil.which("g++")
    assert cxx
    srcs = sorted(glob.glob(os.path.join(HOST, "src", "*.cpp")))
    assert len(srcs) >= 7
    for src in srcs:
        obj = tmp_path / (os.path.basename(src) + ".o")
        r = subprocess.run([cxx, *_flags(), "-c", src, "-o", str(obj)], capture_output=True, text=True)
        assert r.returncode == 0, "%s does not compile against %s:\n%s" % (os.path.basename(src), REF_INC, r.stderr[-4000:])


def test_the_reference_header_build_links_into_the_adapter_library(tmp_path):
    """Same recipe as aria_slam_amd/host/Makefile with the reference headers: the library links against the C-ABI."""
    so = os.path.join(ROOT, "aria_slam_amd", "libaria_orb_hip.so")
    if not os.path.exists(so):
        pytest.skip("libaria_orb_hip.so not built")
    libsrc = [os.path.join(HOST, "src", n) for n in
              ("OrbHipExtractor.cpp", "HipMatcher.cpp", "HipLoopDetector.cpp", "HipFactory.cpp", "FrontEnd.cpp", "AslSequence.cpp")]
    out = tmp_path / "libaria_hip_adapters_ref.so"
    r = subprocess.run([shutil.which("g++"), *_flags(), "-shared", "-o", str(out), *libsrc,
                        "-L" + os.path.dirname(so), "-laria_orb_hip", "-lz", "-Wl,--no-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    # the ports are the reference's own: the vtables name aria::interfaces::IFeatureExtractor etc.
    syms = subprocess.run(["nm", "-DC", str(out)], capture_output=True, text=True).stdout
    for name in ("aria::adapters::hip::OrbHipExtractor::extract", "aria::adapters::hip::HipMatcher::match",
                 "aria::adapters::hip::HipLoopDetector::detect", "aria::pipeline::FrontEnd::processFrame",
                 "aria::adapters::hip::HipMatcher::matchDevice", "aria::adapters::hip::OrbHipExtractor::deviceResult"):
        assert name in syms, name


def test_the_stand_in_is_labelled_and_not_on_the_product_include_path():
    text = open(os.path.join(ROOT, "tests", "cpp", "eigen_standin", "Eigen", "Dense")).read()
    assert "TEST-ONLY STAND-IN" in text and "NOT Eigen" in text
    mk = open(os.path.join(HOST, "Makefile")).read()
    assert "eigen_standin" not in mk
