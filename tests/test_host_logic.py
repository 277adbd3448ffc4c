"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol the header
declares, the host-computed plan geometry / resize tables agree with the oracle, the synthetic generator is
pinned, and -- with no GPU in this container -- every compute entry point fails loudly (no CPU fallback)."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "aria_orb_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(aria_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(aria):
    from aria_slam_amd import _lib
    L = aria.load_library()
    declared = _declared_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), "libaria_orb_hip.so does not export %s" % name
    assert sorted(_lib.EXPORTS) == declared, "python binding list and header disagree"
    assert aria.abi_version() == 4


def test_status_strings(aria):
    for s in range(0, -9, -1):
        assert aria.status_string(s) != "unknown status"
    assert aria.status_string(-99) == "unknown status"


def test_record_layouts_match_reference_types(aria):
    # aria::core::KeyPoint = 5 floats + int (Types.hpp:9-15); aria::core::Match = 2 ints + float (Types.hpp:97-101)
    assert aria.KP_DTYPE.itemsize == 24 and aria.MATCH_DTYPE.itemsize == 12
    assert list(aria.KP_DTYPE.names) == ["x", "y", "size", "angle", "response", "octave"]
    assert list(aria.MATCH_DTYPE.names) == ["query_idx", "train_idx", "distance"]


@pytest.mark.parametrize("w,h", [(640, 480), (752, 480), (1408, 1408), (333, 251), (2047, 1024), (100, 100)])
@pytest.mark.parametrize("nf", [500, 1000, 2000, 4000, 0, 7])
def test_plan_geometry_matches_oracle(aria, oracle, w, h, nf):
    p = oracle.default_params(nf)
    sizes, quotas, scales = oracle.level_sizes(p, w, h), oracle.feature_quotas(p), oracle.layer_scales(p)
    info = aria.level_info(nf, w, h)
    assert [(a, b) for a, b, _, _ in info] == sizes
    assert [q for _, _, q, _ in info] == quotas
    assert [np.float32(s) for _, _, _, s in info] == [np.float32(s) for s in scales]


@pytest.mark.parametrize("w,h", [(640, 480), (752, 480), (1408, 1408), (333, 251)])
def test_resize_tables_match_oracle(aria, oracle, w, h):
    sizes = oracle.level_sizes(oracle.default_params(), w, h)
    for l in range(1, 8):
        for axis in (0, 1):
            ofs, c1 = aria.resize_table(w, h, l, axis)
            oo, oc = oracle.resize_coeffs(sizes[l - 1][axis], sizes[l][axis])
            assert np.array_equal(ofs, oo) and np.array_equal(c1, oc)


@pytest.mark.parametrize("w,h", [(640, 480), (752, 480), (1408, 1408), (333, 251), (2047, 1024), (100, 100), (64, 2047)])
def test_fused_pyramid_row_schedule(aria, oracle, w, h):
    """Every level row is owned by exactly one band, and every row a band reads was computed by that band."""
    from aria_slam_amd import _lib
    bands, bh, lds = _lib.pyramid_bands(w, h)
    sizes = oracle.level_sizes(oracle.default_params(), w, h)
    assert bh >= 8
    if w <= 1408:
        assert lds <= 150 * 1024      # wider images fall back to one resize launch per level
    nb = len(bands)
    for l in range(8):
        lw, lh = sizes[l]
        owned = np.zeros(lh, int)
        for b in range(nb):
            clo, cn, olo, on = bands[b, l]
            assert clo == olo and cn >= on >= 0 and 0 <= clo and clo + cn <= lh
            owned[olo:olo + on] += 1
            if l >= 1 and cn > 0:
                oy, _ = aria.resize_table(w, h, l, 1)
                src_lo, src_n = bands[b, l - 1, 0], bands[b, l - 1, 1]
                need_lo = oy[clo:clo + cn].min()
                need_hi = min(oy[clo:clo + cn].max() + 1, sizes[l - 1][1] - 1)
                assert src_lo <= need_lo and need_hi < src_lo + src_n, (l, b)
        assert np.all(owned == 1), "level %d rows not owned exactly once" % l


def test_algorithmic_bytes_match_baseline_md(aria):
    # BASELINE.md section 3 table
    assert aria.algorithmic_bytes(640, 480, 2000) == (4533474, 2013064)
    assert aria.algorithmic_bytes(752, 480, 1000) == (5253735, 2290734)
    assert aria.algorithmic_bytes(1408, 1408, 4000) == (28775747, 12499464)


def test_invalid_geometry_is_rejected(aria):
    with pytest.raises(aria.AriaError):
        aria.level_info(1000, 4000, 480)      # wider than the 11-bit candidate packing allows
    with pytest.raises(aria.AriaError):
        aria.level_info(1000, 8, 8)


def test_synthetic_generator_is_pinned(aria):
    a, b = aria.synth_frame_pair(1, 640, 480)
    # seed + sha256, not the image
    assert hashlib.sha256(a.tobytes()).hexdigest() == GOLD["synth_seed1_a"]
    assert hashlib.sha256(b.tobytes()).hexdigest() == GOLD["synth_seed1_b"]
    seq = aria.synth_sequence(1, 3, 640, 480, n_threads=3)
    assert np.array_equal(seq[0], a) and np.array_equal(seq[1], b)
    a3, b3 = aria.synth_frame_pair(3, 640, 480)
    assert np.array_equal(seq[4], a3) and np.array_equal(seq[5], b3)
    # B shows A's scene moved by (+3, +2): away from the noise the rectangles line up
    d = np.abs(b[2:, 3:].astype(int) - a[:-2, :-3].astype(int))
    assert np.percentile(d, 95) <= 12


def test_no_gpu_means_loud_failure_not_fallback(aria):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(aria.AriaError) as e:
        aria.OrbHipExtractor()
    assert e.value.status == -2
    with pytest.raises(aria.AriaError) as e:
        aria.HipMatcher()
    assert e.value.status == -2


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under aria_slam_amd/ may import, link or open it."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "aria_slam_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile", ".inc")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"orb_oracle|oracle_py|liborb_oracle|from oracle|import oracle", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


GOLD = {}
_g = os.path.join(ROOT, "tests", "golden", "golden.json")
if os.path.exists(_g):
    import json
    GOLD = json.load(open(_g))


def test_product_library_reads_no_environment():
    """evidence kernels and tuning switches live in the variants build (-DARIA_VARIANTS); the shipped
    library neither imports getenv nor contains an ARIA_* variable name, and its sources reach the environment only through
    common.h's aria_getenv (a constant nullptr without ARIA_VARIANTS)."""
    import subprocess
    so = os.path.join(ROOT, "aria_slam_amd", "libaria_orb_hip.so")
    if not os.path.exists(so):
        pytest.skip("library not built")
    und = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
    assert "getenv" not in und
    names = subprocess.run(["strings", so], capture_output=True, text=True).stdout.split("\n")
    assert not [n for n in names if re.fullmatch(r"ARIA_[A-Z0-9_]+", n.strip())]
    csrc = os.path.join(ROOT, "aria_slam_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp", ".h")):
            txt = open(os.path.join(csrc, f)).read()
            raw = [m.start() for m in re.finditer(r"(?<![_\w])getenv\(", txt)]
            if f in ("common.h", "orb_plan.cpp"):
                assert len(raw) <= 1, f        # the one call inside aria_getenv's ARIA_VARIANTS branch
            else:
                assert not raw, f
    mk = open(os.path.join(csrc, "Makefile")).read()
    prod = re.search(r"^SRC := (.*)$", mk, re.M).group(1).split()
    assert "band_mfma.hip" not in prod and "fast_blur_tile.hip" not in prod and "fast_blur_stream.hip" in prod
