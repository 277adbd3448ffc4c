"""Build-time lint over the device ISA of the extractor kernels (CPU: hipcc cross-compiles gfx950 without a GPU).

Round 3 shipped a work-around for wrong blurred pixels that appeared only while the matcher's MFMA waves shared the SIMDs:
the blur's column pass written on float2 values compiled to `v_pk_add_f32 ... op_sel:[1,0] op_sel_hi:[0,1]` (the LOW result
takes the HIGH half of a register pair). profiles/r4_hazard_isa_diff.md: the failing and the shipped build have the same 7200
instructions in the same order, the same waits and nops and the same 161 VGPRs -- they differ ONLY in register numbering and
in those crossed-half forms. Nothing but the compiler's register coalescing decides whether the crossed form appears, so this
test holds every extractor kernel to "no packed fp32 instruction with a crossed half"; the broadcast form (`op_sel_hi:[1,0]`,
both results from the low half -- k_describe's scalar x vector products) is not affected and is allowed."""
import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_kernel_stats as S   # noqa: E402

CSRC = os.path.join(ROOT, "aria_slam_amd", "csrc")
# the flags of aria_slam_amd/csrc/Makefile (product build), device side only, to assembly
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-fast-math", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "--cuda-device-only", "-S", "-w"]


def _listing(src_name):
    src = os.path.join(CSRC, src_name)
    h = hashlib.sha256()
    for f in [src] + [os.path.join(CSRC, x) for x in ("common.h", "orb_device.h", "orb_kernels.h", "orb_plan.h")]:
        h.update(open(f, "rb").read())
    out_dir = os.path.join(ROOT, "build", "isa")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "%s.%s.s" % (src_name, h.hexdigest()[:16]))
    if not os.path.exists(out):
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, src])
    return open(out).read()


@pytest.mark.parametrize("src_name,kernels", [
    ("fast_blur_stream.hip", ["k_fast_blur_streamILi1E", "k_fast_blur_streamILi2E", "k_fast_blur_streamILi4E"]),
    ("orb_kernels.hip", ["k_describeILi0ELb1E", "k_describeILi1ELb1E", "k_describeILi0ELb0E", "k_describeILi2ELb0E", "k_selectILb0E"]),
])
def test_no_crossed_half_packed_fp32(src_name, kernels):
    text = _listing(src_name)
    for k in kernels:
        body, meta = S.kernel_body(text, k)
        assert len(body) > 200, "kernel %s not found in the listing" % k
        crossed, _broadcast = S.swizzled_pk_f32(body)
        assert not crossed, "%s: packed fp32 with a crossed register half (the form of the round-3 hazard):\n%s" % (k, "\n".join(crossed[:8]))
        if "k_fast_blur_stream" in k:
            # held to 128 VGPRs (four waves per SIMD); what the allocator spills must stay OUTSIDE the loops: a reload is a
            # vector-memory load, and waiting for it waits for every prefetched row issued before it (vmcnt counts in order)
            assert meta.get("TotalNumVgprs", 999) <= 128 and meta.get("Occupancy") == 4, meta
            in_loop, outside = S.scratch_accesses(text, k)
            assert not in_loop, "%s: scratch access inside a loop:\n%s" % (k, "\n".join(in_loop[:8]))
            assert meta.get("ScratchSize", 0) <= 32 and len(outside) <= 8, (meta, outside)
        else:
            assert meta.get("ScratchSize", 0) == 0, "%s spills to scratch" % k


def test_lint_recognises_a_spill_inside_a_loop():
    text = "\n".join(["_Z1kv:", "; %bb.0:", "\tscratch_store_dword off, v5, off offset:16", ".LBB0_1:    ; =>This Loop Header: Depth=1",
                      "\tscratch_load_dword v5, off, off offset:16", "; %bb.2:      ;   in Loop: Header=BB0_1 Depth=1",
                      "\tscratch_load_dword v6, off, off", ".LBB0_3:", "\tscratch_load_dword v7, off, off", "\ts_endpgm",
                      "\t.section .rodata"])
    in_loop, outside = S.scratch_accesses(text, "_Z1kv")
    assert len(in_loop) == 2 and len(outside) == 2


def test_lint_recognises_the_hazard_form():
    body = ["v_pk_add_f32 v[46:47], v[14:15], v[32:33] op_sel:[1,0] op_sel_hi:[0,1]",      # crossed src0
            "v_pk_add_f32 v[46:47], v[40:41], v[14:15] op_sel:[0,1] op_sel_hi:[1,0]",      # crossed src1
            "v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0]",                         # broadcast of src1's low half
            "v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]",
            "v_pk_add_f32 v[2:3], v[4:5], v[6:7]",
            "v_pk_sub_i16 v1, v2, v2 op_sel:[0,1] op_sel_hi:[1,0]"]                          # 16-bit packed: not the hazard's class
    crossed, broadcast = S.swizzled_pk_f32(body)
    assert crossed == body[:2] and broadcast == body[2:4]


@pytest.mark.parametrize("src_name", ["fast_blur_stream.hip", "fast_blur_band.hip"])
def test_fast_score_kernels_keep_half_denormals(src_name):
    """fast9_score_f16 (orb_device.h, round 4) computes the FAST arcs on bytes read as HALF-FLOAT SUBNORMALS (0x00bb =
    b * 2^-24): every kernel that contains it must run with half denormals enabled (the HIP default,
    .amdhsa_float_denorm_mode_16_64 3) -- a build flag that flushed them would turn every difference into 0 silently on the
    CPU side of the build; the GPU parity tests would catch it, this catches it without a GPU."""
    text = _listing(src_name)
    assert "v_pk_minimum3_f16" in text and "v_pk_maximum3_f16" in text, "the half-float score is not in this build"
    modes = [ln.split()[-1] for ln in text.splitlines() if ".amdhsa_float_denorm_mode_16_64" in ln]
    assert modes and all(m == "3" for m in modes), modes
