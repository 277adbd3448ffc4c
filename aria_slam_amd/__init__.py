"""MI355X-native ORB extractor + brute-force Hamming matcher for aria-slam's feature front-end.

The product is the C-ABI shared library (include/aria_orb_hip.h, built from aria_slam_amd/csrc for gfx950) and
the C++ adapters under aria_slam_amd/host that implement the reference's IFeatureExtractor / IMatcher ports.
This Python package is a thin ctypes binding of the same C-ABI used by tests/ and bench.py; it contains no
compute of its own and no CPU fallback: if the HIP library is missing or no GPU is present, it raises.
"""
from ._lib import (KP_DTYPE, MATCH_DTYPE, AriaError, abi_version, algorithmic_bytes, build_library, build_variants_library, level_info,
                   library_path, load_library, resize_table, status_string, synth_frame_pair, synth_sequence)
from .frontend import HipMatcher, OrbHipExtractor, flag_keypoints_device

__all__ = ["KP_DTYPE", "MATCH_DTYPE", "AriaError", "abi_version", "build_library", "build_variants_library", "library_path", "load_library",
           "status_string", "level_info", "resize_table", "algorithmic_bytes", "synth_frame_pair", "synth_sequence", "HipMatcher", "OrbHipExtractor", "flag_keypoints_device"]
