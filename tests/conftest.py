import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running exhaustive check, not part of the default runs")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): parity unpinned restatement of OpenCV 4.9.0 CPU ORB + BFMatcher."""
    from oracle import oracle_py
    oracle_py.build()
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def aria():
    import aria_slam_amd
    if not os.path.exists(aria_slam_amd.library_path()):
        aria_slam_amd.build_library()
    aria_slam_amd.load_library()
    return aria_slam_amd
