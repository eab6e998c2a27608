"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _granted_cores():
    """Cores the container's CPU quota grants (cgroup v2 cpu.max / v1 cfs quota), capped by the affinity mask."""
    import math

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(math.ceil(int(q) / int(per)))))
    except (OSError, ValueError):
        try:
            q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(math.ceil(q / per))))
        except (OSError, ValueError):
            pass
    return n


# the C oracle's OpenMP pool: the GPU boxes show 256 logical CPUs under a 16-core quota, and 256 threads there are throttled to a crawl
os.environ.setdefault("OMP_NUM_THREADS", str(_granted_cores()))
os.environ.setdefault("VH_POISON_WORKSPACE", "1")  # BA workspaces start as NaN bit patterns: reads of never-written memory fail loudly


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_report_header(config):
    """Which binary the suite runs: the id compiled into velocity_amd/libvelocity_hip.so and the hash of the tree's sources."""
    try:
        from velocity_amd import _build, _lib

        info = _lib.build_info()
        return (f"velocity_amd build_id: {info['build_id']} (tree sources hash to {info['source_hash']}, torch ops "
                f"{_build.file_build_id(_build.TORCH_OUT, _build._TMARK)}" + (f", VH_LIB override {info['override']}" if info["override"] else "") + ")")
    except Exception as e:  # reported, never hidden: test_lib_cpu fails on the same condition
        return f"velocity_amd build_id: UNAVAILABLE ({e})"


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Repeat the id in the tail of the output (the driver's GPUTEST record keeps the tail)."""
    try:
        from velocity_amd import _lib

        terminalreporter.write_line(f"velocity_amd build_id: {_lib.build_info()['build_id']}")
    except Exception as e:
        terminalreporter.write_line(f"velocity_amd build_id: UNAVAILABLE ({e})")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "nls_golden.npz"))


@pytest.fixture(scope="session")
def K32(golden):
    return golden["K32"]
