import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _granted_cpus():
    """CPUs this process may really use: the cgroup quota if there is one, else the affinity mask.  A 1-GPU box shows 256 logical
    CPUs and grants 16: torch then starts 128 intra-op threads and the CPU oracle - which most of the GPU suite's wall time
    is - runs 1.6 (float64) to 4.3 (fp32 convolution) times SLOWER than with 16 threads (measured, round 6)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")
    try:                                          # the oracle's threads = the CPUs granted, not the CPUs visible
        import torch
        torch.set_num_threads(max(1, min(torch.get_num_threads(), _granted_cpus())))
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
