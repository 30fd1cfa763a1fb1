import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def usable_cores() -> int:
    """min(affinity, cgroup cpu quota).  torch defaults to one thread per visible core; under a cgroup quota (the GPU boxes
    grant 16 of several hundred cores) that oversubscribes the oracle's f32 matmuls by an order of magnitude."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test of several minutes; runs only with ESMDIFF_RUN_SLOW=1 (its last result is "
                                       "committed under profiles/)")
    import torch
    torch.set_num_threads(usable_cores())


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
