import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_share():
    """CPUs the test process can really use (affinity, cgroup quota): the CPU oracle is OpenMP code,
    and 256 threads on a 16-CPU quota only take turns."""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, math.ceil(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


os.environ.setdefault("OMP_NUM_THREADS", str(_cpu_share()))     # before liborc.so / libgomp are loaded


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_probe.json")) as f:
        return json.load(f)
