import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cpus():
    """CPUs this process may really use: the affinity mask cut by the cgroup's CPU quota (the GPU box gives one GPU's job 16 of its
    256 logical CPUs: torch's default of one thread per logical CPU then runs the CPU oracles several times SLOWER)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p_ = f.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(float(q) / float(p_))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch
        torch.set_num_threads(_usable_cpus())
    except ImportError:
        pass


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
