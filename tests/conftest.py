import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The CPU side of the parity tests (the oracle, gloo ranks) on a many-core host: PyTorch defaults to one thread per physical core
# (128 on the GPU boxes' EPYC 9575F), which measures 5 x SLOWER than 16 threads on a 3072 x 4096 x 1024 product there (199 vs 39 ms:
# the box's usable cores are far fewer than it lists, bench.py's `cpu_baseline.by_threads` shows the same).  Bound it — for this
# process and, through the environment, for the rank / bench processes the tests start — unless the caller chose a count.
CPU_THREADS = 16
if (os.cpu_count() or 1) > CPU_THREADS and "OMP_NUM_THREADS" not in os.environ:
    os.environ["OMP_NUM_THREADS"] = str(CPU_THREADS)
    try:
        import torch as _torch
        _torch.set_num_threads(CPU_THREADS)
    except Exception:      # collection must not depend on it
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. `pytest tests/` here."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
