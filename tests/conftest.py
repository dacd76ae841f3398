import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_pending: GPU tests of code written after the round's GPU budget was spent and not "
                                       "yet run on a GPU; selected only by an explicit `-m gpu_pending`")


def pytest_collection_modifyitems(config, items):
    # GPU tests must fail loudly (not skip) when selected on a box without a GPU; they are simply deselected by
    # `-m "not gpu"` on the CPU container.  `gpu_pending` tests need a GPU too but are NOT part of `-m gpu`: they run only
    # when asked for by name (`-m gpu_pending`), so `-m "not gpu"` on the CPU container must not pick them up either.
    markexpr = config.getoption("-m") or ""
    if "gpu_pending" in markexpr:
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("gpu_pending") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep
