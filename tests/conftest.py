import os
import sys

import pytest  # noqa: F401

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # the product's process default (musev_amd/__init__.py), set before any test touches the GPU

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def pytest_configure(config):
    # GPU tests fail loudly (they do not skip) when selected on a box without a GPU; `-m "not gpu"` deselects them here
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
