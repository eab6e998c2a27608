"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


os.environ.setdefault("VH_POISON_WORKSPACE", "1")  # BA workspaces start as NaN bit patterns: reads of never-written memory fail loudly


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "nls_golden.npz"))


@pytest.fixture(scope="session")
def K32(golden):
    return golden["K32"]
