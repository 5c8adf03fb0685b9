"""pytest configuration: registers the `gpu` marker and puts the repo root (oracle/, bench.py) and the product
package directory (board-game-simulator-python_amd/, which provides the drop-in `simulator` package) on sys.path."""

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "board-game-simulator-python_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The oracle's OpenMP regions are tiny in most tests: on a 256-core GPU box a thread team of 256 costs ~0.1 s per call
# (hundreds of calls per lock-step test).  A small team keeps the suite in minutes; an explicit setting wins.
os.environ.setdefault("OMP_NUM_THREADS", "8")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once, exactly as the driver does.
    # The product itself never builds or falls back on its own -- a missing libbgs.so is an ImportError there.
    if not (os.path.exists(os.path.join(PKG, "libbgs.so")) and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))):
        import __graft_entry__

        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
