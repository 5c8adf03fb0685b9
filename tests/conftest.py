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
PRODUCT_LIB = os.path.join(PKG, "libbgs.so")
TEST_LIB = os.path.join(PKG, "libbgs_test.so")

# The oracle's OpenMP regions are tiny in most tests: on a 256-core GPU box a thread team of 256 costs ~0.1 s per call
# (hundreds of calls per lock-step test).  A small team keeps the suite in minutes; an explicit setting wins.
os.environ.setdefault("OMP_NUM_THREADS", "8")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once, exactly as the driver does.
    # The product itself never builds or falls back on its own -- a missing libbgs.so is an ImportError there.
    if not (os.path.exists(PRODUCT_LIB) and os.path.exists(TEST_LIB) and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))):
        import __graft_entry__

        __graft_entry__.build()
    # The suite runs on the TEST build of the library (libbgs_test.so: the same kernel objects, the host units compiled with
    # -DBGS_TEST_HOOKS): the product library has no BGS_EXPERIMENT parser, so the tests that force a kernel family or inject a
    # fault (tests/knobs.py) could not reach it.  Children inherit the setting; the tests that measure or exercise the PRODUCT
    # (bench.py children, the C host program, the symbol-table and strings checks) take tests.knobs.product_env() / PRODUCT_LIB.
    os.environ.setdefault("BGS_LIBRARY", TEST_LIB)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
