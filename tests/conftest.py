import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; oracle/ceres_oracle.h)."""
    return entry.load_oracle()


class _Problems:
    """The package's problem generators plus the reference-held test vectors (tests/reference_fixtures.py) under one name."""
    def __getattr__(self, name):
        import reference_fixtures
        if hasattr(reference_fixtures, name) and not name.startswith("_"):
            return getattr(reference_fixtures, name)
        return getattr(pkg.problems, name)


@pytest.fixture(scope="session")
def problems():
    return _Problems()


@pytest.fixture(scope="session")
def hip():
    """ctypes binding of the C ABI; GPU tests fail (not skip) if no device is usable,
    as the reference's GPU tests do (internal/ceres/cuda_sparse_matrix_test.cc:53)."""
    hs = pkg.hip_solver
    hs.load_library()
    assert hs.device_count() >= 1, "no gfx950 device visible: GPU tests cannot run"
    return hs
