import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle_lib():
    """oracle/liboracle_hsddp.so — the CPU restatement (checker). Built on demand."""
    path = os.path.join(ROOT, "oracle", "liboracle_hsddp.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return pkg._abi.bind(ctypes.CDLL(path))


@pytest.fixture(scope="session")
def oracle_ld_lib():
    """oracle/liboracle_hsddp_ld.so — the same restatement with 80-bit long double internals: arbiter of the conditioning-limited cases."""
    path = os.path.join(ROOT, "oracle", "liboracle_hsddp_ld.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return pkg._abi.bind(ctypes.CDLL(path))


@pytest.fixture(scope="session")
def hip_lib():
    return pkg.load_hip_library()


@pytest.fixture(scope="session")
def golden():
    d = os.path.join(ROOT, "tests", "golden")
    return {f[:-4]: np.load(os.path.join(d, f)) for f in os.listdir(d) if f.endswith(".npz")}
