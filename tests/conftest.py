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


def pytest_terminal_summary(terminalreporter):
    """Book-keeping of the conditioning arbiter (tests/parity_common.py): every tolerance it widened in this session."""
    import parity_common as pc
    if not pc.GRANTS:
        terminalreporter.write_line("[arbiter] no tolerance was widened in this session")
        return
    worst = {}
    for (tag, f, i, tol, granted, own, err, sc) in pc.GRANTS:
        k = (tag.split(" ")[0], f)
        if k not in worst or granted > worst[k][1]:
            worst[k] = (tol, granted, own, err, sc)
    terminalreporter.write_line(f"[arbiter] {len(pc.GRANTS)} widened tolerances; largest grant per (comparison, field):")
    for (tag, f), (tol, granted, own, err, sc) in sorted(worst.items()):
        terminalreporter.write_line(f"[arbiter]   {tag:24s} {f:12s} plain {tol:.2e} -> granted {granted:.2e}  (oracle-vs-exact {own:.2e}, measured error {err if err is None else format(err, '.2e')}, scale {sc:.2e})")
