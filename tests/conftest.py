import os
import sys

# before anything loads an OpenMP runtime: the oracle's team must block, not spin, when the box throttles it (oracle.py)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_report_header(config):
    try:
        from oracle import oracle
        return [oracle.describe_cpu()]
    except Exception as e:  # noqa: BLE001 - a header line must never break the run
        return [f"oracle: not available ({e})"]


def pytest_terminal_summary(terminalreporter):
    # shown with -q too (the report header is not): the driver's log then tells how the oracle's team was sized
    terminalreporter.write_line(pytest_report_header(None)[0])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: wall-clock bounds on an MI355X - never part of -m gpu (tools/round_end.sh runs -m perf)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure; never imported by karios_amd)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def ops():
    """GPU operators; fails loudly (no skip) when the HIP library or the device is missing."""
    from karios_amd import ops as _ops
    _ops._lib.default_context()  # raises without a GPU: gpu-marked tests must not pass silently
    return _ops


@pytest.fixture(scope="session")
def pair512():
    from karios_amd import synth
    return synth.make_pair(512, 512, 0.5, 0.0)


def rand_u8(shape, seed=0, lo=0, hi=256):
    return np.random.default_rng(seed).integers(lo, hi, shape, dtype=np.uint8)
