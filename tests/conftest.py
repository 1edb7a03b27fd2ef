import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/liboracle.so) -- the checker.  Built on demand with gcc."""
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.lib()
    pyoracle.set_threads(pyoracle.usable_cores())   # OpenMP's default (every logical CPU) oversubscribes a cgroup-limited box
    return pyoracle


@pytest.fixture(scope="session")
def ctx():
    """One rcv_ctx on GPU 0.  No skip-if-missing: -m gpu tests must fail loudly without the HIP path."""
    import rustcv_amd
    c = rustcv_amd.Context(0)
    yield c
    c.close()


@pytest.fixture
def rng():
    return np.random.default_rng(0xC0FFEE)


@pytest.fixture(autouse=True)
def _fresh_knobs(request):
    """librustcv_hip.so reads its RCV_* environment knobs once per process; start every GPU test from the current environment
    (the previous test's monkeypatch has been undone by now)."""
    if request.node.get_closest_marker("gpu") is not None:
        from rustcv_amd import _ffi
        _ffi.lib().rcv__debug_reload_knobs()
    yield


@pytest.fixture
def knob(monkeypatch):
    """knob("RCV_F7_NO_LAT") sets an environment knob of the library for this test and makes the library re-read them."""
    from rustcv_amd import _ffi

    def _set(name, value="1"):
        monkeypatch.setenv(name, str(value))
        _ffi.lib().rcv__debug_reload_knobs()
    return _set
