"""pytest configuration: registers the `gpu` marker and provides shared fixtures.

`-m "not gpu"` tests run on a CPU-only box: the oracle against its golden vectors and against the real
reference pieces in oracle/_ref, the host logic, and that the C-ABI library loads and exports every
symbol of include/ssd_hip.h.  `-m gpu` tests are the parity tests proper (HIP path vs oracle) and call
through the C ABI.
"""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ssd():
    mod = importlib.import_module("stair-step-detector_amd")
    if not os.path.exists(mod.LIB_PATH):
        # a fresh checkout (built artefacts are git-ignored): compile the HIP library and the oracle first.
        # This is a build step, not a fallback: without the library every test below fails loudly.
        import __graft_entry__
        __graft_entry__.build()
    return mod


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding.load_oracle()


@pytest.fixture(scope="session")
def ref():
    import oracle_binding
    r = oracle_binding.load_ref()
    if r is None:
        pytest.skip("oracle/_ref/libssd_ref.so not built (needs /root/reference at build time)")
    return r


@pytest.fixture(scope="session")
def gpu_device(ssd):
    if ssd.device_count() < 1:
        pytest.fail("a -m gpu test ran without a HIP device: the HIP path is mandatory, there is no fallback")
    return 0
