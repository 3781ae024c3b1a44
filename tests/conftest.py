import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure libddcmi.so and the oracle exist (compiles them when missing)."""
    import ddcmd_amd._lib as L
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    import pyoracle
    pyoracle.lib()
    return True


@pytest.fixture(scope="session")
def waterbox(built):
    """examples/waterbox inputs + oracle outputs (tests/golden/waterbox.npz)."""
    from ddcmd_amd.deck import setup_from_dict
    d = dict(np.load(os.path.join(GOLDEN, "waterbox.npz")))
    return setup_from_dict(d), d


def has_gpu():
    try:
        import ddcmd_amd._lib as L
        lib = L.load_library()
        lib.ddcmi_device_count.restype = int
        return lib.ddcmi_device_count() > 0
    except Exception:
        return False


def rel_force_err(f, g):
    """max |f-g| / max |g| over all components (BASELINE.md parity gate)."""
    num = max(np.abs(f[k] - g[k]).max() for k in range(3))
    den = max(np.abs(g[k]).max() for k in range(3))
    return num / den
