import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        from ddp_pinocchio_amd import capi
        return capi.lib().ddp_hip_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run the HIP path; without a device they fail loudly instead of falling back."""
    from ddp_pinocchio_amd import capi
    n = capi.lib().ddp_hip_device_count()
    assert n > 0, "no HIP device visible: -m gpu tests need an MI355X (there is no CPU fallback)"
    return capi
