"""The host-side C++ mirror of the reference interface (include/ddp/*.hpp): the reference's own indexing tests
re-expressed (CPU), and the reference's pendulum driver on the HIP path (GPU)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "host")


def _build():
    subprocess.check_call(["make", "-s", "-C", HOST])


def test_indexer_and_mat_seq_known_answers():
    _build()
    out = subprocess.run([os.path.join(HOST, "test_host")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_driver_refuses_without_a_device():
    from ddp_pinocchio_amd import capi
    if capi.lib().ddp_hip_device_count() > 0:
        pytest.skip("a GPU is visible")
    _build()
    out = subprocess.run([os.path.join(HOST, "pendulum_ddp")], capture_output=True, text=True)
    assert out.returncode == 77 and "no HIP device" in out.stdout    # loud failure, no CPU fallback


@pytest.mark.gpu
def test_pendulum_driver_on_gpu(gpu):
    _build()
    out = subprocess.run([os.path.join(HOST, "pendulum_ddp"), "50"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "x_f:" in out.stdout


@pytest.mark.gpu
def test_pinocchio_driver_on_gpu(gpu):
    """test/pinocchio_ddp.cpp shape (UR5-like arm, horizon 10, config constraint every step) through the C++ mirror"""
    _build()
    out = subprocess.run([os.path.join(HOST, "pinocchio_ddp"), "10", "1e4", "1e-1", "10", "30"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
