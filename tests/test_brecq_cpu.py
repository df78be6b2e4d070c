"""BRECQ / AdaRound stage on CPU (stand-in kernels = autograd of the reference formulas) against the reference's own
forward/backward captured in tests/golden/brecq_toy.npz, plus an end-to-end reconstruct_model smoke."""
import pytest

from adalog_amd import backend
from tests import cpu_backend, layer_cases as LC


@pytest.fixture(autouse=True)
def _cpu_backend():
    backend.set_backend(cpu_backend)
    yield
    backend.set_backend(None)


def test_brecq_toy_forward_backward(golden):
    LC.case_brecq_toy(golden)


def test_brecq_trajectory_matches_reference_loop(golden):
    LC.case_brecq_traj(golden)


def test_brecq_trajectory_with_the_one_launch_adam(golden, monkeypatch):
    """The loop with HipAdam (host logic over the CPU specification of adalog_adam_multi) lands on the reference's trajectory."""
    monkeypatch.setenv("ADALOG_BRECQ_ADAM", "hip")
    LC.case_brecq_traj(golden)


def test_brecq_reconstruct_model():
    LC.case_brecq_reconstruct(iters=30)
