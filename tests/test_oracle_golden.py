"""CPU: the oracle's restatement of the reference-owned arithmetic vs the golden vectors produced by the
reference's own TorchScript functions / VecTask.step (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle.bez_oracle import Oracle
from tests import golden_checks as GC


@pytest.fixture(params=["f64", "f32"])
def oracle(request):
    return Oracle(num_envs=GC.N, precision=request.param)


def test_ext_helper_pins(golden):
    """The un-pinned isaacgym.torch_utils helpers [ext] the fixture was generated with: yaw and quat_rotate."""
    q = golden["ext_quat"].astype(np.float64)
    x, y, z, w = q.T
    yaw = np.arctan2(2 * (w * z + x * y), w * w + x * x - y * y - z * z) % (2 * np.pi)
    np.testing.assert_allclose(yaw, golden["ext_euler"][:, 2], atol=2e-6)
    v = golden["ext_vec"].astype(np.float64)
    qv = q[:, :3]
    rot = v * (2 * w ** 2 - 1)[:, None] + 2 * w[:, None] * np.cross(qv, v) + 2 * qv * (qv * v).sum(1, keepdims=True)
    np.testing.assert_allclose(rot, golden["ext_quat_rotate"], atol=2e-6)


def test_imu(oracle, golden): GC.check_imu(oracle, golden)
def test_off_orn(oracle, golden): GC.check_off_orn(oracle, golden)
def test_feet(oracle, golden): GC.check_feet(oracle, golden)
def test_reward_normal(oracle, golden): GC.check_reward(oracle, golden, "normal")
def test_reward_edge(oracle, golden): GC.check_reward(oracle, golden, "edge")
def test_pre_physics(oracle, golden): GC.check_pre_physics(oracle, golden)
def test_step_sequence(oracle, golden): GC.check_step_sequence(oracle, golden)
