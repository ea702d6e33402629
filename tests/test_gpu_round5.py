"""Round-5 GPU tests (through the C ABI).  All tests need a GPU: `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _getup(backend, name, n):
    import json, os
    from bez_isaacgym_amd import abi
    from tests.scenarios import ROOT, lay_down, make_backend, play
    model = json.load(open(os.path.join(ROOT, "bez_isaacgym_amd", "model", "bez_model.json")))
    sim = make_backend(backend, abi.default_config(n, seed=7))
    sim.step(np.zeros((n, 18), np.float32))
    lay_down(sim, n, name, np.random.default_rng(3))
    return play(sim, n, name, model)


@pytest.mark.parametrize("name", ["getupfront", "getupback", "getupside"])
def test_getup_scenarios_hip_equals_oracle(name):
    """The reference's get-up tables from lying starts (tests/scenarios.py) in 64 HIP envs through the split entry points
    (bez_sim_pre_physics + bez_sim_simulate): a thousand control steps on knees, forearms and the torso's guard points -- contact
    the bez_kick episodes never reach before their fall reset.  The HIP kernels and the fp64 oracle must tell the same story
    (medians over the envs; the envs differ by +-0.02 rad in their start pose)."""
    h = _getup("hip", name, 64)
    o = _getup("oracle", name, 8)
    assert h["finite"] == 1.0, h
    for k, tol in (("final_z", 0.02), ("max_z", 0.02), ("final_up", 0.15), ("max_up", 0.1)):
        assert abs(h[k] - o[k]) < tol, (k, h, o)
    if name == "getupfront":
        assert h["max_z"] > 0.20 and h["max_up"] > 0.75, h   # the squat on the feet is reached


@pytest.mark.xfail(strict=True, reason="measured: 0 of 64 HIP envs stand at the end of simulation_getupfront (the squat is reached; the last key frame tips "
                                       "the torso over the toes) -- the same in the oracle, at damping 2 and at effort 5 N*m: profiles/r05_getup.txt")
def test_getup_front_ends_standing_hip():
    assert _getup("hip", "getupfront", 64)["standing"] >= 0.9


def test_dof_sweep_hip(model):
    """`test_motor_action_agent` (bez_isaacgym/test/test_kick_env.py:142-186) on the HIP simulator: every DOF to its lower limit, its
    upper limit and back, zero gravity, floating base; the same bars as the oracle's run (tests/test_scenarios.py)."""
    from bez_isaacgym_amd import abi
    from tests.scenarios import dof_sweep, make_backend
    from tests.test_scenarios import check_dof_sweep
    n = 64
    cfg = abi.default_config(n, seed=3)
    cfg.gravity[:] = [0.0, 0.0, 0.0]
    sim = make_backend("hip", cfg)
    sim.step(np.zeros((n, 18), np.float32))
    check_dof_sweep(dof_sweep(sim, n, model))
