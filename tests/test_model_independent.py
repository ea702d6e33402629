"""CPU: the baked robot model against an INDEPENDENT reading of the reference's URDF.

The oracle and the HIP kernels share csrc/bez_model_gen.h, and tests/rbd_numpy.py reads the same compiler's JSON: a wrong baked
inertia, axis or offset is common-mode for every HIP-vs-oracle test.  Here the numbers come from tests/urdf_independent.py
(its own xml.etree reading of resources/assets/bez/model/soccerbot_stl.urdf -> tests/golden/urdf_bodies.json; 21 separate bodies,
nothing merged) and the dynamics are Kane's projected Newton-Euler equations in world coordinates.  The oracle's bare ABA must
satisfy them, its forward kinematics must put every body where the URDF says, and the test is shown to be sensitive: a 1 %
error in any mass, COM, inertia, joint origin or axis on the URDF side breaks it.
"""
import copy
import json
import os

import numpy as np
import pytest

from oracle.bez_oracle import Oracle
from tests import urdf_independent as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("BEZ_REFERENCE_ROOT", "/root/reference")
G = np.array([0.0, 0.0, float(np.float32(-9.81))])
TOL = 1e-11  # observed worst residual 2.7e-13 N / N*m over 16 poses with velocities (fp64 ABA); a 1 % error in the SMALLEST link inertia shows as 2e-4


@pytest.fixture(scope="module")
def bodies():
    return U.load_fixture()


def _states(n, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
        out.append(dict(pos=np.array([0.1, -0.2, 0.6]), quat=quat, w0=rng.normal(size=3) * 2.0, v0=rng.normal(size=3),
                        q=rng.uniform(-1.0, 1.0, 18), qd=rng.normal(size=18) * 2.0, tau=rng.normal(size=18) * 0.5))
    return out


def _oracle_accelerations(states):
    o = Oracle(num_envs=1)
    acc = []
    for s in states:
        o.set_env_state_f64(0, s["pos"], s["quat"], s["v0"], s["w0"], s["q"], s["qd"])
        a0, qdd, _, _ = o.forward_dynamics(0, 1, s["tau"])  # bare ABA: no drives, armature, limits, contact
        acc.append((np.array(a0), np.array(qdd)))
    return acc


def _residual(bodies, s, a0, qdd):
    dw0 = a0[:3]
    dv0 = a0[3:] + np.cross(s["w0"], s["v0"])  # spatial -> classical acceleration of the base origin
    f = U.generalized_force(bodies, s["pos"], s["quat"], s["w0"], s["v0"], s["q"], s["qd"], dw0, dv0, qdd, G)
    return f - np.concatenate([np.zeros(6), s["tau"]])


def test_fixture_is_the_reference_urdf(bodies):
    """Where the reference tree is present the committed numbers are exactly a fresh independent parse of it."""
    path = os.path.join(REF, U.URDF_REL)
    if not os.path.exists(path):
        pytest.skip("needs the reference tree (build container only)")
    assert json.loads(json.dumps(U.parse_urdf(path))) == bodies


def test_tree_order_and_totals(bodies):
    """Isaac Gym's body order (DFS, children sorted by joint name) from the URDF text reproduces every index the reference
    hard-codes: IMU body 1, feet 12 / 20 (kick_env.py:175-177,193-196), DOF order = the `Joints` enum (kick_env.py:23-41)."""
    names = [b["name"] for b in bodies]
    assert len(bodies) == 21 and names[0] == "/torso" and names[1] == "/imu_link" and names[12] == "/left_foot" and names[20] == "/right_foot"
    assert U.dof_names(bodies) == U.JOINTS_ENUM
    M, com = U.mass_properties(bodies, [0, 0, 0], [0, 0, 0, 1], np.zeros(18))
    assert abs(M - 2.827994) < 1e-9  # 2.828 kg
    model = json.load(open(os.path.join(ROOT, "bez_isaacgym_amd", "model", "bez_model.json")))
    assert abs(model["total_mass"] - M) < 1e-12
    # the compiler's JSON (what rbd_numpy and the kernels' header are made from): same COM at a bent pose
    from tests import rbd_numpy as R
    q = np.linspace(-0.8, 0.9, 18)
    quat = np.array([0.1, -0.2, 0.3, 0.9]); quat /= np.linalg.norm(quat)
    m2 = R.mechanics(model, [0.3, 0.1, 0.5], quat, np.zeros(6), q, np.zeros(18), G)
    M1, com1 = U.mass_properties(bodies, [0.3, 0.1, 0.5], quat, q)
    np.testing.assert_allclose(m2["com"], com1, atol=1e-13)
    # joint axes / origins / limits, body for body, against the compiler's link table (fixed links are merged there)
    by_name = {L["name"]: L for L in model["links"]}
    k = 0
    for b in bodies:
        if b["type"] != "revolute":
            continue
        L = by_name[b["name"]]
        np.testing.assert_array_equal(L["axis"], b["axis"])
        np.testing.assert_array_equal(L["xyz"], b["xyz"])
        assert model["links"][L["parent"]]["name"] == bodies[b["parent"]]["name"]
        lo, hi = min(b["lower"], b["upper"]), max(b["lower"], b["upper"])  # kick_env.py:393-400 swaps inverted limits
        assert model["dof_lower"][k] == lo and model["dof_upper"][k] == hi
        k += 1


def test_oracle_aba_satisfies_kanes_equations_from_the_urdf(bodies):
    """M(q) [a0; qdd] + h(q, v) - g(q) = [0; tau] with M, h, g from the URDF numbers (16 random poses WITH velocities), the
    accelerations from the oracle's ABA on the baked tables."""
    states = _states(16)
    worst = 0.0
    for s, (a0, qdd) in zip(states, _oracle_accelerations(states)):
        worst = max(worst, float(np.abs(_residual(bodies, s, a0, qdd)).max()))
    assert worst < TOL, worst


def test_oracle_body_frames_are_where_the_urdf_puts_them(bodies):
    """rigid-body tensor of the oracle (Isaac layout, 21 robot rows + ball) vs forward kinematics from the URDF numbers."""
    rng = np.random.default_rng(3)
    o = Oracle(num_envs=1)
    for _ in range(4):
        quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
        pos, q = rng.normal(size=3), rng.uniform(-1, 1, 18)
        o.set_env_state_f64(0, pos, quat, np.zeros(3), np.zeros(3), q, np.zeros(18))
        rb = o.rigid_body_states.reshape(-1, 13)
        R, p, _, _ = U.kinematics(bodies, pos, quat, q)
        np.testing.assert_allclose(rb[:21, :3], np.array(p), atol=2e-6)  # the tensor is fp32
        for b in range(21):
            x, y, z, w = rb[b, 3:7].astype(np.float64)
            Rb = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                           [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                           [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            np.testing.assert_allclose(Rb, R[b], atol=2e-6)


def test_the_check_is_sensitive_to_one_percent(bodies):
    """Every mass, COM, inertia tensor, joint origin and joint axis: a 1 % error on one side must show as a residual far above
    the tolerance (so a wrong baked number cannot hide).  The two 10-gram marker links' 1e-9 kg m^2 inertias are exempt."""
    states = _states(6, seed=5)
    acc = _oracle_accelerations(states)

    def worst(bs):
        return max(float(np.abs(_residual(bs, s, a0, qdd)).max()) for s, (a0, qdd) in zip(states, acc))

    assert worst(bodies) < TOL
    missed = []
    for b, B in enumerate(bodies):
        trials = {"mass": lambda X: X.__setitem__("mass", X["mass"] * 1.01)}
        if np.linalg.norm(B["com"]) > 0:
            trials["com"] = lambda X: X.__setitem__("com", [c * 1.01 for c in X["com"]])
        if B["name"] not in ("/imu_link", "/camera"):
            trials["inertia"] = lambda X: X.__setitem__("inertia", (np.array(X["inertia"]) * 1.01).tolist())
        if np.linalg.norm(B["xyz"]) > 0:
            trials["xyz"] = lambda X: X.__setitem__("xyz", [c * 1.01 for c in X["xyz"]])
        if B["type"] == "revolute":
            def tilt(X):
                a = np.array(X["axis"], float)
                t = np.cross(a, [0.3, 0.5, 0.81]); t /= np.linalg.norm(t)
                X["axis"] = (a + 0.01 * t).tolist()
            trials["axis"] = tilt
        for what, mutate in trials.items():
            bs = copy.deepcopy(bodies)
            mutate(bs[b])
            if not worst(bs) > 50 * TOL:
                missed.append((B["name"], what, worst(bs)))
    assert not missed, missed
