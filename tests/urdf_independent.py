"""Test-only: a second, independent reading of the reference's robot description, and dynamics built from it in a formulation
that shares nothing with the model compiler, the oracle or tests/rbd_numpy.py.

Why: oracle/bez_oracle.c and the HIP kernels both include csrc/bez_model_gen.h (written by model/compile_model.py), and
tests/rbd_numpy.py reads the same compiler's bez_model.json -- a wrong baked inertia, axis or offset would be invisible to every
HIP-vs-oracle test.  Here the URDF text is parsed again with xml.etree, NOTHING is merged or
re-expressed (all 21 bodies stay separate, fixed joints included), and the equations of motion are Kane's / projected
Newton-Euler: body accelerations from world-frame kinematics, then  sum_b J_b^T [n_b; f_b] = generalized force.

  parse_urdf(path)           -> list of bodies in Isaac Gym order (DFS from the root, children sorted by joint name)
  load_fixture() / FIXTURE   -> the same list from tests/golden/urdf_bodies.json (numbers only; made by make_urdf_fixture.py)
  generalized_force(bodies, state, acc, gravity) -> (24,) residual-ready vector [base wrench about the base origin (6); joint torques (18)]
"""
import json
import os
import xml.etree.ElementTree as ET

import numpy as np

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "urdf_bodies.json")
URDF_REL = os.path.join("resources", "assets", "bez", "model", "soccerbot_stl.urdf")

# DOF order the reference hard-codes (bez_isaacgym/tasks/kick_env.py:23-41, `Joints` enum)
JOINTS_ENUM = ["head_motor_0", "head_motor_1", "left_arm_motor_0", "left_arm_motor_1",
               "left_leg_motor_0", "left_leg_motor_1", "left_leg_motor_2", "left_leg_motor_3", "left_leg_motor_4", "left_leg_motor_5",
               "right_arm_motor_0", "right_arm_motor_1",
               "right_leg_motor_0", "right_leg_motor_1", "right_leg_motor_2", "right_leg_motor_3", "right_leg_motor_4", "right_leg_motor_5"]


def _floats(s):
    return [float(t) for t in s.split()]


def _rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def parse_urdf(path):
    root = ET.parse(path).getroot()
    links = {}
    for L in root.findall("link"):
        ine = L.find("inertial")
        o = ine.find("origin")
        R = _rpy(*_floats(o.get("rpy", "0 0 0")))
        I = ine.find("inertia")
        g = lambda k: float(I.get(k))
        Ic = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
        links[L.get("name")] = dict(mass=float(ine.find("mass").get("value")), com=_floats(o.get("xyz", "0 0 0")), inertia=(R @ Ic @ R.T).tolist())
    children, has_parent = {}, set()
    for J in root.findall("joint"):
        o = J.find("origin")
        rec = dict(joint=J.get("name"), type=J.get("type"), child=J.find("child").get("link"), xyz=_floats(o.get("xyz", "0 0 0")),
                   rpy=_floats(o.get("rpy", "0 0 0")), axis=_floats(J.find("axis").get("xyz")) if J.find("axis") is not None else [0, 0, 0])
        lim = J.find("limit")
        rec["lower"], rec["upper"] = (float(lim.get("lower", 0)), float(lim.get("upper", 0))) if lim is not None else (0.0, 0.0)
        children.setdefault(J.find("parent").get("link"), []).append(rec)
        has_parent.add(rec["child"])
    roots = [n for n in links if n not in has_parent]
    assert len(roots) == 1, roots
    bodies = []

    def visit(name, parent, rec):
        idx = len(bodies)
        b = dict(name=name, parent=parent, **links[name])
        if rec is not None:
            assert all(abs(v) < 1e-12 for v in rec["rpy"]), "joint frames are assumed unrotated (SURVEY appendix A)"
            b.update(joint=rec["joint"], type=rec["type"], xyz=rec["xyz"], axis=rec["axis"], lower=rec["lower"], upper=rec["upper"])
        else:
            b.update(joint=None, type="floating", xyz=[0, 0, 0], axis=[0, 0, 0], lower=0.0, upper=0.0)
        bodies.append(b)
        for r in sorted(children.get(name, []), key=lambda r: r["joint"]):
            visit(r["child"], idx, r)

    visit(roots[0], -1, None)
    return bodies


def load_fixture():
    return json.load(open(FIXTURE))["bodies"]


def dof_names(bodies):
    return [b["joint"] for b in bodies if b["type"] == "revolute"]


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def _rot(axis, th):
    a = np.asarray(axis, float)
    a = a / np.linalg.norm(a)
    K = _skew(a)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def _quat_R(q):  # xyzw
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def kinematics(bodies, root_pos, root_quat, q):
    """World pose of every body frame: (R[b], p[b]) and of every revolute joint: world axis, dof index."""
    nb = len(bodies)
    R, p, ax, dof = [None] * nb, [None] * nb, [None] * nb, [-1] * nb
    k = 0
    for b, B in enumerate(bodies):
        if B["parent"] < 0:
            R[b], p[b] = _quat_R(root_quat), np.asarray(root_pos, float)
            continue
        P = B["parent"]
        p[b] = p[P] + R[P] @ np.asarray(B["xyz"], float)
        if B["type"] == "revolute":
            a = np.asarray(B["axis"], float)
            ax[b] = R[P] @ (a / np.linalg.norm(a))
            R[b] = R[P] @ _rot(a, q[k])
            dof[b] = k
            k += 1
        else:
            R[b] = R[P]
    return R, p, ax, dof


def generalized_force(bodies, root_pos, root_quat, w0, v0, q, qd, dw0, dv0, qdd, gravity):
    """Kane's equations.  (w0, v0) = angular velocity / velocity of the base ORIGIN in world axes, (dw0, dv0) their CLASSICAL time
    derivatives.  Returns the generalized inertia + bias + gravity force that the applied generalized force must equal:
    [moment about the base origin (3), force (3), joint torques (18)]."""
    nb = len(bodies)
    R, p, ax, dof = kinematics(bodies, root_pos, root_quat, q)
    w, dw, v, dv = [None] * nb, [None] * nb, [None] * nb, [None] * nb   # of each body frame origin, world axes
    anc = [None] * nb                                                  # revolute bodies on the path root -> b
    for b, B in enumerate(bodies):
        P = B["parent"]
        if P < 0:
            w[b], dw[b], v[b], dv[b], anc[b] = np.asarray(w0, float), np.asarray(dw0, float), np.asarray(v0, float), np.asarray(dv0, float), []
            continue
        r = p[b] - p[P]
        v[b] = v[P] + np.cross(w[P], r)
        dv[b] = dv[P] + np.cross(dw[P], r) + np.cross(w[P], np.cross(w[P], r))
        if dof[b] >= 0:
            k = dof[b]
            w[b] = w[P] + ax[b] * qd[k]
            dw[b] = dw[P] + ax[b] * qdd[k] + np.cross(w[P], ax[b]) * qd[k]
            anc[b] = anc[P] + [b]
        else:
            w[b], dw[b], anc[b] = w[P], dw[P], anc[P]
    out = np.zeros(6 + sum(1 for d in dof if d >= 0))
    g = np.asarray(gravity, float)
    for b, B in enumerate(bodies):
        m = B["mass"]
        c = p[b] + R[b] @ np.asarray(B["com"], float)
        rc = c - p[b]
        ac = dv[b] + np.cross(dw[b], rc) + np.cross(w[b], np.cross(w[b], rc))
        Iw = R[b] @ np.asarray(B["inertia"], float) @ R[b].T
        f = m * (ac - g)
        n = Iw @ dw[b] + np.cross(w[b], Iw @ w[b])
        out[0:3] += n + np.cross(c - p[0], f)
        out[3:6] += f
        for j in anc[b]:
            out[6 + dof[j]] += ax[j] @ (n + np.cross(c - p[j], f))
    return out


def mass_properties(bodies, root_pos, root_quat, q):
    R, p, _, _ = kinematics(bodies, root_pos, root_quat, q)
    M = sum(B["mass"] for B in bodies)
    com = sum(B["mass"] * (p[b] + R[b] @ np.asarray(B["com"], float)) for b, B in enumerate(bodies)) / M
    return M, com
