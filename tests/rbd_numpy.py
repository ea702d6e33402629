"""Independent rigid-body dynamics in numpy (test-only): Featherstone's RNEA for a floating base in
LINK-LOCAL coordinates with Pluecker transforms (RBDA, ch. 5 / table 5.1), plus momentum / energy sums.

Deliberately a different formulation from both the oracle (world-aligned about the torso origin, forward
dynamics) and the HIP kernels, so that agreement is evidence and not tautology.  Reads only the model JSON.
"""
import numpy as np


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


def quat_to_mat(q):  # xyzw, body -> world
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rot_axis(a, th):
    a = np.asarray(a, float)
    K = skew(a)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def plux(E, r):
    """Motion transform A->B where B's origin is at r (A coords) and E maps A coords to B coords."""
    X = np.zeros((6, 6))
    X[:3, :3] = E
    X[3:, 3:] = E
    X[3:, :3] = -E @ skew(r)
    return X


def crm(v):
    X = np.zeros((6, 6))
    X[:3, :3] = skew(v[:3]); X[3:, 3:] = skew(v[:3]); X[3:, :3] = skew(v[3:])
    return X


def crf(v):
    return -crm(v).T


def link_inertia_local(L):
    m = L["mass"]
    c = np.array(L["com"])
    xx, yy, zz, xy, xz, yz = L["inertia"]
    Ic = np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])
    C = skew(c)
    I = np.zeros((6, 6))
    I[:3, :3] = Ic + m * C @ C.T
    I[:3, 3:] = m * C
    I[3:, :3] = m * C.T
    I[3:, 3:] = m * np.eye(3)
    return I


def rnea_floating(model, root_quat, v0_world, a0_world_spatial, q, qd, qdd, gravity):
    """Inverse dynamics.  v0_world = [w; v] of the torso origin (world axes); a0_world_spatial = spatial
    acceleration of the torso about its origin (world axes).  Returns (f0 (6, base coords), tau (18))."""
    links = model["links"]
    n = len(links)
    E0 = quat_to_mat(root_quat).T  # world -> base
    R0 = np.zeros((6, 6)); R0[:3, :3] = E0; R0[3:, 3:] = E0
    v = [None] * n; a = [None] * n; f = [None] * n; Xup = [None] * n; S = [None] * n
    v[0] = R0 @ v0_world
    ag = np.concatenate([np.zeros(3), gravity])
    a[0] = R0 @ (a0_world_spatial - ag)
    I = [link_inertia_local(L) for L in links]
    f[0] = I[0] @ a[0] + crf(v[0]) @ I[0] @ v[0]
    for i in range(1, n):
        L = links[i]
        p = L["parent"]
        axis = np.array(L["axis"])
        XJ = plux(rot_axis(axis, q[i - 1]).T, np.zeros(3))
        XT = plux(np.eye(3), np.array(L["xyz"]))
        Xup[i] = XJ @ XT
        S[i] = np.concatenate([axis, np.zeros(3)])
        vJ = S[i] * qd[i - 1]
        v[i] = Xup[i] @ v[p] + vJ
        a[i] = Xup[i] @ a[p] + S[i] * qdd[i - 1] + crm(v[i]) @ vJ
        f[i] = I[i] @ a[i] + crf(v[i]) @ I[i] @ v[i]
    tau = np.zeros(n - 1)
    for i in range(n - 1, 0, -1):
        tau[i - 1] = S[i] @ f[i]
        p = links[i]["parent"]
        f[p] = f[p] + Xup[i].T @ f[i]
    return f[0], tau


def mechanics(model, root_pos, root_quat, v0_world, q, qd, gravity):
    """Total mass, COM, linear momentum, angular momentum about the world origin, kinetic and potential energy."""
    links = model["links"]
    n = len(links)
    E = [None] * n; r = [None] * n; w = [None] * n; vo = [None] * n
    E[0] = quat_to_mat(root_quat); r[0] = np.array(root_pos, float)
    w[0] = np.array(v0_world[:3], float); vo[0] = np.array(v0_world[3:], float)
    M = 0.0; P = np.zeros(3); Lang = np.zeros(3); KE = 0.0; PE = 0.0; mc = np.zeros(3)
    for i in range(n):
        L = links[i]
        if i > 0:
            p = L["parent"]
            axis = np.array(L["axis"])
            r[i] = r[p] + E[p] @ np.array(L["xyz"])
            E[i] = E[p] @ rot_axis(axis, q[i - 1])
            aw = E[p] @ axis
            w[i] = w[p] + aw * qd[i - 1]
            vo[i] = vo[p] + np.cross(w[p], r[i] - r[p])
        m = L["mass"]
        c = r[i] + E[i] @ np.array(L["com"])
        vc = vo[i] + np.cross(w[i], c - r[i])
        xx, yy, zz, xy, xz, yz = L["inertia"]
        Ic = E[i] @ np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]]) @ E[i].T
        M += m; mc += m * c; P += m * vc
        Lang += np.cross(c, m * vc) + Ic @ w[i]
        KE += 0.5 * m * vc @ vc + 0.5 * w[i] @ Ic @ w[i]
        PE += -m * np.dot(gravity, c)
    return dict(mass=M, com=mc / M, P=P, L=Lang, KE=KE, PE=PE)
