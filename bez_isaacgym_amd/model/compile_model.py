#!/usr/bin/env python3
"""Offline model compiler: Bez URDF + ball URDF + bez_kick.yaml -> flat constant tables.

Runs ONLY in the build container (it reads /root/reference); the GPU box never parses
URDF.  Outputs (both committed, both pure data):

  bez_isaacgym_amd/csrc/bez_model_gen.h   constant tables for the HIP kernels and the C oracle
  bez_isaacgym_amd/model/bez_model.json   same numbers for Python-side tests (numpy CRBA/RNEA)

What is baked (reference file:line each number comes from):
  * tree, joint origins/axes/limits, link mass/COM/inertia, leg collision boxes
      resources/assets/bez/model/soccerbot_stl.urdf:34-587
  * ball mass/inertia/radius
      resources/assets/objects/ball.urdf:9-28
  * ready pose (default DOF targets), init states, goal, drive gains, sim rate
      bez_isaacgym/cfg/task/bez_kick.yaml:11-147
  * per-DOF overrides stiffness/damping/armature/velocity/friction/effort
      bez_isaacgym/tasks/kick_env.py:322-329
Body order = Isaac Gym order [ext]: depth-first from the root, children sorted by joint name
('/' < letters so '/torso_imu' first).  It reproduces the indices the reference hard-codes:
IMU body 1 (kick_env.py:175-177), feet 12/20 (kick_env.py:193-196), DOF order = Joints enum
(kick_env.py:23-41).  Fixed links (/imu_link, /camera) stay output bodies but their inertia
is merged into the parent for the dynamics.
"""
import json
import math
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np
import yaml

REF = os.environ.get("BEZ_REFERENCE_ROOT", "/root/reference")
URDF_BEZ = os.path.join(REF, "resources/assets/bez/model/soccerbot_stl.urdf")
URDF_BALL = os.path.join(REF, "resources/assets/objects/ball.urdf")
YAML_TASK = os.path.join(REF, "bez_isaacgym/cfg/task/bez_kick.yaml")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_H = os.path.join(HERE, "..", "csrc", "bez_model_gen.h")
OUT_JSON = os.path.join(HERE, "bez_model.json")


def _vec(s, n=3):
    v = [float(x) for x in s.split()]
    assert len(v) == n, s
    return v


def parse_urdf(path):
    root = ET.parse(path).getroot()
    links = {}
    for ln in root.findall("link"):
        name = ln.get("name")
        inert = ln.find("inertial")
        mass = float(inert.find("mass").get("value"))
        org = inert.find("origin")
        com = _vec(org.get("xyz"))
        assert all(abs(x) < 1e-12 for x in _vec(org.get("rpy"))), "inertial rpy must be 0"
        I = inert.find("inertia")
        inertia = [float(I.get(k)) for k in ("ixx", "iyy", "izz", "ixy", "ixz", "iyz")]
        box = None
        col = ln.find("collision")
        sphere = None
        if col is not None:
            g = col.find("geometry")
            corg = col.find("origin")
            cxyz = _vec(corg.get("xyz")) if corg is not None else [0, 0, 0]
            if corg is not None:
                assert all(abs(x) < 1e-12 for x in _vec(corg.get("rpy"))), "collision rpy must be 0"
            if g.find("box") is not None:
                size = _vec(g.find("box").get("size"))
                box = {"center": cxyz, "half": [0.5 * s for s in size]}
            if g.find("sphere") is not None:
                sphere = float(g.find("sphere").get("radius"))
        links[name] = {"mass": mass, "com": com, "inertia": inertia, "box": box, "sphere": sphere}
    joints = []
    for jn in root.findall("joint"):
        org = jn.find("origin")
        assert all(abs(x) < 1e-12 for x in _vec(org.get("rpy"))), "joint rpy must be 0"
        lim = jn.find("limit")
        joints.append({
            "name": jn.get("name"), "type": jn.get("type"),
            "parent": jn.find("parent").get("link"), "child": jn.find("child").get("link"),
            "xyz": _vec(org.get("xyz")),
            "axis": _vec(jn.find("axis").get("xyz")) if jn.find("axis") is not None else [0, 0, 0],
            "lower": float(lim.get("lower")) if lim is not None else 0.0,
            "upper": float(lim.get("upper")) if lim is not None else 0.0,
        })
    return links, joints


def inertia_mat(v):
    xx, yy, zz, xy, xz, yz = v
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]], dtype=np.float64)


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


def merge_inertia(m1, c1, I1, m2, c2, I2):
    """Combine two rigid bodies given (mass, com, inertia about own com), same axes."""
    m = m1 + m2
    c = (m1 * np.asarray(c1) + m2 * np.asarray(c2)) / m
    def shift(mm, cc, II):
        d = np.asarray(cc) - c
        return II + mm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    return m, c, shift(m1, c1, I1) + shift(m2, c2, I2)


def build(urdf=None, cleats=False, box=False):
    links, joints = parse_urdf(urdf or URDF_BEZ)
    ball_links, _ = parse_urdf(URDF_BALL)
    cfg = yaml.safe_load(open(YAML_TASK))
    env = cfg["env"]

    children = {}
    for j in joints:
        children.setdefault(j["parent"], []).append(j)
    child_names = {j["child"] for j in joints}
    roots = [n for n in links if n not in child_names]
    assert roots == ["/torso"], roots

    # Isaac order: DFS, children sorted by joint name
    bodies = []  # dicts: name, parent_body, joint
    def dfs(name, parent_idx, joint):
        idx = len(bodies)
        bodies.append({"name": name, "parent": parent_idx, "joint": joint})
        for j in sorted(children.get(name, []), key=lambda jj: jj["name"]):
            dfs(j["child"], idx, j)
    dfs("/torso", -1, None)
    names = [b["name"] for b in bodies]
    if cleats:  # kick_env.py:187-191: contact rows 13:17 are the left cleats, 25:29 the right ones
        assert len(bodies) == 29 and names[12] == "/left_foot" and names[24] == "/right_foot", names
        assert names[13:17] == ["/left_foot_cleat_%d" % i for i in (4, 5, 6, 7)] and names[25:29] == ["/right_foot_cleat_%d" % i for i in range(4)]
    else:
        assert len(bodies) == 21
        assert names[1] == "/imu_link" and names[12] == "/left_foot" and names[20] == "/right_foot", names

    # dynamic links: root + revolute children; fixed children merged into parent link
    link_of_body = [None] * len(bodies)
    body_off = [[0.0, 0.0, 0.0] for _ in bodies]  # body origin in its link frame
    dyn = []  # link dicts
    for bi, b in enumerate(bodies):
        j = b["joint"]
        L = links[b["name"]]
        if j is None or j["type"] == "revolute":
            li = len(dyn)
            link_of_body[bi] = li
            dyn.append({
                "name": b["name"], "body": bi,
                "parent": -1 if j is None else link_of_body[b["parent"]],
                "joint_name": None if j is None else j["name"],
                "axis": [0.0, 0.0, 0.0] if j is None else j["axis"],
                "xyz": [0.0, 0.0, 0.0] if j is None else j["xyz"],
                "lower": 0.0 if j is None else j["lower"], "upper": 0.0 if j is None else j["upper"],
                "mass": L["mass"], "com": np.array(L["com"]), "I": inertia_mat(L["inertia"]),
                "box": L["box"],
            })
            if j is not None:
                assert bodies[b["parent"]]["joint"] is None or bodies[b["parent"]]["joint"]["type"] == "revolute", \
                    "revolute child of a fixed body not supported"
        else:
            assert j["type"] == "fixed"
            pl = link_of_body[b["parent"]]
            link_of_body[bi] = pl
            off = (np.array(body_off[b["parent"]]) + np.array(j["xyz"])).tolist()
            body_off[bi] = off
            d = dyn[pl]
            m, c, I = merge_inertia(d["mass"], d["com"], d["I"], L["mass"], np.array(off) + np.array(L["com"]),
                                    inertia_mat(L["inertia"]))
            d["mass"], d["com"], d["I"] = m, c, I
    assert len(dyn) == 19
    # DOF d <-> link d+1; check against the reference's Joints enum order (kick_env.py:23-41)
    dof_names = [d["joint_name"] for d in dyn[1:]]
    expect = (["head_motor_0", "head_motor_1", "left_arm_motor_0", "left_arm_motor_1"] +
              ["left_leg_motor_%d" % i for i in range(6)] + ["right_arm_motor_0", "right_arm_motor_1"] +
              ["right_leg_motor_%d" % i for i in range(6)])
    assert dof_names == expect, dof_names

    lower, upper, default = [], [], []
    for d in dyn[1:]:
        lo, hi = d["lower"], d["upper"]
        if lo > hi:  # kick_env.py:393-400
            lo, hi = hi, lo
        lower.append(lo); upper.append(hi)
        default.append(float(env["readyJointAngles"][d["joint_name"]]))

    # collision boxes used for ball contact: foot, ankle, calve, thigh, hip_front per leg (+ the torso box, appended below)
    boxes = []
    for li, d in enumerate(dyn):
        if d["box"] is not None and min(d["box"]["half"]) > 1e-3 and "cleat" not in d["name"] and ("_leg" in (d["joint_name"] or "") if box else True):
            boxes.append({"link": li, "center": d["box"]["center"], "half": d["box"]["half"]})
    # ball <-> torso: the torso collision mesh is approximated by its bounding box (same vertex data as the guard points below);
    # the soccerbot_box*.urdf variants (asset.stl: False, kick_env.py:272-276) give the torso an explicit box
    if box:
        tb = links["/torso"]["box"]
        boxes.append({"link": 0, "center": list(tb["center"]), "half": list(tb["half"])})
    else:
        boxes.append({"link": 0, "center": [0.012, 0.0, -0.033], "half": [0.052, 0.0725, 0.095]})
    # ground contact points: foot box bottom corners + guard points on the torso and on the other
    # chain-end links (head, forearms).  Mid-chain links carry no ground points: an episode ends at
    # torso z < 0.275 (kick_env.py:1331) long before a knee could reach the floor.
    # Upper-body meshes are approximated by bounding-box corners measured from the .dae vertex data
    # (torso x[-.040,.064] y[+-.0725] z[-.128,.062]; head z top .062; forearm tip z -.131).
    points = []
    def add(link_name, p, kind):
        li = [i for i, d in enumerate(dyn) if d["name"] == link_name][0]
        points.append({"link": li, "p": [float(x) for x in p], "kind": kind})
    for side in ("left", "right"):
        if cleats:
            # the four cleats of a foot (1 cm x 1 cm x 2 mm boxes, 2 mm proud of the foot plate) carry the ground contact:
            # one point at the bottom centre of each, reported in the cleat body's own contact row
            for bi, b in enumerate(bodies):
                if b["name"].startswith("/%s_foot_cleat" % side):
                    cb = links[b["name"]]["box"]
                    add("/%s_foot" % side, [cb["center"][0], cb["center"][1], cb["center"][2] - cb["half"][2]], "cleat")
                    points[-1]["body"] = bi
        else:
            fb = links["/%s_foot" % side]["box"]
            c, h = fb["center"], fb["half"]
            for sx in (+1, -1):
                for sy in (+1, -1):
                    add("/%s_foot" % side, [c[0] + sx * h[0], c[1] + sy * h[1], c[2] - h[2]], "foot")
    if box:
        # box assets: the same guard points taken from the URDF's own collision boxes -- the torso box's eight corners, the
        # four top corners of the head box, the bottom-face centre of each forearm box
        tb, hb = links["/torso"]["box"], links["/head"]["box"]
        for sx in (-1, 1):
            for sy in (-1, 1):
                for sz in (-1, 1):
                    add("/torso", [tb["center"][k] + sgn * tb["half"][k] for k, sgn in enumerate((sx, sy, sz))], "guard")
        for sx in (-1, 1):
            for sy in (-1, 1):
                add("/head", [hb["center"][0] + sx * hb["half"][0], hb["center"][1] + sy * hb["half"][1], hb["center"][2] + hb["half"][2]], "guard")
        for side in ("left", "right"):
            fb = links["/%s_forearm" % side]["box"]
            add("/%s_forearm" % side, [fb["center"][0], fb["center"][1], fb["center"][2] - fb["half"][2]], "guard")
    else:
        for sx in (-0.040, 0.064):
            for sy in (-0.0725, 0.0725):
                for sz in (-0.128, 0.062):
                    add("/torso", [sx, sy, sz], "guard")
        hc = [-0.013, 0.0, 0.025]  # head collision origin (urdf:530)
        for sx in (-0.04775, 0.02345):
            for sy in (-0.04555, 0.04845):
                add("/head", [hc[0] + sx, hc[1] + sy, hc[2] + 0.0619], "guard")
        for side, sgn in (("left", 1.0), ("right", -1.0)):
            add("/%s_forearm" % side, [-0.0055 - 0.005, sgn * (0.005 + 0.0245), -0.131], "guard")

    # leg <-> leg self-collision shapes (the reference enables self-collision: kick_env.py:365-366, collision_filter 0):
    # every leg box becomes a capsule along its longest axis with the larger cross-section half-extent as radius;
    # the 3 mm foot plate becomes two thin capsules along its long edges.  Pairs: left x right, the hip boxes only
    # against the other hip box / thigh (they cannot reach further down).
    capsules = []
    for b in boxes:
        if b["link"] == 0:
            continue
        h, c = b["half"], b["center"]
        ax = int(np.argmax(h))
        o1, o2 = [i for i in range(3) if i != ax]
        side = "L" if dyn[b["link"]]["name"].startswith("/left") else "R"
        short = dyn[b["link"]]["name"].split("_", 1)[1]
        if min(h) < 3e-3:
            mid = o1 if h[o1] > h[o2] else o2
            r = 0.01
            for sgn in (+1.0, -1.0):
                p0, p1 = list(c), list(c)
                p0[mid] += sgn * (h[mid] - r); p1[mid] += sgn * (h[mid] - r)
                p0[ax] -= h[ax] - r; p1[ax] += h[ax] - r
                capsules.append({"link": b["link"], "p0": p0, "p1": p1, "r": r, "side": side, "part": short})
        else:
            r = max(h[o1], h[o2])
            hl = max(h[ax] - r, 0.0)
            p0, p1 = list(c), list(c)
            p0[ax] -= hl; p1[ax] += hl
            capsules.append({"link": b["link"], "p0": p0, "p1": p1, "r": r, "side": side, "part": short})
    cap_pairs = []
    for i, a in enumerate(capsules):
        for j, b in enumerate(capsules):
            if a["side"] == "L" and b["side"] == "R":
                if "hip_front" in (a["part"], b["part"]) and not ({a["part"], b["part"]} <= {"hip_front", "thigh"}):
                    continue
                cap_pairs.append([i, j])

    ball = ball_links["base_link"]
    for pt in points:
        pt.setdefault("body", dyn[pt["link"]]["body"])
    model = {
        "num_bodies": len(bodies), "num_links": 19, "num_dofs": 18,
        "body_names": names, "dof_names": dof_names,
        "body_link": link_of_body, "body_offset": body_off,
        "links": [{
            "name": d["name"], "body": d["body"], "parent": d["parent"], "axis": d["axis"], "xyz": d["xyz"],
            "mass": d["mass"], "com": d["com"].tolist(),
            "inertia": [d["I"][0, 0], d["I"][1, 1], d["I"][2, 2], d["I"][0, 1], d["I"][0, 2], d["I"][1, 2]],
        } for d in dyn],
        "dof_lower": lower, "dof_upper": upper, "dof_default": default,
        "boxes": boxes, "ground_points": points,
        "capsules": [{k: c[k] for k in ("link", "p0", "p1", "r")} for c in capsules], "capsule_pairs": cap_pairs,
        "ball": {"mass": ball["mass"], "inertia": ball["inertia"][0], "radius": ball["sphere"]},
        "cfg": {
            "dt": float(cfg["sim"]["dt"]), "substeps": int(cfg["sim"]["substeps"]),
            "gravity": [float(x) for x in cfg["sim"]["gravity"]],
            "kp": float(env["control"]["stiffness"]), "kd": float(env["control"]["damping"]),
            "armature": float(env["urdfAsset"]["armature"]),
            "effort": 2.5, "vel_limit": 2.0 * math.pi, "joint_friction": 0.1,  # kick_env.py:327-329
            "plane_friction": float(env["plane"]["dynamicFriction"]),
            "clip_actions": float(env["clipActions"]),
            "episode_length_s": float(env["learn"]["episodeLength_s"]),
            "bez_init": [float(x) for x in env["bezInitState"]["pos"] + env["bezInitState"]["rot"]],
            "ball_init": [float(x) for x in env["ballInitState"]["pos"] + env["ballInitState"]["rot"]],
            "goal": [float(x) for x in env["goalState"]["goal"]],
        },
        "total_mass": float(sum(d["mass"] for d in dyn)),
    }
    return model


def fmt(x):
    return repr(float(x))


def arr(vals):
    return "{" + ", ".join(fmt(v) for v in vals) + "}"


def emit_header(m, mc=None, mb=None, mbc=None):
    L = m["links"]
    o = []
    o.append("/* GENERATED by bez_isaacgym_amd/model/compile_model.py -- do not edit.\n"
             " * Flat model constants for the Bez humanoid as bez_kick loads it (soccerbot_stl.urdf,\n"
             " * ball.urdf, bez_kick.yaml).  Pure data: shared by the HIP kernels and the C oracle. */")
    o.append("#ifndef BEZ_MODEL_GEN_H\n#define BEZ_MODEL_GEN_H")
    o.append("#if defined(__cplusplus)\n#define BEZ_TBL static constexpr\n#else\n#define BEZ_TBL static const\n#endif")
    o.append("#define BEZ_NB 21      /* robot rigid bodies, Isaac order */")
    o.append("#define BEZ_NBE 22     /* bodies per env incl. ball */")
    o.append("#define BEZ_NL 19      /* dynamic links (fixed children merged) */")
    o.append("#define BEZ_ND 18      /* actuated DOFs; DOF d drives link d+1 */")
    o.append("#define BEZ_NBOX %d" % len(m["boxes"]))
    o.append("#define BEZ_NPT %d" % len(m["ground_points"]))
    o.append("#define BEZ_IMU_BODY 1\n#define BEZ_LFOOT_BODY 12\n#define BEZ_RFOOT_BODY 20")
    o.append("#define BEZ_LFOOT_LINK %d\n#define BEZ_RFOOT_LINK %d" % (m["body_link"][12], m["body_link"][20]))
    def axis_code(a):
        for k in range(3):
            if abs(abs(a[k]) - 1.0) < 1e-12:
                return int(math.copysign(k + 1, a[k]))
        return 0
    o.append("BEZ_TBL int BEZ_LINK_PARENT[BEZ_NL] = {%s};" % ", ".join(str(l["parent"]) for l in L))
    o.append("BEZ_TBL int BEZ_LINK_BODY[BEZ_NL] = {%s};" % ", ".join(str(l["body"]) for l in L))
    o.append("/* joint axis code: +-1 = +-x, +-2 = +-y, +-3 = +-z, 0 = root */")
    o.append("BEZ_TBL int BEZ_LINK_AXIS[BEZ_NL] = {%s};" % ", ".join(str(axis_code(l["axis"])) for l in L))
    o.append("BEZ_TBL double BEZ_LINK_AXIS_VEC[BEZ_NL][3] = {%s};" % ", ".join(arr(l["axis"]) for l in L))
    o.append("BEZ_TBL double BEZ_LINK_XYZ[BEZ_NL][3] = {%s};" % ", ".join(arr(l["xyz"]) for l in L))
    o.append("BEZ_TBL double BEZ_LINK_MASS[BEZ_NL] = %s;" % arr(l["mass"] for l in L))
    o.append("BEZ_TBL double BEZ_LINK_COM[BEZ_NL][3] = {%s};" % ", ".join(arr(l["com"]) for l in L))
    o.append("/* inertia about COM, link axes: xx yy zz xy xz yz */")
    o.append("BEZ_TBL double BEZ_LINK_INERTIA[BEZ_NL][6] = {%s};" % ", ".join(arr(l["inertia"]) for l in L))
    o.append("BEZ_TBL int BEZ_BODY_LINK[BEZ_NB] = {%s};" % ", ".join(str(x) for x in m["body_link"]))
    o.append("BEZ_TBL double BEZ_BODY_OFFSET[BEZ_NB][3] = {%s};" % ", ".join(arr(x) for x in m["body_offset"]))
    o.append("BEZ_TBL double BEZ_DOF_LOWER[BEZ_ND] = %s;" % arr(m["dof_lower"]))
    o.append("BEZ_TBL double BEZ_DOF_UPPER[BEZ_ND] = %s;" % arr(m["dof_upper"]))
    o.append("BEZ_TBL double BEZ_DOF_DEFAULT[BEZ_ND] = %s;" % arr(m["dof_default"]))
    o.append("BEZ_TBL int BEZ_BOX_LINK[BEZ_NBOX] = {%s};" % ", ".join(str(b["link"]) for b in m["boxes"]))
    o.append("BEZ_TBL double BEZ_BOX_CENTER[BEZ_NBOX][3] = {%s};" % ", ".join(arr(b["center"]) for b in m["boxes"]))
    o.append("BEZ_TBL double BEZ_BOX_HALF[BEZ_NBOX][3] = {%s};" % ", ".join(arr(b["half"]) for b in m["boxes"]))
    o.append("/* ground contact points (link-local), only on the torso and on chain-end links; the first 8 are the foot-box bottom corners (4 left, 4 right) */")
    o.append("BEZ_TBL int BEZ_PT_LINK[BEZ_NPT] = {%s};" % ", ".join(str(p["link"]) for p in m["ground_points"]))
    o.append("BEZ_TBL double BEZ_PT_POS[BEZ_NPT][3] = {%s};" % ", ".join(arr(p["p"]) for p in m["ground_points"]))
    o.append("/* body whose NET_CONTACT_FORCE row reports the point */")
    o.append("BEZ_TBL int BEZ_PT_BODY[BEZ_NPT] = {%s};" % ", ".join(str(p["body"]) for p in m["ground_points"]))
    if mc is not None:
        LC = mc["links"]
        assert len(mc["ground_points"]) == len(m["ground_points"]) and [p["link"] for p in mc["ground_points"]] == [p["link"] for p in m["ground_points"]]
        assert [l["parent"] for l in LC] == [l["parent"] for l in L] and [l["xyz"] for l in LC] == [l["xyz"] for l in L]
        o.append("/* ---- cleats variant (asset.cleats: True -> soccerbot_stl_sensor.urdf, kick_env.py:267-276): 8 cleat bodies fixed to the\n"
                 " * feet (29 robot bodies; contact rows 13:17 / 25:29, kick_env.py:187-191); same tree, the cleats' mass is merged into\n"
                 " * the feet, and the foot ground points are the cleats' bottom centres. */")
        o.append("#define BEZ_NB_CL %d\n#define BEZ_NBE_CL %d\n#define BEZ_NBE_MAX %d" % (mc["num_bodies"], mc["num_bodies"] + 1, mc["num_bodies"] + 1))
        o.append("#define BEZ_LFOOT_BODY_CL 12\n#define BEZ_RFOOT_BODY_CL 24\n#define BEZ_LCLEAT_BODY_CL 13\n#define BEZ_RCLEAT_BODY_CL 25")
        o.append("BEZ_TBL int BEZ_LINK_BODY_CL[BEZ_NL] = {%s};" % ", ".join(str(l["body"]) for l in LC))
        o.append("BEZ_TBL double BEZ_LINK_MASS_CL[BEZ_NL] = %s;" % arr(l["mass"] for l in LC))
        o.append("BEZ_TBL double BEZ_LINK_COM_CL[BEZ_NL][3] = {%s};" % ", ".join(arr(l["com"]) for l in LC))
        o.append("BEZ_TBL double BEZ_LINK_INERTIA_CL[BEZ_NL][6] = {%s};" % ", ".join(arr(l["inertia"]) for l in LC))
        o.append("BEZ_TBL int BEZ_BODY_LINK_CL[BEZ_NB_CL] = {%s};" % ", ".join(str(x) for x in mc["body_link"]))
        o.append("BEZ_TBL double BEZ_BODY_OFFSET_CL[BEZ_NB_CL][3] = {%s};" % ", ".join(arr(x) for x in mc["body_offset"]))
        o.append("BEZ_TBL double BEZ_PT_POS_CL[BEZ_NPT][3] = {%s};" % ", ".join(arr(p["p"]) for p in mc["ground_points"]))
        o.append("BEZ_TBL int BEZ_PT_BODY_CL[BEZ_NPT] = {%s};" % ", ".join(str(p["body"]) for p in mc["ground_points"]))
    o.append("/* leg self-collision capsules (link-local segment p0-p1, radius) and the left x right pair list */")
    o.append("#define BEZ_NCAP %d\n#define BEZ_NCPAIR %d" % (len(m["capsules"]), len(m["capsule_pairs"])))
    o.append("BEZ_TBL int BEZ_CAP_LINK[BEZ_NCAP] = {%s};" % ", ".join(str(c["link"]) for c in m["capsules"]))
    o.append("BEZ_TBL double BEZ_CAP_P0[BEZ_NCAP][3] = {%s};" % ", ".join(arr(c["p0"]) for c in m["capsules"]))
    o.append("BEZ_TBL double BEZ_CAP_P1[BEZ_NCAP][3] = {%s};" % ", ".join(arr(c["p1"]) for c in m["capsules"]))
    o.append("BEZ_TBL double BEZ_CAP_R[BEZ_NCAP] = %s;" % arr(c["r"] for c in m["capsules"]))
    o.append("BEZ_TBL int BEZ_CPAIR[BEZ_NCPAIR][2] = {%s};" % ", ".join("{%d, %d}" % tuple(p) for p in m["capsule_pairs"]))
    o.append("#define BEZ_BALL_MASS %s\n#define BEZ_BALL_INERTIA %s\n#define BEZ_BALL_RADIUS %s" % (
        fmt(m["ball"]["mass"]), fmt(m["ball"]["inertia"]), fmt(m["ball"]["radius"])))
    c = m["cfg"]
    o.append("/* bez_kick.yaml / kick_env.py defaults (runtime-overridable through BezSimConfig) */")
    for k in ("dt", "kp", "kd", "armature", "effort", "vel_limit", "joint_friction", "plane_friction",
              "clip_actions", "episode_length_s"):
        o.append("#define BEZ_DEFAULT_%s %s" % (k.upper(), fmt(c[k])))
    o.append("#define BEZ_DEFAULT_SUBSTEPS %d" % c["substeps"])
    o.append("BEZ_TBL double BEZ_DEFAULT_GRAVITY[3] = %s;" % arr(c["gravity"]))
    o.append("BEZ_TBL double BEZ_DEFAULT_BEZ_INIT[7] = %s;" % arr(c["bez_init"]))
    o.append("BEZ_TBL double BEZ_DEFAULT_BALL_INIT[7] = %s;" % arr(c["ball_init"]))
    o.append("BEZ_TBL double BEZ_DEFAULT_GOAL[2] = %s;" % arr(c["goal"]))
    if mb is not None:
        # asset.stl: False (kick_env.py:266-276): soccerbot_box.urdf / soccerbot_box_sensor.urdf.  Same tree, inertias, joints and
        # leg boxes as the stl assets (asserted here); only the collision shapes of the torso / head / arms differ, i.e. the
        # upper-body guard points and the ball <-> torso box
        def same_dynamics(a, b, known=()):
            assert len(a["links"]) == len(b["links"])
            for la, lb in zip(a["links"], b["links"]):
                for k in ("name", "parent", "body"):
                    assert la[k] == lb[k], (k, la, lb)
                for k in ("axis", "xyz", "com", "inertia"):
                    assert (la["name"], k) in known or np.allclose(la[k], lb[k], rtol=0, atol=1e-12), (k, la["name"], la[k], lb[k])
                assert abs(la["mass"] - lb["mass"]) < 1e-15
            assert a["boxes"][:-1] == b["boxes"][:-1] and a["boxes"][-1]["link"] == b["boxes"][-1]["link"] == 0
            assert a["capsules"] == b["capsules"] and a["capsule_pairs"] == b["capsule_pairs"]
            assert [(q["link"], q["body"]) for q in a["ground_points"]] == [(q["link"], q["body"]) for q in b["ground_points"]]
            for qa, qb in zip(a["ground_points"], b["ground_points"]):
                if qa["kind"] != "guard":
                    assert qa["p"] == qb["p"]
            assert a["body_names"] == b["body_names"] and a["dof_lower"] == b["dof_lower"] and a["dof_upper"] == b["dof_upper"]
        same_dynamics(m, mb)
        # soccerbot_box_sensor.urdf (box + cleats) differs from soccerbot_stl_sensor.urdf in ONE dynamic constant: its right ankle
        # joint sits at z = -0.0827 instead of -0.0865 (urdf right_leg_motor_4).  Everything else matches (asserted).
        same_dynamics(mc, mbc, known=(("/right_ankle", "xyz"),))
        qi = [i for i, l in enumerate(mbc["links"]) if l["name"] == "/right_ankle"][0]
        ra, rs_ = mbc["links"][qi]["xyz"], mc["links"][qi]["xyz"]
        assert ra[0] == rs_[0] == 0.0 and ra[1] == rs_[1] == 0.0 and ra[2] != rs_[2], (ra, rs_)
        assert [q["p"] for q in mbc["ground_points"]][8:] == [q["p"] for q in mb["ground_points"]][8:]   # upper-body points: the box asset's
        assert [q["p"] for q in mbc["ground_points"]][:8] == [q["p"] for q in mc["ground_points"]][:8]   # cleat points: the cleats asset's
        o.append("/* ---- box assets (asset.stl: False -> soccerbot_box.urdf / soccerbot_box_sensor.urdf, kick_env.py:266-276; BEZ_FLAG_BOX_ASSET):\n"
                 " * dynamics, leg boxes, capsules and foot / cleat points are those of the stl assets; the upper-body guard points come from the\n"
                 " * URDF's own torso / head / forearm collision boxes and the ball <-> torso box is the URDF's torso box.  With cleats\n"
                 " * (soccerbot_box_sensor.urdf) one joint origin differs as well: link BEZ_BOXCL_LINK sits at z = BEZ_BOXCL_LINK_Z. */")
        o.append("#define BEZ_BOXCL_LINK %d\n#define BEZ_BOXCL_LINK_Z %s" % (qi, fmt(ra[2])))
        o.append("#define BEZ_TORSO_BOX %d" % (len(m["boxes"]) - 1))
        o.append("BEZ_TBL double BEZ_PT_POS_BOX[BEZ_NPT][3] = {%s};" % ", ".join(arr(q["p"]) for q in mb["ground_points"]))
        o.append("BEZ_TBL double BEZ_TORSO_BOX_CENTER_BOX[3] = %s;" % arr(mb["boxes"][-1]["center"]))
        o.append("BEZ_TBL double BEZ_TORSO_BOX_HALF_BOX[3] = %s;" % arr(mb["boxes"][-1]["half"]))
        assert mb["boxes"][-1] == mbc["boxes"][-1]
    o.append("#endif /* BEZ_MODEL_GEN_H */\n")
    return "\n".join(o)


def main():
    m = build()
    mc = build(os.path.join(REF, "resources/assets/bez/model/soccerbot_stl_sensor.urdf"), cleats=True)
    m["cleats"] = {k: mc[k] for k in ("num_bodies", "body_names", "body_link", "body_offset", "links", "ground_points", "total_mass")}
    mb = build(os.path.join(REF, "resources/assets/bez/model/soccerbot_box.urdf"), box=True)
    mbc = build(os.path.join(REF, "resources/assets/bez/model/soccerbot_box_sensor.urdf"), cleats=True, box=True)
    m["box_asset"] = {"ground_points": mb["ground_points"], "torso_box": mb["boxes"][-1],
                      "cleats_right_ankle_xyz": [l for l in mbc["links"] if l["name"] == "/right_ankle"][0]["xyz"]}
    with open(OUT_JSON, "w") as f:
        json.dump(m, f, indent=1)
    with open(OUT_H, "w") as f:
        f.write(emit_header(m, mc, mb, mbc))
    print("links:")
    for i, l in enumerate(m["links"]):
        print(i, l["name"], "parent", l["parent"], "axis", l["axis"], "xyz", l["xyz"], "m=%.6f" % l["mass"])
    print("total mass %.6f kg; %d boxes; %d ground points" % (m["total_mass"], len(m["boxes"]), len(m["ground_points"])))


if __name__ == "__main__":
    sys.exit(main())
