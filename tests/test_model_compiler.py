"""bez_isaacgym_amd/model/compile_model.py bakes the reference's URDFs / YAML into csrc/bez_model_gen.h + model/bez_model.json (the
tables the HIP kernels and the C oracle share).  Where the reference tree is present (the build container) the committed files must
be exactly what the compiler produces from it; everywhere, the committed tables must be self-consistent."""
import importlib.util
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "bez_isaacgym_amd", "csrc", "bez_model_gen.h")
JSN = os.path.join(ROOT, "bez_isaacgym_amd", "model", "bez_model.json")
REF = os.environ.get("BEZ_REFERENCE_ROOT", "/root/reference")


def _table(name, text):
    m = re.search(r"BEZ_TBL double %s(?:\[[A-Z_0-9]+\])+ = (\{.*?\});" % name, text, re.S)
    assert m, name
    return np.array(json.loads(m.group(1).replace("{", "[").replace("}", "]")))


def test_committed_tables_are_self_consistent():
    text = open(HDR).read()
    m = json.load(open(JSN))
    pts, pts_cl, pts_box = _table("BEZ_PT_POS", text), _table("BEZ_PT_POS_CL", text), _table("BEZ_PT_POS_BOX", text)
    assert pts.shape == pts_cl.shape == pts_box.shape == (22, 3)
    # the box asset keeps the foot points and moves upper-body points only; the cleats asset moves the foot points only
    np.testing.assert_array_equal(pts_box[:8], pts[:8])
    np.testing.assert_array_equal(pts_cl[8:], pts[8:])
    assert np.abs(pts_box[8:] - pts[8:]).max() > 1e-3
    # torso box of the box asset = its eight guard corners
    c, h = _table("BEZ_TORSO_BOX_CENTER_BOX", text), _table("BEZ_TORSO_BOX_HALF_BOX", text)
    corners = np.array([[c[0] + sx * h[0], c[1] + sy * h[1], c[2] + sz * h[2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
    np.testing.assert_allclose(np.sort(pts_box[8:16], axis=0), np.sort(corners, axis=0), atol=1e-12)
    # the one joint origin soccerbot_box_sensor.urdf moves
    link = int(re.search(r"#define BEZ_BOXCL_LINK (\d+)", text).group(1))
    z = float(re.search(r"#define BEZ_BOXCL_LINK_Z (\S+)", text).group(1))
    xyz = _table("BEZ_LINK_XYZ", text)
    assert m["links"][link]["name"] == "/right_ankle" and xyz[link][0] == xyz[link][1] == 0.0 and abs(z - xyz[link][2]) > 1e-3
    assert m["box_asset"]["cleats_right_ankle_xyz"] == [0.0, 0.0, z]
    assert abs(m["total_mass"] - 2.827994) < 1e-6 and abs(m["cleats"]["total_mass"] - 2.867994) < 1e-6


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "resources", "assets", "bez", "model")), reason="needs the reference tree (build container only)")
def test_committed_model_is_what_the_compiler_produces(tmp_path):
    spec = importlib.util.spec_from_file_location("compile_model", os.path.join(ROOT, "bez_isaacgym_amd", "model", "compile_model.py"))
    cm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cm)
    cm.OUT_H, cm.OUT_JSON = str(tmp_path / "gen.h"), str(tmp_path / "model.json")
    cm.main()
    assert open(cm.OUT_H).read() == open(HDR).read()
    assert json.load(open(cm.OUT_JSON)) == json.load(open(JSN))
