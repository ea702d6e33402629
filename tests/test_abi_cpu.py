"""CPU: the C-ABI library loads without a GPU and exports every symbol include/bez_sim.h declares;
the ctypes mirror of BezSimConfig matches the C struct; the product refuses to run without a GPU."""
import ctypes as C
import numpy as np
import os
import re

import pytest
import torch

from bez_isaacgym_amd import abi
from bez_isaacgym_amd.build import build, lib_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build()
    from bez_isaacgym_amd.sim import load_library
    return load_library()


def test_exports_match_header(lib):
    hdr = open(os.path.join(ROOT, "include", "bez_sim.h")).read()
    declared = set(re.findall(r"\b(bez_sim_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    from bez_isaacgym_amd.sim import EXPORTS
    assert declared == set(EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    # the PPO glue kernels of the same header / library: every declared entry point is exported and bound (ppo/fused.py)
    ppo = set(re.findall(r"\b(bez_ppo_[a-z_0-9]+)\s*\(", hdr))
    assert len(ppo) >= 20
    from bez_isaacgym_amd.ppo.fused import _SIGS
    assert ppo == set(_SIGS) | {"bez_ppo_abi_version"}, ppo ^ set(_SIGS)
    for name in ppo:
        assert hasattr(lib, name), name
    from bez_isaacgym_amd.ppo.fused import PPO_ABI_VERSION
    lib.bez_ppo_abi_version.restype = C.c_int32
    assert lib.bez_ppo_abi_version() == PPO_ABI_VERSION == int(re.search(r"#define BEZ_PPO_ABI_VERSION (\d+)", hdr).group(1))


def test_default_config_matches_python(lib):
    c = abi.BezSimConfig()
    assert lib.bez_sim_default_config(C.byref(c), 4096) == 0
    assert bytes(c) == bytes(abi.default_config(4096))
    assert c.max_episode_length == 900 and c.substeps == 2 and c.num_envs == 4096


def test_create_argument_errors(lib):
    h = C.c_void_p()
    c = abi.default_config(8)
    c.abi_version = 99
    assert lib.bez_sim_create(C.byref(c), 0, C.byref(h)) < 0
    assert b"abi_version" in lib.bez_sim_last_error(None)
    c = abi.default_config(0)
    assert lib.bez_sim_create(C.byref(c), 0, C.byref(h)) < 0


def test_oracle_only_model_variants_are_refused(lib):
    """BEZ_FLAG_HARD_CONTACT, BEZ_FLAG_TGS_SOLVER and BezSimConfig.tune[] select solver experiments that exist only in the CPU
    oracle (include/bez_sim.h): the HIP library must say so (rc -5, before it even looks for a device) instead of stepping
    the compliant model under a flag it ignores.  bez_sim_set_flags applies the same rule (GPU: tests/test_gpu_round3.py)."""
    for mutate in (lambda c: setattr(c, "flags", c.flags | abi.FLAG_HARD_CONTACT),
                   lambda c: setattr(c, "flags", c.flags | abi.FLAG_TGS_SOLVER),
                   lambda c: setattr(c, "flags", c.flags | abi.FLAG_ANKLE_STOP | abi.FLAG_CLEATS),          # (round 6: both run on the lane kernel -- for the stl asset without cleats only)
                   lambda c: setattr(c, "flags", c.flags | abi.FLAG_ALL_GROUND_SHAPES | abi.FLAG_BOX_ASSET),
                   lambda c: c.tune.__setitem__(3, 0.5), lambda c: c.tune.__setitem__(23, 1.0)):
        h = C.c_void_p()
        c = abi.default_config(8)
        mutate(c)
        assert lib.bez_sim_create(C.byref(c), 0, C.byref(h)) == -5
        assert not h.value
        assert b"oracle" in lib.bez_sim_last_error(None).lower()
    hdr = open(os.path.join(ROOT, "include", "bez_sim.h")).read()
    assert int(re.search(r"#define BEZ_FLAG_HARD_CONTACT (\d+)u", hdr).group(1)) == abi.FLAG_HARD_CONTACT
    assert int(re.search(r"#define BEZ_FLAG_TGS_SOLVER (\d+)u", hdr).group(1)) == abi.FLAG_TGS_SOLVER
    assert int(re.search(r"#define BEZ_FLAG_ANKLE_STOP (\d+)u", hdr).group(1)) == abi.FLAG_ANKLE_STOP
    assert int(re.search(r"#define BEZ_FLAG_ALL_GROUND_SHAPES (\d+)u", hdr).group(1)) == abi.FLAG_ALL_GROUND_SHAPES


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_fails_loudly_without_gpu(lib):
    from bez_isaacgym_amd.sim import BezSim, BezSimError
    with pytest.raises(BezSimError):
        BezSim(abi.default_config(8), 0)
    h = C.c_void_p()
    c = abi.default_config(8)
    assert lib.bez_sim_create(C.byref(c), 0, C.byref(h)) < 0  # no device: error, not a CPU fallback
    assert b"no HIP device" in lib.bez_sim_last_error(None) or b"hip" in lib.bez_sim_last_error(None).lower()


def test_product_does_not_import_oracle():
    """The shipped package must never reference oracle/ (the oracle is the checker, not a fallback)."""
    pkg = os.path.join(ROOT, "bez_isaacgym_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "bez_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, os.path.join(dp, f)


@pytest.mark.parametrize("path,value", [(("env", "urdfAsset", "disable_gravity"), True),
                                        (("env", "urdfAsset", "angular_damping"), 0.05), (("env", "urdfAsset", "linear_damping"), 0.1),
                                        (("env", "plane", "restitution"), 0.3), (("env", "plane", "staticFriction"), 0.5),
                                        (("env", "controlFrequencyInv"), 0)])
def test_unmodelled_config_values_are_refused(path, value):
    """kick_env.py:250-256,283-294 forward these keys to Isaac Gym; the HIP step models only the reference yaml's values, so anything
    else raises in abi.config_from_task_cfg instead of being read and ignored (missing 5)."""
    from bez_isaacgym_amd.utils.config import load_config
    cfg = load_config(["task=bez_kick", "num_envs=8", "headless=True"])["task"]
    abi.config_from_task_cfg(cfg)           # the reference's own yaml passes
    node = cfg
    for k in path[:-1]:
        node = node[k]
    node[path[-1]] = value
    with pytest.raises(ValueError, match=path[-1]):
        abi.config_from_task_cfg(cfg)


def test_per_env_uniform_is_the_oracles_philox_and_shard_invariant():
    """The setup-only mass scale draw (bez_kick.yaml:175, vec_task.py: rigid_body_properties.mass without a schedule) uses
    utils.per_env_uniform: word k of Philox4x32-10 keyed by (seed, GLOBAL env id, tag) -- the oracle's generator word for word,
    and a shard [off, off + n) of the envs draws exactly the rows of the whole range."""
    from bez_isaacgym_amd.utils.utils import per_env_uniform
    from oracle.bez_oracle import Oracle
    o = Oracle(num_envs=2)
    seed, tag = (42 << 32) | 7, 0x4D415353
    whole = per_env_uniform(seed, range(0, 300), tag, 19)
    for g in (0, 5, 299):
        for k in (0, 3, 4, 18):
            w = o.philox_word(seed, g, tag, k)      # counter = (env lo, env hi, third word, k >> 2), word k & 3
            assert whole[g, k] == np.float32(w >> 8) * np.float32(1.0 / 16777216.0)
    assert np.array_equal(per_env_uniform(seed, range(100, 228), tag, 19), whole[100:228])
    assert 0.0 <= whole.min() and whole.max() < 1.0 and abs(whole.mean() - 0.5) < 0.02


def test_fix_base_link_translates_to_its_flag():
    """urdfAsset.fixBaseLink (kick_env.py:287, bez_kick.yaml:83) is modelled: BEZ_FLAG_FIX_BASE, the header's value."""
    from bez_isaacgym_amd.utils.config import load_config
    cfg = load_config(["task=bez_kick", "num_envs=8", "headless=True"])["task"]
    assert not abi.config_from_task_cfg(cfg).flags & abi.FLAG_FIX_BASE
    cfg["env"]["urdfAsset"]["fixBaseLink"] = True
    assert abi.config_from_task_cfg(cfg).flags & abi.FLAG_FIX_BASE
    hdr = open(os.path.join(ROOT, "include", "bez_sim.h")).read()
    assert int(re.search(r"#define BEZ_FLAG_FIX_BASE (\d+)u", hdr).group(1)) == abi.FLAG_FIX_BASE
