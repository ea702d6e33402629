"""CPU: the no-unpickle reader of rl_games checkpoints (bez_isaacgym_amd/utils/rlg_checkpoint.py).

Always: a synthetic torch.save file round-trips through the reader and unexpected globals / opcodes are refused.
Where the reference tree is present (build container only): the shipped Bez_Kick_33.pth has the layout SURVEY.md section 6
reports and loads into this build's PPO modules under the same key names (checkpoint interop)."""
import os
import pickle

import numpy as np
import pytest
import torch

from bez_isaacgym_amd.ppo.a2c_continuous import ModelA2CContinuousLogStd, RunningMeanStd
from bez_isaacgym_amd.utils.rlg_checkpoint import _symbolic_eval, load_into_agent_modules, read_rlgames_checkpoint

REF_CK = "/root/reference/bez_isaacgym/results/Bez_Kick/Normal/Bez_Kick_33.pth"


def test_roundtrip_of_own_checkpoint(tmp_path):
    torch.manual_seed(0)
    m = ModelA2CContinuousLogStd(54, 18, (400, 200, 100))
    rms = RunningMeanStd((54,))
    rms.running_mean += torch.randn(54, dtype=torch.float64)
    path = str(tmp_path / "ck.pth")
    torch.save({"model": m.state_dict(), "running_mean_std": rms.state_dict(), "epoch": 7, "frame": 12345,
                "strided": torch.arange(12.0).reshape(3, 4).t()}, path)
    ck = read_rlgames_checkpoint(path)
    for k, v in m.state_dict().items():
        np.testing.assert_array_equal(ck["model"][k], v.numpy())
    np.testing.assert_array_equal(ck["running_mean_std"]["running_mean"], rms.running_mean.numpy())
    assert ck["epoch"] == 7 and ck["frame"] == 12345
    np.testing.assert_array_equal(ck["strided"], np.arange(12.0).reshape(3, 4).T)  # non-contiguous strides honoured
    m2 = ModelA2CContinuousLogStd(54, 18, (400, 200, 100))
    load_into_agent_modules(ck, m2)
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a, b)


def test_refuses_anything_executable():
    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned",))
    with pytest.raises(ValueError):
        _symbolic_eval(pickle.dumps({"x": Evil()}, protocol=2))


@pytest.mark.skipif(not os.path.exists(REF_CK), reason="reference checkpoint only exists in the build container")
def test_reference_checkpoint_layout_and_interop():
    ck = read_rlgames_checkpoint(REF_CK)
    assert {"model", "running_mean_std", "reward_mean_std", "optimizer", "epoch", "frame", "last_mean_rewards"} <= set(ck)
    assert sum(v.size for v in ck["model"].values() if isinstance(v, np.ndarray)) == 124237
    assert ck["epoch"] == 6156 and ck["frame"] == 806879232 and abs(ck["last_mean_rewards"] - 87.55) < 0.01
    m = ModelA2CContinuousLogStd(54, 18, (400, 200, 100)); rms = RunningMeanStd((54,)); vms = RunningMeanStd((1,))
    load_into_agent_modules(ck, m, rms, vms)
    assert torch.isfinite(m.a2c_network(torch.zeros(2, 54))[0]).all()
    # the reference's own observation statistics show the quirks this build reproduces:
    assert abs(float(rms.running_mean[38]) - 1.0) < 0.05      # imu z-acc == R(q)(0,0,1): prev_lin_vel aliasing (Q1)
    assert float(rms.running_var[52]) < 1e-6 and float(rms.running_var[53]) < 1e-6  # constant ball_init tail (Q5)


def _tamper(path, out, fn):
    """Rewrite the tensor geometry inside data.pkl of a torch.save archive (the zip payloads stay as they are)."""
    import pickletools
    import zipfile
    with zipfile.ZipFile(path) as zf:
        names = zf.namelist()
        blobs = {n: zf.read(n) for n in names}
    pkl = [n for n in names if n.endswith("/data.pkl")][0]
    blobs[pkl] = fn(blobs[pkl])
    with zipfile.ZipFile(out, "w") as zf:
        for n in names:
            zf.writestr(n, blobs[n])


def test_refuses_out_of_bounds_tensor_geometry(tmp_path):
    """offset / size / stride come from the untrusted pickle: a view that leaves its storage must be refused, not read."""
    src = str(tmp_path / "ok.pth")
    torch.save({"w": torch.arange(6.0)}, src)
    assert read_rlgames_checkpoint(src)["w"].shape == (6,)
    # BININT1 6 (the size tuple's only entry) -> 200: reaches far past the 6-element storage
    big = str(tmp_path / "big.pth")
    _tamper(src, big, lambda b: b.replace(b"K\x06\x85", b"K\xc8\x85", 1))
    with pytest.raises(ValueError):
        read_rlgames_checkpoint(big)
    # stride 1 -> BININT -1
    neg = str(tmp_path / "neg.pth")
    _tamper(src, neg, lambda b: b.replace(b"K\x06\x85q\x05K\x01\x85", b"K\x06\x85q\x05J\xff\xff\xff\xff\x85", 1)
            if b"K\x06\x85q\x05K\x01\x85" in b else b.replace(b"K\x01\x85", b"J\xff\xff\xff\xff\x85", 1))
    with pytest.raises(ValueError):
        read_rlgames_checkpoint(neg)
