"""Round-4 GPU tests (through the C ABI).  All tests need a GPU: `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_set_flags_refuses_oracle_only_variants():
    """include/bez_sim.h: BEZ_FLAG_HARD_CONTACT / BEZ_FLAG_TGS_SOLVER exist only in the CPU oracle.  A live simulator must
    refuse them in bez_sim_set_flags (rc -5, message) and keep stepping with its previous flag word."""
    import torch
    from bez_isaacgym_amd import abi
    from bez_isaacgym_amd.sim import BezSim, BezSimError
    sim = BezSim(abi.default_config(64, seed=2), 0)
    before = int(sim.cfg.flags)
    for bad in (abi.FLAG_HARD_CONTACT, abi.FLAG_TGS_SOLVER):
        with pytest.raises(BezSimError, match="oracle"):
            sim.set_flags(before | bad)
    sim.set_flags(before)  # a legal word is still accepted
    sim.step(torch.zeros(64 * 18, device=sim.device))
    torch.cuda.synchronize()
    assert np.isfinite(sim.tensor(abi.TENSOR_OBS).cpu().numpy()).all()
    c = abi.default_config(64)
    c.tune[9] = 1.0
    with pytest.raises(BezSimError, match="oracle"):
        BezSim(c, 0)
