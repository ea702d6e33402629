"""Round-5 GPU tests (through the C ABI).  All tests need a GPU: `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _getup(backend, name, n, flags=None):
    import json, os
    from bez_isaacgym_amd import abi
    from tests.scenarios import ROOT, lay_down, make_backend, play
    model = json.load(open(os.path.join(ROOT, "bez_isaacgym_amd", "model", "bez_model.json")))
    cfg = abi.default_config(n, seed=7)
    if flags is not None:
        cfg.flags = flags
    sim = make_backend(backend, cfg)
    sim.step(np.zeros((n, 18), np.float32))
    lay_down(sim, n, name, np.random.default_rng(3))
    return play(sim, n, name, model)


@pytest.mark.parametrize("shapes", [False, True])
@pytest.mark.parametrize("name", ["getupfront", "getupback", "getupside"])
def test_getup_scenarios_hip_equals_oracle(name, shapes):
    """The reference's get-up tables from lying starts (tests/scenarios.py) in 64 HIP envs through the split entry points
    (bez_sim_pre_physics + bez_sim_simulate): a thousand control steps on knees, forearms and the torso's guard points -- contact
    the bez_kick episodes never reach before their fall reset.  The HIP kernels and the fp64 oracle must tell the same story
    (medians over the envs; the envs differ by +-0.02 rad in their start pose)."""
    from bez_isaacgym_amd import abi
    # shapes: BEZ_FLAG_ALL_GROUND_SHAPES (round 6: on the HIP back-end too, one-env-per-lane kernel) -- ground contact at the corners of
    # every collision shape of soccerbot_stl.urdf:172-278, which is what lets the back get-up roll over (soccer_trajectories.py:56-91)
    fl = (abi.FLAG_IMU_PREV_ALIAS | abi.FLAG_ALL_GROUND_SHAPES) if shapes else None
    h = _getup("hip", name, 64, fl)
    o = _getup("oracle", name, 8, fl)
    assert h["finite"] == 1.0, h
    for k, tol in (("final_z", 0.02), ("max_z", 0.02), ("final_up", 0.15), ("max_up", 0.1)):
        assert abs(h[k] - o[k]) < tol, (k, h, o)
    if name == "getupfront" or (shapes and name == "getupback"):
        assert h["max_z"] > 0.20 and h["max_up"] > 0.75, h   # the squat on the feet is reached (from the back: only with every shape on the ground)
    if name == "getupback" and not shapes:
        assert h["max_z"] < 0.12, h                           # with the baked contact set the robot never leaves its back


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


def _kick_env(n, overrides=(), seed=42, env_id_offset=0, mutate=None):
    from bez_isaacgym_amd.tasks import isaacgym_task_map
    from bez_isaacgym_amd.utils.config import load_config
    cfg = load_config(["task=bez_kick", "num_envs=%d" % n, "headless=True"] + list(overrides))["task"]
    cfg["rl_device"] = "cuda:0"
    cfg["seed"] = seed
    cfg["env_id_offset"] = env_id_offset
    if mutate:
        mutate(cfg)
    return isaacgym_task_map["bez_kick"](cfg=cfg, sim_device="cuda:0", graphics_device_id=0, headless=True)


def test_control_frequency_inv_loops_the_simulate_call():
    """vec_task.py:322-324: `for i in range(self.control_freq_inv): self.gym.simulate(self.sim)` (the key is env.controlFrequencyInv,
    vec_task.py:90).  KickEnv.step with controlFrequencyInv = 2 must equal pre_physics + two simulate calls + post_physics driven by
    hand on a twin, and differ from the one-simulate fused step."""
    import torch
    n = 128
    two = _kick_env(n, mutate=lambda c: c["env"].__setitem__("controlFrequencyInv", 2))
    twin = _kick_env(n)
    one = _kick_env(n)
    assert two.control_freq_inv == 2 and one.control_freq_inv == 1
    g = torch.Generator(device="cpu").manual_seed(3)
    for t in range(12):
        a = (torch.rand(n, 18, generator=g) * 2 - 1).to("cuda:0")
        o2, r2, d2, _ = two.step(a)
        twin.pre_physics_step(torch.clamp(a, -twin.clip_actions, twin.clip_actions))
        twin.sim.simulate(); twin.sim.simulate()
        twin.post_physics_step()
        one.step(a)
        assert torch.equal(o2["obs"], twin.obs_buf) and torch.equal(r2, twin.rew_buf) and torch.equal(d2, twin.reset_buf)
    assert not torch.equal(two.obs_buf, one.obs_buf)
    assert torch.equal(two.progress_buf, one.progress_buf) or (two.reset_buf != one.reset_buf).any()  # progress counts env steps, not simulate calls


def test_setup_only_mass_randomisation_is_shard_invariant():
    """rigid_body_properties.mass without a linear schedule is a real one-time draw (vec_task.py:505-725, `setup_only`): keyed by the
    GLOBAL env id like every other per-env draw, so two shards of 64 envs hold the rows of one env of 128 (round 4 drew it from a torch
    generator seeded with the shard offset)."""
    import torch
    from bez_isaacgym_amd import abi

    def no_schedule(c):
        m = c["task"]["randomization_params"]["actor_params"]["bez"]["rigid_body_properties"]["mass"]
        m.pop("schedule", None); m.pop("schedule_steps", None)
    whole = _kick_env(128, ["task.task.randomize=True"], mutate=no_schedule)
    parts = [_kick_env(64, ["task.task.randomize=True"], env_id_offset=64 * k, mutate=no_schedule) for k in range(2)]
    mw = whole.sim.get_env_params(abi.PARAM_MASS_SCALE)
    mp = torch.cat([p.sim.get_env_params(abi.PARAM_MASS_SCALE) for p in parts])
    assert torch.equal(mw, mp)
    lo, hi = whole.cfg["task"]["randomization_params"]["actor_params"]["bez"]["rigid_body_properties"]["mass"]["range"]
    assert float(mw.min()) >= lo and float(mw.max()) <= hi and float(mw.std()) > 0.05 * (hi - lo)


def test_agent_runs_eager_when_graph_replay_is_not_safe():
    """Round-4 advisor finding (medium): `torch.cuda.init()` before `import bez_isaacgym_amd` leaves the HIP runtime's graph packet
    capture on -- replayed kernels then run with clobbered arguments once enough eager launches happen in between.
    The agent must not replay graphs in that state (hip_graphs: auto -> eager, with a warning), on the fused path too."""
    import os, subprocess, sys
    code = r'''
import os, sys, warnings
os.environ.pop("DEBUG_CLR_GRAPH_PACKET_CAPTURE", None)
import torch
torch.cuda.init(); torch.zeros(1, device="cuda:0")
sys.path.insert(0, %r)
import bez_isaacgym_amd
assert not bez_isaacgym_amd.GRAPH_REPLAY_SAFE
from tests.test_gpu_round2 import _agent
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    a = _agent(256, 2048)
assert a.fused and not a.use_graphs, (a.fused, a.use_graphs)
assert any("HIP graphs are OFF" in str(x.message) for x in w), [str(x.message)[:60] for x in w]
a.obs = a.env_reset()
for _ in range(4):
    s = a.train_epoch()
assert a._g_rollout is None and a._g_update is None
import numpy as np
assert np.isfinite([s["kl"], s["a_loss"], s["c_loss"]]).all()
print("EAGER-OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "EAGER-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_external_weight_writes_reach_the_mfma_kernels():
    """Round-4 advisor finding: with the fused optimiser the fp16 working copy and the fragment-major copies are refreshed only when the
    model's load_state_dict hook fires.  An in-place write to the fp32 master weights between epochs (p.mul_, an EMA tool, a late
    broadcast) must reach the rollout's MFMA forward too: play_steps() notices it from the parameters' version counters (host only)."""
    import torch
    from tests.test_gpu_round2 import _agent
    a = _agent(256, 2048, hip_graphs=False)
    a.obs = a.env_reset()
    a.train_epoch()
    net = a.model.a2c_network
    half_before = a._hflat.clone() if getattr(a, "_hflat", None) is not None else None
    assert half_before is not None and a._copies_kept_current()
    with torch.no_grad():
        for p in a.model.parameters():
            p.mul_(0.5)
    a.play_steps()
    assert not torch.equal(a._hflat, half_before)
    for p16, p32 in zip(net._p16, net._p32):          # the fp16 working copy is the rounded master copy again
        assert torch.equal(p16, p32.detach().half())


def test_dr_hand_out_can_be_taken_back():
    """bez_sim_dr_step_args / bez_sim_dr_prelaunch mark the coming step's randomisation as "done by the caller".  If the caller's launch
    never runs, bez_sim_dr_cancel (also implied by set_randomization / seed) gives it back: the step after a cancelled hand-out equals the
    step of a twin that never handed anything out, and a second hand-out is accepted again (round-4 advisor finding)."""
    import torch
    from bez_isaacgym_amd import abi
    from bez_isaacgym_amd.sim import BezSim, BezSimError
    from tests.test_gpu_round3 import _dr_cfg
    n = 256
    a, b = BezSim(abi.default_config(n, seed=9), 0), BezSim(abi.default_config(n, seed=9), 0)
    for s in (a, b):
        s.set_randomization(_dr_cfg(freq=1, sched=0))
    act = torch.rand(n * 18, device=a.device) * 2 - 1
    for t in range(6):
        blob = a.dr_step_args()
        assert blob is not None
        with pytest.raises(BezSimError, match="already"):
            a.dr_step_args()
        a.dr_cancel()                      # ... the launch that should have carried `blob` failed
        a.step(act); b.step(act)
        torch.cuda.synchronize()
        for p in (abi.PARAM_FRICTION, abi.PARAM_KP_SCALE, abi.PARAM_DOF_LOWER):
            assert torch.equal(a.get_env_params(p), b.get_env_params(p)), (t, p)
        assert torch.equal(a.tensor(abi.TENSOR_OBS), b.tensor(abi.TENSOR_OBS))
        assert torch.equal(a.tensor(abi.TENSOR_RANDOMIZE_BUF), b.tensor(abi.TENSOR_RANDOMIZE_BUF))
    a.dr_step_args(); a.seed(9)            # seeding voids a hand-out too
    assert a.dr_step_args() is not None


def test_staged_dataset_prep_equals_the_one_call_form():
    """bez_ppo_dataset_prep_staged: the data-parallel loop runs the SAME launches with its two collectives between the stages (stage 1 |
    all-reduce of the moments | stage 2 | all-reduce of the advantage sums | stage 4).  On one rank -- the all-reduces are the identity --
    the staged form must reproduce the one-call form bit for bit, normaliser statistics included; and with the moment / advantage-sum
    buffers DOUBLED between the stages (what a second rank holding the same rows would contribute) the normalisations must be those of the
    doubled batch: same mean, the unbiased variance of 2n samples."""
    import torch
    from bez_isaacgym_amd.ppo import fused as F
    from bez_isaacgym_amd.ppo.a2c_continuous import RunningMeanStd
    DEV = "cuda:0"
    torch.manual_seed(4)
    h, n, nmb, d = 32, 512, 4, 54
    values, returns = torch.randn(h, n, 1, device=DEV) * 2 + 0.5, torch.randn(h, n, 1, device=DEV) * 3 - 1.0
    rows = h * n // nmb
    obs = torch.randn(nmb * rows, d, device=DEV) * 1.7 + 0.3

    def run(staged, double=False):
        rms = RunningMeanStd((1,)).to(DEV)
        rms.running_mean.fill_(0.2); rms.running_var.fill_(1.5); rms.count.fill_(1000.0)
        mom = torch.zeros(nmb * (2 * d + 1) + 6, device=DEV, dtype=torch.float64)      # one buffer, as the agent's _mom_pack
        om, vm, rm = mom[:nmb * (2 * d + 1)], mom[-6:-3], mom[-3:]
        ov, rt, adv = torch.zeros(h * n, 1, device=DEV), torch.zeros(h * n, 1, device=DEV), torch.zeros(h * n, device=DEV)
        sc = F.dataset_prep_scratch(nmb, h, n, DEV)
        sums = torch.zeros(6, device=DEV, dtype=torch.float64)
        call = lambda st, s=None: F.dataset_prep(obs, rows, nmb, om, values, returns, rms, vm, rm, ov, rt, adv, True, sc, stages=st, adv_sums=s)
        if not staged:
            assert call(7)
        else:
            assert call(1)
            if double:
                mom.mul_(2.0)
            assert call(2, sums)
            if double:
                sums[:3].mul_(2.0)
            assert call(4, sums)
        torch.cuda.synchronize()
        return dict(ov=ov, rt=rt, adv=adv, mom=mom.clone(), mean=rms.running_mean.clone(), var=rms.running_var.clone(), count=rms.count.clone(), sums=sums)
    one, st = run(False), run(True)
    for k in ("ov", "rt", "adv", "mom", "mean", "var", "count"):
        assert torch.equal(one[k], st[k]), k
    a = (one["rt"] - one["ov"]).reshape(-1).double()   # un-normalised advantages of the one-call run: recompute what stage 2 summed
    dbl = run(True, double=True)
    assert float(dbl["count"]) == 1000.0 + 4 * h * n   # two updates (values, returns) of 2 x h x n samples each
    a2 = (dbl["rt"] - dbl["ov"]).reshape(-1).double()
    m2, n2 = a2.mean(), 2.0 * a2.numel()
    std2 = torch.sqrt(((a2 * a2).sum() * 2 - n2 * m2 * m2) / (n2 - 1))
    torch.testing.assert_close(dbl["adv"].double(), (a2 - m2) / (std2 + 1e-8), rtol=2e-5, atol=2e-5)
    assert not torch.equal(dbl["adv"], one["adv"]) and a.numel() == a2.numel()


def test_norm_shares_after_an_all_reduce_and_the_folded_rank_division():
    """Data-parallel optimiser step: bez_ppo_grad_norm_parts re-forms the per-workgroup (sum g^2, non-finite count) shares from the
    all-reduced buffer, and BezPpoAdamExtra.grad_div folds the division by the number of ranks into the unscale factor.  (a) with the shares
    the step equals the step in which every workgroup reads the whole gradient (clip coefficient to rounding: a different fixed order);
    (b) the SUM of two identical rank gradients with grad_div = 2 equals the plain step on one of them; (c) a non-finite element anywhere
    -- the n % 4 tail included -- skips the step and halves the loss scale."""
    import torch
    from bez_isaacgym_amd.ppo import fused as F
    DEV = "cuda:0"
    torch.manual_seed(8)
    n = 124237                                        # bez_kickPPO.yaml's parameter count: n % 4 = 1
    g = torch.randn(n + 3, device=DEV)[:n] * 3.0      # (a slice: 16-byte aligned start, odd length)
    p0 = torch.randn(n, device=DEV)

    grid_buf = torch.zeros(F.ADAM_GRIDNORM_FLOATS, device=DEV)

    def step(grad, parts=None, div=1.0, scale0=1024.0, grid=False):
        p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        steps, lr = torch.zeros(1, device=DEV), torch.tensor([3e-4], device=DEV)
        scale, gt = torch.tensor([scale0], device=DEV), torch.zeros(1, device=DEV, dtype=torch.int32)
        work = torch.zeros(F.ADAM_WORK_FLOATS, device=DEV)
        buf = torch.zeros(64, 2, device=DEV)
        np_ = F.grad_norm_parts(grad, buf) if parts else None
        F.adam_step(p, grad, m, v, steps, lr, (0.9, 0.999), 1e-8, 0.0, 1.0, scale, gt, 2.0, 0.5, 2000, work, norm_parts=np_, grad_div=div,
                    grid_norm=grid_buf if grid else None)
        torch.cuda.synchronize()
        assert float(work.abs().max()) == 0.0
        if grid:   # the arrival counter is back at zero: the buffer serves the next launch as it is
            assert int(grid_buf[512:513].view(torch.int32)) == 0
        return p, m, float(scale), (np_.clone() if np_ is not None else None)
    gs = (g * 1024.0).contiguous()                    # "still scaled" gradient
    p_ref, m_ref, sc_ref, _ = step(gs)
    p_a, m_a, sc_a, parts = step(gs, parts=True)
    assert parts.shape[0] == (n // 4 + n % 4 + 1023) // 1024 and sc_a == sc_ref == 1024.0
    torch.testing.assert_close(parts[:, 0].double().sum(), (gs.double() ** 2).sum(), rtol=1e-5, atol=0)
    torch.testing.assert_close(p_a, p_ref, rtol=0, atol=2e-9); torch.testing.assert_close(m_a, m_ref, rtol=1e-6, atol=1e-9)
    p_b, m_b, _, _ = step((gs * 2.0).contiguous(), parts=True, div=2.0)
    torch.testing.assert_close(p_b, p_ref, rtol=0, atol=2e-9); torch.testing.assert_close(m_b, m_ref, rtol=1e-6, atol=1e-9)
    for bad_at in (5, n - 1):
        gb = gs.clone(); gb[bad_at] = float("inf")
        p_c, _, sc_c, parts_c = step(gb, parts=True)
        assert torch.equal(p_c, p0) and sc_c == 512.0 and float(parts_c[:, 1].sum()) == 1.0
    # (d) the norm formed inside the optimiser launch (BezPpoAdamExtra.grid_norm_dev: own-slice sums, the workgroups meet at a counter): the
    # same step again, three launches in a row on the same buffer, with the rank division, and a non-finite element skips the step
    for _ in range(3):
        p_d, m_d, sc_d, _ = step(gs, grid=True)
        torch.testing.assert_close(p_d, p_ref, rtol=0, atol=2e-9); torch.testing.assert_close(m_d, m_ref, rtol=1e-6, atol=1e-9)
    first = step(gs, grid=True)[0]
    assert torch.equal(first, p_d)                      # fixed order: bit-identical from launch to launch
    p_e, m_e, _, _ = step((gs * 2.0).contiguous(), div=2.0, grid=True)
    torch.testing.assert_close(p_e, p_ref, rtol=0, atol=2e-9)
    for bad_at in (5, n - 1):
        gb = gs.clone(); gb[bad_at] = float("nan")
        p_f, _, sc_f, _ = step(gb, grid=True)
        assert torch.equal(p_f, p0) and sc_f == 512.0


def test_captured_collectives_equal_the_segmented_update_on_one_rank():
    """`dp_capture_collectives: True` (opt-in, experimental): the RCCL calls are captured into the same graphs as everything else, so the
    data-parallel epoch is the single-GPU path's two replays again (bench: 4.17 against 4.52 ms segmented, 4.16 single-GPU).  On a 1-rank RCCL
    group -- all a 1-GPU box can hold -- it must train to the SAME bits as the segmented update: same kernels, same order, only the launch
    mechanism differs."""
    import os, subprocess, sys
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import bez_isaacgym_amd   # before anything initialises HIP: graph replay is only safe with the runtime's packet capture off (DESIGN.md 6.2)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="%d", RANK="0", WORLD_SIZE="1", BEZ_PPO_FORCE_DIST="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from tests.test_gpu_round2 import _agent
snaps = []
for capture in (False, True):
    torch.manual_seed(77); torch.cuda.manual_seed_all(77)
    a = _agent(512, 4096, dp_capture_collectives=capture)
    assert a._segmented == (not capture)
    a.obs = a.env_reset()
    st = [a.train_epoch() for _ in range(6)]
    torch.cuda.synchronize()
    assert (a._seg is None) == capture and (a._g_update is not None) == capture and a._g_rollout is not None, (capture, a.use_graphs, a._seg is None, a._g_update is None, a._g_rollout is None)
    assert all(np.isfinite([s["kl"], s["a_loss"], s["c_loss"]]).all() for s in st), st
    snaps.append([p.detach().clone() for p in a.model.parameters()] + [a._mflat.clone(), a.scaler._scale.clone(), a.running_mean_std.running_mean.clone()])
    a.release_env(); del a
assert all(torch.equal(x, y) for x, y in zip(*snaps)), [float((x.float() - y.float()).abs().max()) for x, y in zip(*snaps)]
dist.destroy_process_group()
print("CAPTURE_OK")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), __import__("tests.conftest", fromlist=["free_port"]).free_port())
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "CAPTURE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_fixed_base_hip_equals_oracle_and_the_torso_does_not_move(model):
    """BEZ_FLAG_FIX_BASE (urdfAsset.fixBaseLink, kick_env.py:287): stepped by the one-env-per-lane kernel.  (a) 128 envs x 40 random-action control
    steps through the split entry points against the fp64 oracle on the same states: the torso rows stay bit-equal to their start, the joints
    agree within the usual bars; (b) the reference's form of the per-DOF sweep (gravity on, torso welded a metre up) with the oracle's bars."""
    from bez_isaacgym_amd import abi
    from tests.scenarios import dof_sweep, make_backend
    from tests.test_scenarios import check_fixed_base_sweep
    n = 128
    mk = lambda backend: make_backend(backend, (lambda c: (setattr(c, "flags", c.flags | abi.FLAG_FIX_BASE), c)[1])(abi.default_config(n, seed=5)))
    h, o = mk("hip"), mk("oracle")
    for s_ in (h, o):
        s_.step(np.zeros((n, 18), np.float32))
    root0 = h.root_states.reshape(n, -1, 13)[:, 0].copy()
    rng = np.random.default_rng(2)
    for t in range(40):
        a = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        for s_ in (h, o):
            s_.pre_physics(a); s_.simulate()
        dh, do = h.dof_state.reshape(n, 18, 2), o.dof_state.reshape(n, 18, 2)
        assert np.abs(dh[..., 0] - do[..., 0]).max() < 1.5e-4 * (t + 1) and np.abs(dh[..., 1] - do[..., 1]).max() < 1.5e-2 * (t + 1), t   # not resynchronised: the bars grow with the steps
        o.set_dof_state(dh.reshape(-1, 2)); o.set_root_states(h.root_states)   # resynchronise (as the parity tests do)
    assert np.array_equal(h.root_states.reshape(n, -1, 13)[:, 0], root0)
    n2 = 64
    cfg = abi.default_config(n2, seed=3)
    cfg.flags |= abi.FLAG_FIX_BASE
    sim = make_backend("hip", cfg)
    sim.step(np.zeros((n2, 18), np.float32))
    rows = dof_sweep(sim, n2, model)
    expect = np.zeros(13, np.float32); expect[2] = 1.0; expect[6] = 1.0
    check_fixed_base_sweep(rows, np.tile(expect, (n2, 1)), sim.root_states.reshape(n2, -1, 13)[:, 0])


@pytest.mark.gpu
def test_pipelined_epochs_train_to_the_same_bits_and_report_the_same_numbers():
    """A2CAgent.train() reads epoch k's report while epoch k + 1 is queued (train_epoch_launch / train_epoch_finish): the same kernels in the
    same stream order as train_epoch() one epoch at a time -- same weights, Adam moments and normaliser bit for bit through eager, capturing and
    replayed epochs, and the same per-epoch report (KL, losses, learning rate, finished-episode sums cleared in stream order, not a report late)."""
    import torch
    from tests.test_gpu_round2 import _agent
    n_ep = 7
    reports = {}
    states = {}
    for mode in ("one_at_a_time", "pipelined"):
        torch.manual_seed(5)
        ag = _agent(512, 4096)
        ag.obs = ag.env_reset()
        reps, hist = [], []
        if mode == "one_at_a_time":
            for _ in range(n_ep):
                reps.append(ag.train_epoch())
                hist.append(list(ag._ep_hist[-1]) if ag._ep_hist else None)
        else:
            pending = None
            for _ in range(n_ep):
                t = ag.train_epoch_launch()
                if t is None:      # the first epoch (the rollout buffers do not exist yet): one at a time, as A2CAgent.train() runs it
                    assert pending is None and not reps
                    reps.append(dict(ag.train_epoch(), epoch=ag.epoch_num, frame=ag.frame)); hist.append(list(ag._ep_hist[-1]) if ag._ep_hist else None)
                    continue
                if pending is not None:
                    reps.append(ag.train_epoch_finish(pending)); hist.append(list(ag._ep_hist[-1]) if ag._ep_hist else None)
                pending = t
            reps.append(ag.train_epoch_finish(pending)); hist.append(list(ag._ep_hist[-1]) if ag._ep_hist else None)
            assert [r["epoch"] for r in reps] == list(range(1, n_ep + 1)) and reps[-1]["frame"] == ag.frame
        assert ag.use_graphs and ag._g_update is not None      # the later epochs were replays
        torch.cuda.synchronize()
        reports[mode] = [(r["kl"], r["a_loss"], r["c_loss"], r["lr"]) for r in reps], hist
        states[mode] = ([p.detach().clone() for p in ag.model.parameters()], ag._mflat.clone(), ag._vflat.clone(),
                        ag.running_mean_std.running_mean.clone(), float(ag.ep_stats.sum()))
        ag.release_env()
        del ag
    assert reports["one_at_a_time"][0] == reports["pipelined"][0]
    for a, b in zip(reports["one_at_a_time"][1], reports["pipelined"][1]):     # finished-episode sums: fp64 atomics, order-dependent in the last bits
        assert (a is None) == (b is None)
        if a is not None:
            np.testing.assert_allclose(a, b, rtol=1e-12)
    sa, sb = states["one_at_a_time"], states["pipelined"]
    assert all(torch.equal(x, y) for x, y in zip(sa[0], sb[0])) and torch.equal(sa[1], sb[1]) and torch.equal(sa[2], sb[2]) and torch.equal(sa[3], sb[3])
    assert sa[4] == sb[4] == 0.0     # both leave the device sums cleared
