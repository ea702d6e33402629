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


def _twin_agents(**over_b):
    """Two agents with the same weights, rollout dataset and normaliser state; `over_b` = config overrides of the second one."""
    from tests.test_gpu_round2 import _agent
    a = _agent(512, 4096, hip_graphs=False, **over_b)
    b = _agent(512, 4096, hip_graphs=False)
    b.model.load_state_dict(a.model.state_dict())
    a.obs = a.env_reset()
    a.play_steps()
    b._alloc_static()
    for k in a.dataset:
        b.dataset[k].copy_(a.dataset[k])
    b._mom_pack.copy_(a._mom_pack)
    for ra, rb in ((a.running_mean_std, b.running_mean_std), (a.value_mean_std, b.value_mean_std)):
        rb.load_state_dict(ra.state_dict())
    for ag in (a, b):
        if getattr(ag, "_hflat", None) is not None:
            ag.model.a2c_network.refresh_half()
        ag._packed_stale = True
    return a, b


def test_one_launch_gradient_reduction_equals_the_three_launch_path():
    """Round 4: the step's three second-stage reductions (split-K weight-gradient images, bias column sums, the loss kernel's sums)
    are ONE launch that WRITES the whole flat gradient.  Against the three-launch path (`fused_grad_reduce: False`: clear, then three
    accumulating reductions) on the same minibatch: weight gradients bit-equal (same partials, same order), bias / log-sigma
    gradients and the statistics to rounding (4 row lanes instead of 16) -- and the flat buffer is poisoned with NaN first: a slot
    the launch did not write would show."""
    import torch
    a, b = _twin_agents(fused_grad_reduce=False)   # a: three launches, b: one
    out = []
    for ag in (a, b):
        mb = ag._minibatch(1)
        ag._f_obs_rms.moments(mb["obs"], out=ag._obs_mom[1])
        if ag is b:
            ag._flat.fill_(float("nan"))
        ag._phase_b(mb)
        out.append(ag._flat.detach().clone())
    torch.cuda.synchronize()
    ga, gb = out
    n = a._nparam
    assert torch.isfinite(gb[:n + 5]).all()
    off, wmask = 0, torch.zeros(n, dtype=torch.bool, device=ga.device)
    for p in a.model.parameters():
        if p.dim() == 2:
            wmask[off:off + p.numel()] = True
        off += p.numel()
    assert torch.equal(ga[:n][wmask], gb[:n][wmask])
    scale = float(ga[:n].abs().max())
    np.testing.assert_allclose(gb[:n + 5].cpu().numpy(), ga[:n + 5].cpu().numpy(), rtol=2e-5, atol=2e-6 * scale)
    # and bit-reproducible: the same step again gives the same bits (the KL / entropy sums aside: the first pass wrote the current
    # mu / sigma back into the minibatch, PPODataset.update_mu_sigma)
    b._rms_preapplied = True   # (the moments were applied above: do not absorb them twice)
    b._flat.fill_(float("nan"))
    b._phase_b(b._minibatch(1))
    assert torch.equal(b._flat[:n + 3], gb[:n + 3])


def test_optimiser_launch_folds_the_weight_scatter_and_the_next_normaliser_update():
    """Round 4: bez_ppo_adam_step is one launch, and it also (a) writes the fragment-major weight copies the MFMA policy kernels read
    (what PackedWeights.refresh() = bez_ppo_scatter2_f16 did in front of the next forward) and (b) lets the input normaliser absorb
    the NEXT minibatch's moments (what bez_ppo_rms_apply did).  Both against the separate launches, bit for bit."""
    import torch
    from bez_isaacgym_amd.ppo import fused as F
    a, b = _twin_agents()
    for ag in (a, b):
        assert ag._fused_opt and ag._packed is not None and ag._hflat is not None
        for i in range(ag.num_minibatches):
            ag._f_obs_rms.moments(ag._minibatch(i)["obs"], out=ag._obs_mom[i])
        ag._packed.refresh(); ag._packed_stale = False
        ag._rms_preapplied = False
        ag.kl_acc.zero_(); ag.loss_acc.zero_()
    # a: one optimiser step with the extras folded in; b: the same step without them, then the two separate launches
    a._phase_b(a._minibatch(0)); a._phase_c(a.kl_acc[0], a.loss_acc, next_i=1)
    assert a._rms_preapplied
    b._phase_b(b._minibatch(0))
    packed, b._packed_stale = b._packed, True          # stale: _phase_c leaves the copies alone
    b._phase_c(b.kl_acc[0], b.loss_acc, next_i=None)
    packed.refresh()
    b._f_obs_rms.apply(b._obs_mom[1])
    torch.cuda.synchronize()
    assert torch.equal(a._pflat, b._pflat) and torch.equal(a._hflat, b._hflat)
    assert torch.equal(a._packed.flat, b._packed.flat)
    for k in ("running_mean", "running_var", "count"):
        assert torch.equal(getattr(a.running_mean_std, k), getattr(b.running_mean_std, k)), k
    assert float(a._opt_work[0]) == 0.0 and float(a.scaler._scale) == float(b.scaler._scale)
    assert torch.equal(a._steps, b._steps) and float(a._steps[0]) == 1.0


def test_rollout_bookkeeping_inside_the_policy_launch_changes_nothing():
    """Round 4: the reward shaping / done flags / episode statistics of env step n ride in the policy launch of step n + 1
    (BezPpoRolloutPost), so a rollout step is two launches.  Two agents, same seed, one with the separate bez_ppo_rollout_post launch:
    every rollout row, the dataset, the running episode sums and the statistics are bit-identical over two rollouts (the second
    one starts from the state the first one's LAST, separately launched, bookkeeping left)."""
    import torch
    from tests.test_gpu_round2 import _agent
    a = _agent(512, 4096, hip_graphs=False, fold_rollout_post=False)
    b = _agent(512, 4096, hip_graphs=False)
    b.model.load_state_dict(a.model.state_dict())
    for ag in (a, b):
        torch.manual_seed(11)
        ag.obs = ag.env_reset()
        ag.play_steps(); ag.play_steps()
    torch.cuda.synchronize()
    assert b._env_buf_ptrs[1] >= 2 * (b.horizon - 1)   # the fold was live (the env's buffers are persistent)
    for k in a.mb:
        assert torch.equal(a.mb[k], b.mb[k]), k
    for k in a.dataset:
        assert torch.equal(a.dataset[k], b.dataset[k]), k
    assert torch.equal(a.dones, b.dones) and torch.equal(a.current_rewards, b.current_rewards) and torch.equal(a.current_lengths, b.current_lengths)
    np.testing.assert_allclose(a.ep_stats.cpu().numpy(), b.ep_stats.cpu().numpy(), rtol=1e-12)   # fp64 atomics: order-dependent in the last bits
    assert float(a.ep_stats[0]) > 0   # episodes did end (random initial policy falls within the horizon)
    # round 5: b's policy launches add their finished-episode sums to per-workgroup slots (BezPpoRolloutPost.ep_parts, no atomics); the rollout's
    # last bookkeeping launch folded them into ep_stats (compared above) and left them cleared
    assert getattr(a, "_ep_parts", None) is not None and float(b._ep_parts.abs().sum()) == 0.0


@pytest.mark.parametrize("fast", [{}, {"dr_prelaunch": True}])
def test_randomised_rollout_at_full_speed_changes_nothing(fast):
    """Round 4, BASELINE config 5 at speed: with task.randomize=True the agent (a) lets the policy launch add the env's action noise
    (BezPpoActionNoise: the same Philox bits as the env's own lambda) and (b) runs the randomisation of the coming env step as one
    extra workgroup of that launch (bez_sim_dr_step_args; default) -- or, `dr_prelaunch`, launches the randomisation kernel early on a
    side stream (bez_sim_dr_prelaunch).  Against an agent that leaves both to the env (two more launches per step, in series),
    same seeds, redraws every 5 steps: every rollout row, the simulator's state, the per-env randomised parameters and the
    randomisation clocks are bit-identical."""
    import torch
    from bez_isaacgym_amd import abi
    from bez_isaacgym_amd.ppo.a2c_continuous import A2CAgent
    from bez_isaacgym_amd.utils.config import load_config
    from bez_isaacgym_amd.utils.rlgames_utils import RLGPUEnv, get_rlgames_env_creator

    def make(**over):
        cfg = load_config(["task=bez_kick", "num_envs=512", "headless=True", "task.task.randomize=True"])
        cfg["task"]["seed"] = 42
        cfg["task"]["task"]["randomization_params"]["frequency"] = 5
        venv = RLGPUEnv("rlgpu", 512, env_creator=get_rlgames_env_creator(cfg["task"], "bez_kick", "cuda:0", "cuda:0", 0, True))
        params = cfg["train"]["params"]
        params["config"].update(minibatch_size=4096, save_frequency=0, save_best_after=10 ** 9, hip_graphs=False, **over)
        return A2CAgent(params, venv, "cuda:0")
    a = make(fold_action_noise=False, dr_prelaunch=False, fold_dr_step=False)
    b = make(**fast)
    b.model.load_state_dict(a.model.state_dict())
    for ag in (a, b):
        torch.manual_seed(3)
        ag.obs = ag.env_reset()
        ag.play_steps(); ag.play_steps()
    torch.cuda.synchronize()
    ea, eb = a.vec_env.env, b.vec_env.env
    assert eb.action_noise_source() is not None and not eb.external_action_noise
    assert (getattr(b, "_side_stream", None) is not None) == bool(fast.get("dr_prelaunch"))
    for k in a.mb:
        assert torch.equal(a.mb[k], b.mb[k]), k
    for t in (abi.TENSOR_ROOT_STATE, abi.TENSOR_DOF_STATE, abi.TENSOR_RANDOMIZE_BUF, abi.TENSOR_DR_NOISE, abi.TENSOR_OBS, abi.TENSOR_RESET):
        assert torch.equal(ea.sim.refresh(t) if t < abi.TENSOR_OBS else ea.sim.tensor(t), eb.sim.refresh(t) if t < abi.TENSOR_OBS else eb.sim.tensor(t)), t
    for p in (abi.PARAM_FRICTION, abi.PARAM_KP_SCALE, abi.PARAM_KD_SCALE, abi.PARAM_DOF_LOWER, abi.PARAM_GRAVITY):
        assert torch.equal(ea.sim.get_env_params(p), eb.sim.get_env_params(p)), p
    fr = ea.sim.get_env_params(abi.PARAM_FRICTION)
    assert float(fr.std()) > 0   # redraws did happen
    # the noise really is in the actions the env stepped with: a noise-free twin differs
    assert float(ea.sim.tensor(abi.TENSOR_DR_NOISE)[3]) > 0


@pytest.mark.parametrize("fused", [False, True])
def test_eager_work_and_checkpoint_saves_between_graph_replays_change_nothing(fused, tmp_path):
    """VERDICT round 3, item 4 / ADVICE (medium): the plain AMP path replayed as HIP graphs 'stopped learning once agent.save() ran
    between replays'.  Root cause (tools/plain_graph_probe.py, profiles/r04_plain_graph_probe.txt): with the HIP runtime's graph packet
    capture on, kernel arguments launched EAGERLY between two replays (>= ~8 KB for the plain path's graph, a few hundred KB for the
    fused one) clobber the replayed kernels' arguments.  The package switches the feature off (bez_isaacgym_amd/__init__.py).  Here: two
    agents per path, same seed; one of them runs 1024 eager reductions (~800 KB of kernel arguments) and a full checkpoint save
    between every two replayed epochs.  Weights, Adam moments and the loss scale must stay bit-identical to the undisturbed twin."""
    import io
    import os
    import torch
    import bez_isaacgym_amd
    from tests.test_gpu_round2 import _agent
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0" and bez_isaacgym_amd.GRAPH_REPLAY_SAFE
    x = torch.ones(100, device="cuda")
    out = []
    for disturbed in (False, True):
        torch.manual_seed(7)
        ag = _agent(512, 4096, fused_ops=fused, hip_graphs=True)
        assert ag.use_graphs   # (the plain AMP path is replayed again: round 3 had to run it eagerly)
        ag.obs = ag.env_reset()
        for ep in range(14):
            ag.train_epoch()
            if disturbed and ag._g_update is not None:
                for _ in range(1024):
                    junk = x.sum()
                del junk
                torch.save(ag.get_full_state_weights(), io.BytesIO())
                ag.save(str(tmp_path / "ck.pth"))
        torch.cuda.synchronize()
        assert ag._g_update is not None
        out.append(([p.detach().clone() for p in ag.model.parameters()], float(ag.scaler.get_scale()),
                    [ag.optimizer.state[p]["exp_avg"].clone() for p in ag.model.parameters()]))
        del ag
    (pa, sa, ma), (pb, sb, mb) = out
    assert sa == sb and sa >= 1024.0
    assert all(torch.isfinite(p).all() for p in pb)
    for u, v in zip(pa + ma, pb + mb):
        assert torch.equal(u, v)


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("n,d,units,a", [(64 * 3 + 21, 34, (96, 50), 7), (64 * 4, 54, (400, 200, 100), 18)])
def test_backward_chain_matches_a_plain_torch_restatement(n, d, units, a, packed):
    """bez_ppo_policy_backward against the chain written out in torch (fp16 tensors, fp32 arithmetic, one rounding per autocast
    node): gz_L = fp16(g_L * elu'(y_L)), bias gradient = sum of the rounded gz_L, g_(L-1) = fp16(gz_L W_L).  Shapes: full 64-row tiles
    with widths that are multiples of 4 (the 8-byte fast path of the elementwise pass), a ragged last tile and a width that is only
    even (the guarded column-pair path), fragment-major and row-major weights."""
    import torch
    from bez_isaacgym_amd.ppo import fused as F
    from bez_isaacgym_amd.ppo.a2c_continuous import RunningMeanStd
    DEV = "cuda:0"
    torch.manual_seed(23)
    nl = len(units)
    dims = [d] + list(units)
    shapes = [(dims[i + 1], dims[i]) for i in range(nl)] + [(a, dims[-1]), (1, dims[-1])]
    hflat = (torch.randn(sum(o * k + o for o, k in shapes), device=DEV) * 0.1).half()
    views, layout, off = [], [], 0
    for o, k in shapes:
        w = hflat[off:off + o * k].view(o, k); layout.append((off, o, k)); off += o * k
        b = hflat[off:off + o]; off += o
        views.append((w, b))
    rms = RunningMeanStd((d,)).to(DEV)
    pk = F.PackedWeights(hflat, layout, a) if packed else None
    pf = F.PolicyForward(views[:nl], views[nl], views[nl + 1], rms, pk)
    obs = torch.randn(n, d, device=DEV)
    x0 = torch.zeros(n, d, device=DEV, dtype=torch.float16)
    acts = [torch.zeros(n, w, device=DEV, dtype=torch.float16) for w in units]
    mu, v = torch.zeros(n, a, device=DEV), torch.zeros(n, 1, device=DEV)
    pf.train_forward(obs, x0, acts, mu, v)
    # the forward the kernel kept, against torch on the same fp16 weights
    h = x0
    for (w, b), y in zip(views[:nl], acts):
        h = torch.nn.functional.elu((h.float() @ w.float().t() + b.float()).half().float()).half()
        torch.testing.assert_close(y.float(), h.float(), rtol=4e-3, atol=2e-3)
    pb = F.PolicyBackward(hflat, layout, a, pk)
    pb.refresh()
    gmu, gval = torch.randn(n, a, device=DEV) * 1e-2, torch.randn(n, 1, device=DEV) * 1e-2
    gz = [torch.zeros(n, w, device=DEV, dtype=torch.float16) for w in units]
    g16, v16 = torch.zeros(n, a, device=DEV, dtype=torch.float16), torch.zeros(n, 1, device=DEV, dtype=torch.float16)
    bg = [torch.zeros(w, device=DEV) for w in units]
    bm, bv = torch.zeros(a, device=DEV), torch.zeros(1, device=DEV)
    pb(gmu, gval, acts, gz, g16, v16, bg, bm, bv)
    assert torch.equal(g16, gmu.half()) and torch.equal(v16, gval.half())
    torch.testing.assert_close(bm, g16.float().sum(0), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bv, v16.float().sum(0), rtol=1e-4, atol=1e-5)
    g = (g16.float() @ views[nl][0].float() + v16.float() @ views[nl + 1][0].float()).half()
    for L in range(nl - 1, -1, -1):
        y = acts[L].float()
        ref = (g.float() * torch.where(y > 0, torch.ones_like(y), y + 1.0)).half()
        # (g itself is a rounded MFMA sum whose order differs from torch's: one fp16 ulp of g moves gz by one ulp)
        torch.testing.assert_close(gz[L].float(), ref.float(), rtol=4e-3, atol=2e-5)
        torch.testing.assert_close(bg[L], gz[L].float().sum(0), rtol=1e-4, atol=1e-5)   # the bias gradient sums what the kernel stored
        if L > 0:
            g = (gz[L].float() @ views[L][0].float()).half()


def test_dataset_prep_in_four_launches_equals_the_separate_launches():
    """bez_ppo_dataset_prep against the path it replaces (per-minibatch moments, value / return moments, two RunningMeanStd updates, two
    normalisations, advantage + its normalisation, env-major transposes): same dataset and the same normaliser statistics after a rollout."""
    import torch
    from tests.test_gpu_round2 import _agent
    out = []
    for fused_prep in (False, True):
        torch.manual_seed(11)
        ag = _agent(512, 4096, fused_ops=True, hip_graphs=False, fused_dataset_prep=fused_prep)
        ag.obs = ag.env_reset()
        for _ in range(2):
            ag.play_steps()
        torch.cuda.synchronize()
        assert (getattr(ag, "_prep_scratch", None) is not None) == fused_prep
        ds = ag.dataset
        out.append(dict(old_values=ds["old_values"].clone(), returns=ds["returns"].clone(), advantages=ds["advantages"].clone(),
                        obs_mom=ag._obs_mom.clone(), val_mom=ag._val_mom.clone(), ret_mom=ag._ret_mom.clone(),
                        vmean=ag.value_mean_std.running_mean.clone(), vvar=ag.value_mean_std.running_var.clone(), vcount=ag.value_mean_std.count.clone()))
        del ag
    a, b = out
    assert float(a["advantages"].abs().max()) > 0 and float(a["old_values"].abs().max()) > 0
    for k in ("obs_mom", "val_mom", "ret_mom", "vmean", "vvar", "vcount"):
        torch.testing.assert_close(b[k], a[k], rtol=1e-12, atol=1e-9)    # fp64 sums, a different fixed order
    for k in ("old_values", "returns"):
        torch.testing.assert_close(b[k].reshape(-1), a[k].reshape(-1), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(b["advantages"].reshape(-1), a["advantages"].reshape(-1), rtol=2e-5, atol=2e-5)   # torch's fp32 mean / std against fp64 sums


def test_loss_in_front_of_the_backward_chain_changes_nothing():
    """bez_ppo_policy_backward_with_loss (the tile's loss terms and d loss / d mu, d loss / d value formed in the backward launch, read from
    LDS) against bez_ppo_loss + bez_ppo_policy_backward: the same code on the same numbers -- weights, Adam moments, loss scale and the
    dataset's updated mu / sigma are bit-identical after four epochs (80 optimiser steps, update_mu_sigma and adaptive lr included)."""
    import torch
    from tests.test_gpu_round2 import _agent
    out = []
    for fused in (False, True):
        torch.manual_seed(5)
        ag = _agent(512, 4096, hip_graphs=False, fused_loss_backward=fused)
        torch.manual_seed(5)
        ag.model.load_state_dict({k: v.clone() for k, v in ag.model.state_dict().items()})
        seen = []
        if fused:   # the fused entry point is what runs (it would fall back silently on an unsupported shape)
            orig = ag._policy_bwd.with_loss
            def spy(*a, **k):
                r = orig(*a, **k)
                seen.append(r)
                return r
            ag._policy_bwd.with_loss = spy
        ag.obs = ag.env_reset()
        stats = [ag.train_epoch() for _ in range(4)]
        torch.cuda.synchronize()
        if fused:
            assert len(seen) == 4 * ag.mini_epochs * ag.num_minibatches and all(seen)
        out.append(([p.detach().clone() for p in ag.model.parameters()], [ag.optimizer.state[p]["exp_avg_sq"].clone() for p in ag.model.parameters()],
                    float(ag.scaler.get_scale()), ag.dataset["mu"].clone(), ag.dataset["sigma"].clone(), [s["kl"] for s in stats]))
        del ag
    (pa, va, sa, mua, sga, kla), (pb, vb, sb, mub, sgb, klb) = out
    assert sa == sb and kla == klb
    for u, v in zip(pa + va + [mua, sga], pb + vb + [mub, sgb]):
        assert torch.equal(u, v)


@pytest.mark.parametrize("h,n,nmb,with_rms,norm_adv", [(32, 4096, 4, True, True), (16, 256, 2, False, True), (8, 64, 0, True, False), (5, 333, 1, True, True)])
def test_dataset_prep_kernels_against_a_torch_restatement(h, n, nmb, with_rms, norm_adv):
    """bez_ppo_dataset_prep called directly, against prepare_dataset written out in torch (fp64 moments, RunningMeanStd's parallel-variance
    update twice, fp32 normalisation with the clamp, advantage and its unbiased-std normalisation, env-major transposes): with and without
    the value normaliser, with and without advantage normalisation, without observations, and a size the kernel declines (H * N not a
    multiple of 64 -> False, nothing written)."""
    import torch
    from bez_isaacgym_amd.ppo import fused as F
    from bez_isaacgym_amd.ppo.a2c_continuous import RunningMeanStd
    DEV = "cuda:0"
    torch.manual_seed(3)
    d = 54
    values, returns = torch.randn(h, n, 1, device=DEV) * 2 + 0.5, torch.randn(h, n, 1, device=DEV) * 3 - 1.0
    rows = h * n // max(nmb, 1)
    obs = torch.randn(max(nmb, 1) * rows, d, device=DEV) * 1.7 + 0.3
    rms = RunningMeanStd((1,)).to(DEV) if with_rms else None
    if rms is not None:
        rms.running_mean.fill_(0.2); rms.running_var.fill_(1.5); rms.count.fill_(1000.0)
        ref = RunningMeanStd((1,)).to(DEV)
        ref.load_state_dict(rms.state_dict())
    obs_mom = torch.zeros(max(nmb, 1), 2 * d + 1, device=DEV, dtype=torch.float64)
    vm, rm = torch.zeros(3, device=DEV, dtype=torch.float64), torch.zeros(3, device=DEV, dtype=torch.float64)
    ov, rt, adv = (torch.full((h * n, 1), 7.0, device=DEV), torch.full((h * n, 1), 7.0, device=DEV), torch.full((h * n,), 7.0, device=DEV))
    sc = F.dataset_prep_scratch(nmb, h, n, DEV)
    ok = F.dataset_prep(obs if nmb else None, rows, nmb, obs_mom[:max(nmb, 1)].reshape(-1)[:nmb * (2 * d + 1)] if nmb else None, values, returns, rms, vm, rm, ov, rt, adv, norm_adv, sc)
    if (h * n) % 64:
        assert ok is False and float(ov.min()) == 7.0 and float(adv.min()) == 7.0
        return
    assert ok is True
    flat = lambda t: t.transpose(0, 1).reshape(h * n, 1)
    v, r = flat(values), flat(returns)
    for i in range(nmb):
        x = obs[i * rows:(i + 1) * rows].double()
        want = torch.cat([x.sum(0), (x * x).sum(0), torch.tensor([float(rows)], device=DEV, dtype=torch.float64)])
        torch.testing.assert_close(obs_mom[i], want, rtol=1e-12, atol=1e-9)
    for mom, t in ((vm, v), (rm, r)):
        want = torch.stack([t.double().sum(), (t.double() ** 2).sum(), torch.tensor(float(h * n), device=DEV, dtype=torch.float64)])
        torch.testing.assert_close(mom, want, rtol=1e-12, atol=1e-9)
    if rms is not None:
        ref.eval()
        ref.update_from_moments(vm); v = ref(v)
        ref.update_from_moments(rm); r = ref(r)
        for k in ("running_mean", "running_var", "count"):
            torch.testing.assert_close(getattr(rms, k), getattr(ref, k), rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(ov, v, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(rt, r, rtol=1e-6, atol=1e-6)
    a = (r - v).sum(dim=1)
    if norm_adv:
        a = (a - a.mean()) / (a.std() + 1e-8)
    torch.testing.assert_close(adv, a, rtol=2e-5, atol=2e-5)
