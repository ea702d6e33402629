"""GPU: the PPO glue kernels (csrc/bez_ppo.hip, C ABI bez_ppo_*) against the plain torch formulation of the same
rl_games a2c_continuous arithmetic (bez_isaacgym_amd/ppo/a2c_continuous.py) -- term by term, then a whole optimiser step."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_running_mean_std_kernels():
    from bez_isaacgym_amd.ppo.a2c_continuous import RunningMeanStd
    from bez_isaacgym_amd.ppo.fused import FusedRunningMeanStd
    torch.manual_seed(0)
    ref, fus = RunningMeanStd((54,)).to(DEV), RunningMeanStd((54,)).to(DEV)
    f = FusedRunningMeanStd(fus)
    ref.train()
    for b in (32768, 4096, 100):
        x = (torch.randn(b, 54, device=DEV) * torch.linspace(0.1, 5, 54, device=DEV) + 2.0).contiguous()
        ref(x)
        f.update(x)
        np.testing.assert_allclose(fus.running_mean.cpu(), ref.running_mean.cpu(), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(fus.running_var.cpu(), ref.running_var.cpu(), rtol=1e-10)
        assert float(fus.count) == float(ref.count)
    ref.eval()
    y = ref(x)
    np.testing.assert_allclose(f.normalize(x, torch.empty_like(x)).cpu(), y.cpu(), atol=2e-6)
    y16 = f.normalize(x, torch.empty(x.shape, device=DEV, dtype=torch.float16))
    np.testing.assert_allclose(y16.float().cpu(), y.cpu(), atol=3e-3)
    assert float(y.abs().max()) <= 5.0


def test_sample_and_rollout_bookkeeping_kernels():
    from bez_isaacgym_amd.ppo import fused as F
    from bez_isaacgym_amd.ppo.a2c_continuous import ModelA2CContinuousLogStd
    torch.manual_seed(1)
    n, a = 4096, 18
    mu, logstd, noise = torch.randn(n, a, device=DEV), torch.randn(a, device=DEV) * 0.5 - 1, torch.randn(n, a, device=DEV)
    act, env_act, nlp, sig = (torch.empty(n, a, device=DEV), torch.empty(n, a, device=DEV), torch.empty(n, device=DEV), torch.empty(n, a, device=DEV))
    F.sample(mu, logstd, noise, act, env_act, nlp, sig)
    ref_act = mu + torch.exp(logstd) * noise
    np.testing.assert_allclose(act.cpu(), ref_act.cpu(), atol=1e-6)
    np.testing.assert_allclose(env_act.cpu(), ref_act.clamp(-1, 1).cpu(), atol=1e-6)
    ref_nlp = ModelA2CContinuousLogStd.neglogp(act, mu, torch.exp(logstd), logstd)
    np.testing.assert_allclose(nlp.cpu(), ref_nlp.cpu(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(sig.cpu(), torch.exp(logstd).expand(n, a).cpu(), rtol=1e-6)
    # one env step of bookkeeping
    rew = torch.randn(n, device=DEV); dones = (torch.rand(n, device=DEV) < 0.1).long(); tmo = (torch.rand(n, device=DEV) < 0.05).long()
    val = torch.randn(n, 1, device=DEV)
    cur_r, cur_l = torch.randn(n, device=DEV), torch.randint(0, 50, (n,), device=DEV).float()
    r0, l0 = cur_r.clone(), cur_l.clone()
    shaped, dones_f, stats = torch.empty(n, 1, device=DEV), torch.empty(n, device=DEV), torch.zeros(3, device=DEV, dtype=torch.float64)
    F.rollout_post(rew, dones, tmo, val, 0.01, 0.99, True, shaped, dones_f, cur_r, cur_l, stats)
    np.testing.assert_allclose(shaped.cpu(), (rew.unsqueeze(1) * 0.01 + 0.99 * val * tmo.unsqueeze(1).float()).cpu(), atol=1e-6)
    d = dones.float()
    np.testing.assert_array_equal(dones_f.cpu(), d.cpu())
    np.testing.assert_allclose(cur_r.cpu(), ((r0 + rew) * (1 - d)).cpu(), atol=1e-6)
    np.testing.assert_allclose(cur_l.cpu(), ((l0 + 1) * (1 - d)).cpu(), atol=0)
    np.testing.assert_allclose(stats.cpu(), torch.stack([d.sum(), ((r0 + rew) * d).sum(), ((l0 + 1) * d).sum()]).double().cpu(), rtol=1e-5)


@pytest.mark.parametrize("half", [True, False])
def test_rollout_pre_kernel(half):
    """One launch = .float() of the network outputs + RunningMeanStd(unnorm) of the value + four rollout-buffer rows + sampling."""
    from bez_isaacgym_amd.ppo import fused as F
    from bez_isaacgym_amd.ppo.a2c_continuous import ModelA2CContinuousLogStd, RunningMeanStd
    torch.manual_seed(4)
    n, a, d = 1000, 18, 54
    dt = torch.float16 if half else torch.float32
    mu, value = (torch.randn(n, a, device=DEV) * 0.7).to(dt), (torch.randn(n, 1, device=DEV) * 4).to(dt)
    logstd, noise = torch.randn(a, device=DEV) * 0.2 - 0.5, torch.randn(n, a, device=DEV)
    obs, dones = torch.randn(n, d, device=DEV), (torch.rand(n, device=DEV) < 0.1).float()
    rms = RunningMeanStd((1,)).to(DEV)
    rms.running_mean.fill_(0.3); rms.running_var.fill_(2.5)
    z = lambda *s: torch.full(s, -7.0, device=DEV)
    mb_obs, mb_dones, mb_mu, mb_val, act, env_act, nlp, sig = z(n, d), z(n), z(n, a), z(n, 1), z(n, a), z(n, a), z(n), z(n, a)
    F.rollout_pre(mu, value, logstd, noise, obs, dones, rms, mb_obs, mb_dones, mb_mu, mb_val, act, env_act, nlp, sig)
    mu32, sigma = mu.float(), torch.exp(logstd).expand(n, a)
    assert torch.equal(mb_obs, obs) and torch.equal(mb_dones, dones) and torch.equal(mb_mu, mu32)
    rms.eval()
    np.testing.assert_allclose(mb_val.cpu(), rms(value.float(), True).cpu(), rtol=1e-6, atol=1e-6)
    want = mu32 + sigma * noise
    np.testing.assert_allclose(act.cpu(), want.cpu(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(env_act.cpu(), want.clamp(-1, 1).cpu(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sig.cpu(), sigma.cpu(), rtol=1e-6)
    np.testing.assert_allclose(nlp.cpu(), ModelA2CContinuousLogStd.neglogp(act, mu32, sigma, logstd.expand(n, a)).cpu(), rtol=1e-5, atol=1e-4)
    F.rollout_pre(mu, value, logstd, noise, obs, dones, None, mb_obs, mb_dones, mb_mu, mb_val, act, env_act, nlp, sig)
    assert torch.equal(mb_val, value.float())


@pytest.mark.parametrize("n,d,units,a", [(4096, 54, (400, 200, 100), 18), (1000, 54, (400, 200, 100), 18), (130, 33, (96, 50), 7)])
def test_policy_forward_kernel_matches_torch_fp16_path(n, d, units, a):
    """The one-launch MFMA forward of the rollout against the same network evaluated by torch on the same fp16 weights
    (addmm + ELU per layer, fp16 tensors).  Asymmetric random weights: a swapped fragment map cannot pass."""
    from bez_isaacgym_amd.ppo import fused as F
    from bez_isaacgym_amd.ppo.a2c_continuous import RunningMeanStd
    torch.manual_seed(13)
    dims = [d] + list(units)
    hidden = [((torch.randn(dims[i + 1], dims[i], device=DEV) / dims[i] ** 0.5).half().contiguous(), (torch.randn(dims[i + 1], device=DEV) * 0.1).half())
              for i in range(len(units))]
    mu_wb = ((torch.randn(a, dims[-1], device=DEV) / dims[-1] ** 0.5).half().contiguous(), (torch.randn(a, device=DEV) * 0.1).half())
    val_wb = ((torch.randn(1, dims[-1], device=DEV) / dims[-1] ** 0.5).half().contiguous(), (torch.randn(1, device=DEV) * 0.1).half())
    rms = RunningMeanStd((d,)).to(DEV)
    rms.running_mean.copy_(torch.randn(d, dtype=torch.float64) * 0.3); rms.running_var.copy_(torch.rand(d, dtype=torch.float64) + 0.5)
    obs = torch.randn(n, d, device=DEV) * 1.5
    mu, val = torch.full((n, a), 7.0, device=DEV), torch.full((n, 1), 7.0, device=DEV)
    F.PolicyForward(hidden, mu_wb, val_wb, rms)(obs, mu, val)
    rms.eval()
    h = rms(obs).half()
    for w, b in hidden:
        h = torch.nn.functional.elu(torch.addmm(b, h, w.t()))
    want_mu, want_v = torch.addmm(mu_wb[1], h, mu_wb[0].t()).float(), torch.addmm(val_wb[1], h, val_wb[0].t()).float()
    # fp16 outputs of O(1): one or two fp16 ulps (different accumulation order inside the GEMMs, then rounding at every layer)
    np.testing.assert_allclose(mu.cpu(), want_mu.cpu(), rtol=4e-3, atol=4e-3)
    np.testing.assert_allclose(val.cpu(), want_v.cpu(), rtol=4e-3, atol=4e-3)
    assert float((mu - want_mu).abs().mean()) < 4e-4


@pytest.mark.parametrize("n,d,units,a", [(4096, 54, (400, 200, 100), 18), (130, 33, (96, 50), 7)])
def test_policy_forward_kernel_against_an_fp32_reference_with_a_derived_bound(n, d, units, a):
    """The MFMA forward against a plain fp32 evaluation of the same network (the test above holds it against
    torch's fp16 path).  What separates the two is known: the kernel rounds the normalised input and every layer's output to fp16
    (as torch's fp16 path does) and accumulates fp16 x fp16 products in fp32.  The reference below applies exactly those roundings to an
    fp32 / fp64 evaluation (products of fp16 numbers are exact in fp64), so the remaining difference is (i) the fp32 accumulation order
    inside a dot product of K terms -- bounded by K * eps32 * sum |w_k x_k| -- and (ii) the resulting occasional flip of an fp16 rounding, one
    fp16 ulp of the layer's output, which the next layers carry forward with gain <= ||W||_inf row sums.  The bound asserted is that
    propagation, computed from the actual weights."""
    from bez_isaacgym_amd.ppo import fused as F
    from bez_isaacgym_amd.ppo.a2c_continuous import RunningMeanStd
    torch.manual_seed(17)
    dims = [d] + list(units)
    hidden = [((torch.randn(dims[i + 1], dims[i], device=DEV) / dims[i] ** 0.5).half().contiguous(), (torch.randn(dims[i + 1], device=DEV) * 0.1).half())
              for i in range(len(units))]
    mu_wb = ((torch.randn(a, dims[-1], device=DEV) / dims[-1] ** 0.5).half().contiguous(), (torch.randn(a, device=DEV) * 0.1).half())
    val_wb = ((torch.randn(1, dims[-1], device=DEV) / dims[-1] ** 0.5).half().contiguous(), (torch.randn(1, device=DEV) * 0.1).half())
    rms = RunningMeanStd((d,)).to(DEV)
    rms.running_mean.copy_(torch.randn(d, dtype=torch.float64) * 0.3); rms.running_var.copy_(torch.rand(d, dtype=torch.float64) + 0.5)
    obs = torch.randn(n, d, device=DEV) * 1.5
    mu, val = torch.full((n, a), 7.0, device=DEV), torch.full((n, 1), 7.0, device=DEV)
    F.PolicyForward(hidden, mu_wb, val_wb, rms)(obs, mu, val)
    rms.eval()
    h = rms(obs).half().double()                       # the kernel's fp16 input tile
    ulp16 = 2.0 ** -10                                 # relative spacing of fp16
    err = torch.zeros(n, 1, dtype=torch.float64, device=DEV)   # running bound on |kernel - reference| of a layer's fp16 output, per row
    for w, b in hidden:
        z = h @ w.double().t() + b.double()            # exact products, fp64 sums: the "infinitely precise" layer
        k = w.shape[1]
        acc = k * 2.0 ** -24 * (h.abs() @ w.double().abs().t())            # (i) fp32 accumulation order
        gain = w.double().abs().sum(1).max()                                # row-sum norm: how an input error spreads
        y = torch.nn.functional.elu(z.float().half().float()).half()       # fp16 rounding of the pre-activation and of the ELU, as the kernel
        err = err * gain + acc.max(1, keepdim=True).values + 2 * ulp16 * y.double().abs().max(1, keepdim=True).values.clamp_min(2.0 ** -14)
        h = y.double()
    for (w, b), got in ((mu_wb, mu), (val_wb, val)):
        z = h @ w.double().t() + b.double()
        bound = err * w.double().abs().sum(1).max() + w.shape[1] * 2.0 ** -24 * (h.abs() @ w.double().abs().t()).max(1, keepdim=True).values + ulp16 * z.abs() + 1e-6
        diff = (got.double() - z).abs()
        assert bool((diff <= bound).all()), (float(diff.max()), float(bound.min()), float(bound.max()))
        assert float(diff.mean()) < 5e-4, float(diff.mean())   # and on average far inside it: most elements see no rounding flip at all


@pytest.mark.parametrize("n,normalize_value,units", [
    (4096, True, (400, 400, 200, 100)), (333, False, (400, 400, 200, 100)),
    # the yaml's three hidden layers (the last activations end in the OTHER tile) at the edges of the 32-row tiling: one row, a tile minus / plus
    # one row, the largest launch on 32-row tiles and the first on 64-row tiles (csrc/bez_policy.hip small_batch)
    (1, True, (400, 200, 100)), (31, False, (400, 200, 100)), (33, True, (400, 200, 100)), (8192, False, (400, 200, 100)), (8193, True, (400, 200, 100))])
def test_policy_rollout_step_equals_forward_then_rollout_pre(n, normalize_value, units):
    """bez_ppo_policy_rollout_step = bez_ppo_policy_forward + bez_ppo_rollout_pre in one launch: same mu / value / action /
    clamp / sigma / obs / dones rows bit for bit; neglogp differs only by the order of an 18-term fp32 sum."""
    from bez_isaacgym_amd.ppo import fused as F
    from bez_isaacgym_amd.ppo.a2c_continuous import RunningMeanStd
    torch.manual_seed(21)
    d, a = 54, 18
    dims = [d] + list(units)
    hidden = [((torch.randn(dims[i + 1], dims[i], device=DEV) / dims[i] ** 0.5).half().contiguous(), (torch.randn(dims[i + 1], device=DEV) * 0.1).half())
              for i in range(len(units))]
    mu_wb = ((torch.randn(a, dims[-1], device=DEV) / dims[-1] ** 0.5).half().contiguous(), (torch.randn(a, device=DEV) * 0.1).half())
    val_wb = ((torch.randn(1, dims[-1], device=DEV) * 3 / dims[-1] ** 0.5).half().contiguous(), (torch.randn(1, device=DEV) * 0.1).half())
    rms = RunningMeanStd((d,)).to(DEV)
    rms.running_mean.copy_(torch.randn(d, dtype=torch.float64) * 0.3); rms.running_var.copy_(torch.rand(d, dtype=torch.float64) + 0.5)
    vrms = None
    if normalize_value:
        vrms = RunningMeanStd((1,)).to(DEV)
        vrms.running_mean.fill_(0.7); vrms.running_var.fill_(2.5)
    obs = torch.randn(n, d, device=DEV) * 1.5
    logstd = torch.randn(a, device=DEV) * 0.3 - 0.5
    noise = torch.randn(n, a, device=DEV)
    dones = (torch.rand(n, device=DEV) < 0.1).float()
    pf = F.PolicyForward(hidden, mu_wb, val_wb, rms)

    def bufs():
        z = lambda *s: torch.full(s, 7.0, device=DEV)
        return dict(mb_obs=z(n, d), mb_dones=z(n), mb_mu=z(n, a), mb_val=z(n, 1), act=z(n, a), env_act=z(n, a), neglogp=z(n), sigma=z(n, a))
    want, got = bufs(), bufs()
    mu, val = torch.empty(n, a, device=DEV), torch.empty(n, 1, device=DEV)
    pf(obs, mu, val)
    F.rollout_pre(mu, val, logstd, noise, obs, dones, vrms, want["mb_obs"], want["mb_dones"], want["mb_mu"], want["mb_val"], want["act"], want["env_act"],
                  want["neglogp"], want["sigma"])
    pf.rollout_step(obs, logstd, noise, dones, vrms, got["mb_obs"], got["mb_dones"], got["mb_mu"], got["mb_val"], got["act"], got["env_act"], got["neglogp"],
                    got["sigma"])
    for k in want:
        if k == "neglogp":
            np.testing.assert_allclose(got[k].cpu(), want[k].cpu(), rtol=2e-6, atol=1e-5)
        else:
            np.testing.assert_array_equal(got[k].cpu().numpy(), want[k].cpu().numpy(), err_msg=k)
    assert float(got["env_act"].abs().max()) <= 1.0
    if n >= 333:
        assert float((got["act"].abs() > 1.0).float().mean()) > 0.01  # the clamp did something


def _torch_loss(mu, logstd, value, mb, e, critic_coef, entropy_coef, bounds_coef, clip_value):
    from bez_isaacgym_amd.ppo.a2c_continuous import ModelA2CContinuousLogStd, policy_kl
    sigma = torch.exp(logstd).unsqueeze(0).expand_as(mu)
    neglogp = ModelA2CContinuousLogStd.neglogp(mb["actions"], mu, sigma, logstd.unsqueeze(0).expand_as(mu))
    entropy = (0.5 + 0.5 * math.log(2 * math.pi) + logstd.unsqueeze(0).expand_as(mu)).sum(-1)
    ratio = torch.exp(mb["old_logp"] - neglogp)
    a_loss = torch.max(-mb["advantages"] * ratio, -mb["advantages"] * torch.clamp(ratio, 1 - e, 1 + e))
    if clip_value:
        vclip = mb["old_values"] + (value - mb["old_values"]).clamp(-e, e)
        c_loss = torch.max((value - mb["returns"]) ** 2, (vclip - mb["returns"]) ** 2)
    else:
        c_loss = (mb["returns"] - value) ** 2
    b_loss = (torch.clamp_min(mu - 1.1, 0.0) ** 2 + torch.clamp_max(mu + 1.1, 0.0) ** 2).sum(-1)
    loss = a_loss.mean() + 0.5 * c_loss.mean() * critic_coef - entropy.mean() * entropy_coef + b_loss.mean() * bounds_coef
    kl = policy_kl(mu.detach(), sigma.detach(), mb["mu"], mb["sigma"])
    return loss, a_loss.mean(), c_loss.mean(), b_loss.mean(), kl, entropy.mean()


@pytest.mark.parametrize("b,a", [(32768, 18), (1000, 7)])  # config 3's minibatch; a ragged batch (partial workgroup) of another action width
@pytest.mark.parametrize("clip_value,entropy_coef", [(True, 0.0), (False, 0.01)])
def test_loss_kernel_value_and_gradient(clip_value, entropy_coef, b, a):
    from bez_isaacgym_amd.ppo import fused as F
    torch.manual_seed(2)
    mu = (torch.randn(b, a, device=DEV) * 0.8).requires_grad_()
    logstd = (torch.randn(a, device=DEV) * 0.3 - 1.0).requires_grad_()
    value = torch.randn(b, 1, device=DEV).requires_grad_()
    old_sigma = torch.exp(logstd.detach() + 0.05 * torch.randn(a, device=DEV)).expand(b, a).contiguous()
    old_mu = (mu.detach() + 0.05 * torch.randn(b, a, device=DEV)).contiguous()
    actions = (old_mu + old_sigma * torch.randn(b, a, device=DEV)).contiguous()
    from bez_isaacgym_amd.ppo.a2c_continuous import ModelA2CContinuousLogStd
    old_logp = ModelA2CContinuousLogStd.neglogp(actions, old_mu, old_sigma, torch.log(old_sigma)) + 0.3 * torch.randn(b, device=DEV)
    mb = dict(actions=actions, old_logp=old_logp.contiguous(), advantages=torch.randn(b, device=DEV), old_values=torch.randn(b, 1, device=DEV),
              returns=torch.randn(b, 1, device=DEV), mu=old_mu, sigma=old_sigma)
    e, cc, bc = 0.2, 2.0, 0.001
    loss, a_l, c_l, b_l, kl, ent = _torch_loss(mu, logstd, value, mb, e, cc, entropy_coef, bc, clip_value)
    S = 1024.0
    (loss * S).backward()
    gmu, gval, glog, stats = torch.empty(b, a, device=DEV), torch.empty(b, 1, device=DEV), torch.empty(a, device=DEV), torch.empty(5, device=DEV)
    F.loss(mu.detach(), logstd.detach(), value.detach(), mb, e, cc, entropy_coef, bc, clip_value, torch.tensor([S], device=DEV), gmu, gval, glog, stats)
    st = (stats / b).cpu().numpy()
    np.testing.assert_allclose(st[:5], [float(a_l), float(c_l), float(b_l), float(kl), float(ent)], rtol=2e-4, atol=2e-6)
    scale = float(mu.grad.abs().max())
    np.testing.assert_allclose(gmu.cpu(), mu.grad.cpu(), atol=2e-5 * scale, rtol=1e-4)
    np.testing.assert_allclose(gval.cpu(), value.grad.cpu(), atol=2e-5 * float(value.grad.abs().max()), rtol=1e-4)
    np.testing.assert_allclose(glog.cpu(), logstd.grad.cpu(), rtol=2e-3, atol=2e-4 * float(logstd.grad.abs().max()))
    # fixed-order sums (scratch): the same values, and bit-identical from call to call (the atomic path is not)
    sc = F.loss_scratch(b, a, DEV)
    outs = []
    for _ in range(3):
        gl, st2 = torch.zeros(a, device=DEV), torch.zeros(5, device=DEV)
        F.loss(mu.detach(), logstd.detach(), value.detach(), mb, e, cc, entropy_coef, bc, clip_value, torch.tensor([S], device=DEV), torch.empty_like(gmu), gval,
               gl, st2, zero_glog=False, zero_stats=False, scratch=sc)
        outs.append((gl.clone(), st2.clone()))
    assert all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    np.testing.assert_allclose(outs[0][1].cpu(), stats.cpu(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(outs[0][0].cpu(), glog.cpu(), rtol=1e-4, atol=2e-4 * float(logstd.grad.abs().max()))
    # rl_games' PPODataset.update_mu_sigma folded into the same launch: identical results, and afterwards the minibatch's old mu / sigma
    # hold the current policy's (a second pass measures its KL against this one: exactly 0 for unchanged weights up to the 1e-5 terms)
    keep_mu, keep_sigma = old_mu.clone(), old_sigma.clone()
    g2, s2 = torch.empty_like(gmu), torch.empty(5, device=DEV)
    F.loss(mu.detach(), logstd.detach(), value.detach(), mb, e, cc, entropy_coef, bc, clip_value, torch.tensor([S], device=DEV), g2, gval, glog, s2,
           update_mu_sigma=True)
    assert torch.equal(g2, gmu) and torch.allclose(s2, stats, rtol=1e-5, atol=1e-3)   # (the sums are float atomics: order varies)
    assert torch.equal(mb["mu"], mu.detach()) and torch.allclose(mb["sigma"], torch.exp(logstd.detach()).expand(b, a), rtol=1e-6, atol=0)
    assert not torch.equal(keep_mu, mb["mu"]) and not torch.equal(keep_sigma, mb["sigma"])
    F.loss(mu.detach(), logstd.detach(), value.detach(), mb, e, cc, entropy_coef, bc, clip_value, torch.tensor([S], device=DEV), g2, gval, glog, s2)
    assert abs(float(s2[3]) / b) < 2e-3   # (c2's 1e-5 makes the self-KL slightly negative: -0.5 * 1e-5 / sigma^2 per action)


def test_half_linear_gradient_reductions():
    """split-K partial sums and bias column sums of the fp16 linear layers, reduced straight into an fp32 gradient."""
    from bez_isaacgym_amd.ppo import fused as F
    torch.manual_seed(9)
    part = torch.randn(64, 200, 400, device=DEV).half()
    out = torch.randn(200, 400, device=DEV)
    want = out + part.float().sum(0)
    F.wgrad_sum(part, out, accumulate=True)
    np.testing.assert_allclose(out.cpu(), want.cpu(), rtol=1e-5, atol=1e-4)
    F.wgrad_sum(part, out, accumulate=False)
    np.testing.assert_allclose(out.cpu(), part.float().sum(0).cpu(), rtol=1e-5, atol=1e-4)
    part = torch.randn(16, 7, 9, device=DEV).half()  # odd element count: scalar loads
    out = torch.zeros(7, 9, device=DEV)
    F.wgrad_sum(part, out)
    np.testing.assert_allclose(out.cpu(), part.float().sum(0).cpu(), rtol=1e-5, atol=1e-4)
    for b, d in ((32768, 400), (1000, 19), (513, 100), (32768, 1)):
        y = torch.randn(b, d, device=DEV).half()
        o = torch.ones(d, device=DEV)
        F.colsum_f16(y, o, accumulate=True)
        np.testing.assert_allclose(o.cpu(), (1.0 + y.float().sum(0)).cpu(), rtol=1e-4, atol=2e-2)
        F.colsum_f16(y, o, accumulate=False)
        np.testing.assert_allclose(o.cpu(), y.float().sum(0).cpu(), rtol=1e-4, atol=2e-2)


def test_elu_backward_fused_with_bias_gradient():
    from bez_isaacgym_amd.ppo import fused as F
    torch.manual_seed(11)
    for b, d in ((32768, 400), (777, 101)):
        pre = (torch.randn(b, d, device=DEV) * 1.5).half().requires_grad_()
        y = torch.nn.functional.elu(pre)
        gy = torch.randn(b, d, device=DEV).half()
        (want,) = torch.autograd.grad(y, pre, gy)
        gz, gb = torch.empty_like(gy), torch.full((d,), 2.0, device=DEV)
        F.elu_bwd_colsum_f16(gy, y.detach(), gz, gb, accumulate=True)
        np.testing.assert_allclose(gz.float().cpu(), want.float().cpu(), rtol=2e-3, atol=1e-3)  # one fp16 ulp: torch derives elu' from the input
        np.testing.assert_allclose(gb.cpu(), (2.0 + gz.float().sum(0)).cpu(), rtol=1e-4, atol=2e-2)


def test_adam_step_kernel_matches_torch_amp_clip_adam():
    """unscale + clip_grad_norm_ + Adam + GradScaler.update in one call vs the torch objects it replaces, over 7 steps: clean
    steps, a step whose gradient holds an inf (skipped by both, scale backed off), and a growth of the scale (interval 3)."""
    from bez_isaacgym_amd.ppo import fused as F
    torch.manual_seed(5)
    shapes = [(400, 54), (400,), (200, 400), (200,), (18,)]
    n = sum(int(np.prod(sh)) for sh in shapes)
    ref = [torch.nn.Parameter(torch.randn(sh, device=DEV) * 0.1) for sh in shapes]
    lr = torch.tensor(3e-4, device=DEV)
    opt = torch.optim.Adam(ref, lr=lr, eps=1e-8, capturable=True, fused=True)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=3)
    scaler._lazy_init_scale_growth_tracker(torch.device(DEV))  # what scaler.scale(loss) would do
    pflat = torch.cat([p.detach().reshape(-1) for p in ref]).clone()
    mflat, vflat = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    steps, work = torch.zeros(len(shapes), device=DEV), torch.zeros(F.ADAM_WORK_FLOATS, device=DEV)
    hflat = torch.zeros(n, device=DEV, dtype=torch.float16)
    scale, tracker = torch.tensor([1024.0], device=DEV), torch.zeros(1, device=DEV, dtype=torch.int32)
    seen = []
    acc, src = torch.zeros(3, device=DEV), torch.tensor([1.5, -2.0, 0.25], device=DEV)
    for it in range(7):
        g = [torch.randn(sh, device=DEV) * (3.0 if it % 2 else 0.02) for sh in shapes]  # norms above and below the clip threshold
        if it == 4:
            g[2][3, 7] = float("inf")
        s_ref = scaler.get_scale()
        for p, gi in zip(ref, g):
            p.grad = gi * s_ref
        gflat = torch.cat([(gi * float(scale)).reshape(-1) for gi in g]).contiguous()
        scaler.unscale_(opt)
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        scaler.step(opt)
        scaler.update()
        F.adam_step(pflat, gflat, mflat, vflat, steps, lr, (0.9, 0.999), 1e-8, 0.0, 1.0, scale, tracker, 2.0, 0.5, 3, work, hflat,
                    tail=((acc[0:1], src[0:1], 2.0), (acc[1:2], src[1:2], 0.5), (acc[2:3], src[2:3], 1.0)))
        assert work[:2].tolist() == [0.0, 0.0]       # left zero for the next step (no memset per step)
        assert acc.tolist() == [3.0 * (it + 1), -1.0 * (it + 1), 0.25 * (it + 1)]   # the bookkeeping sums of the last launch, skipped step or not
        if it != 4:
            assert torch.equal(hflat, pflat.half())  # the fp16 working copy written in the same pass
        assert float(scale) == scaler.get_scale(), (it, float(scale), scaler.get_scale())
        seen.append(float(scale))
        want = torch.cat([p.detach().reshape(-1) for p in ref])
        np.testing.assert_allclose(pflat.cpu(), want.cpu(), rtol=0, atol=3e-7, err_msg="step %d" % it)
        assert float(steps[0]) == float(opt.state[ref[0]]["step"]) and bool((steps == steps[0]).all())
    assert seen == [1024.0, 1024.0, 2048.0, 2048.0, 1024.0, 1024.0, 1024.0]  # grown after 3 clean steps, backed off by the inf step
    want_m = torch.cat([opt.state[p]["exp_avg"].reshape(-1) for p in ref])
    np.testing.assert_allclose(mflat.cpu(), want_m.cpu(), rtol=1e-5, atol=1e-8)
    # the 'legacy' schedule riding in the last launch: the lr moves AFTER the step that used it, by AdaptiveScheduler's rule
    lr2, klt = torch.tensor([3e-4], device=DEV), torch.zeros(1, device=DEV)
    for kl, want in ((0.02, 2e-4), (0.02, 2e-4 / 1.5), (0.001, 2e-4), (0.008, 2e-4), (0.0, 3e-4)):
        klt.fill_(kl)
        before = pflat.clone()
        F.adam_step(pflat, gflat, mflat, vflat, steps, lr2, (0.9, 0.999), 1e-8, 0.0, 1.0, None, None, 2.0, 0.5, 3, work, None,
                    adapt=(klt, 0.008, 1e-6, 1e-2))
        assert abs(float(lr2) - want) < 1e-9 and not torch.equal(before, pflat)


def test_fused_and_plain_optimiser_step_agree():
    """Two agents with identical weights and an identical rollout dataset: one optimiser step through the fused kernels and
    one through the plain torch formulation leave the same parameters (AMP on, as bez_kickPPO.yaml)."""
    from tests.test_gpu_round2 import _agent
    plain = _agent(512, 4096, fused_ops=False, hip_graphs=False)
    fused = _agent(512, 4096, fused_ops=True, hip_graphs=False)
    assert fused.fused and not plain.fused
    fused.model.load_state_dict(plain.model.state_dict())
    plain.obs = plain.env_reset()
    plain.play_steps()                      # one rollout + dataset on the plain agent ...
    fused._alloc_static()
    for k in plain.dataset:                 # ... shared with the fused one
        fused.dataset[k].copy_(plain.dataset[k])
    fused._mom_pack.copy_(plain._mom_pack)  # (the epoch's precomputed observation / value moments belong to the dataset)
    for rms_a, rms_b in ((plain.running_mean_std, fused.running_mean_std), (plain.value_mean_std, fused.value_mean_std)):
        rms_b.load_state_dict(rms_a.state_dict())
    kl_a, kl_b = torch.zeros((), device=DEV), torch.zeros((), device=DEV)
    la, lb = torch.zeros(2, device=DEV), torch.zeros(2, device=DEV)
    plain.calc_gradients(plain._minibatch(0), kl_a, la)
    fused.calc_gradients(fused._minibatch(0), kl_b, lb)
    assert abs(float(kl_a) - float(kl_b)) < 1e-5 + 1e-3 * abs(float(kl_a))
    np.testing.assert_allclose(lb.cpu(), la.cpu(), rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(fused.running_mean_std.running_mean.cpu(), plain.running_mean_std.running_mean.cpu(), rtol=1e-10, atol=1e-12)
    for (name, pa), pb in zip(plain.model.named_parameters(), fused.model.parameters()):
        # the first Adam step moves every weight by ~lr * sign(grad) = 3e-4: a gradient that is ~0 may round to either sign, so
        # all weights agree to 2 * lr and 99.9 % of them to a tenth of a step
        d = (pb.detach() - pa.detach()).abs().cpu().numpy()
        assert d.max() < 6.5e-4 and (d < 3e-5).mean() > 0.999, (name, d.max(), (d < 3e-5).mean())
    # and the fused rollout + whole epochs run (eager, then captured)
    fused2 = _agent(512, 4096, fused_ops=True)
    fused2.obs = fused2.env_reset()
    stats = [fused2.train_epoch() for _ in range(4)]
    assert fused2._g_rollout is not None and all(np.isfinite([s["kl"], s["a_loss"], s["c_loss"]]).all() for s in stats)
    assert len(fused2.game_rewards) > 0


def test_packed_weights_give_the_same_forward_and_backward_as_row_major():
    """The fragment-major weight copies (coalesced MFMA operand loads) feed the same products in the same order: forward outputs,
    kept activations, input gradients and bias gradients of the packed path equal the row-major path's bit for bit."""
    from bez_isaacgym_amd.ppo import fused as F
    from bez_isaacgym_amd.ppo.a2c_continuous import RunningMeanStd
    torch.manual_seed(17)
    n, d, units, a = 2048 + 77, 54, (400, 200, 100), 18
    dims = [d] + list(units)
    shapes = [(dims[i + 1], dims[i]) for i in range(3)] + [(a, dims[-1]), (1, dims[-1])]
    sizes = []
    for o, k in shapes:
        sizes += [o * k, o]
    hflat = (torch.randn(sum(sizes), device=DEV) * 0.08).half()
    views, layout, off = [], [], 0
    for o, k in shapes:
        w = hflat[off:off + o * k].view(o, k); layout.append((off, o, k)); off += o * k
        b = hflat[off:off + o]; off += o
        views.append((w, b))
    rms = RunningMeanStd((d,)).to(DEV)
    rms.running_mean.copy_(torch.randn(d, dtype=torch.float64) * 0.3); rms.running_var.copy_(torch.rand(d, dtype=torch.float64) + 0.5)
    packed = F.PackedWeights(hflat, layout, a)
    obs = torch.randn(n, d, device=DEV) * 1.5
    outs = []
    for pk in (None, packed):
        pf = F.PolicyForward(views[:3], views[3], views[4], rms, pk)
        x0 = torch.zeros(n, d, device=DEV, dtype=torch.float16)
        acts = [torch.zeros(n, w, device=DEV, dtype=torch.float16) for w in units]
        mu, v = torch.zeros(n, a, device=DEV), torch.zeros(n, 1, device=DEV)
        pf.train_forward(obs, x0, acts, mu, v)
        mu2, v2 = torch.zeros(n, a, device=DEV), torch.zeros(n, 1, device=DEV)
        pf(obs, mu2, v2)
        assert torch.equal(mu, mu2) and torch.equal(v, v2)
        pb = F.PolicyBackward(hflat, layout, a, pk)
        pb.refresh()
        gmu, gval = torch.randn(n, a, device=DEV) * 1e-2, torch.randn(n, 1, device=DEV) * 1e-2
        torch.manual_seed(5); gmu.normal_(0, 1e-2); gval.normal_(0, 1e-2)
        gz = [torch.zeros(n, w, device=DEV, dtype=torch.float16) for w in units]
        g16, v16 = torch.zeros(n, a, device=DEV, dtype=torch.float16), torch.zeros(n, 1, device=DEV, dtype=torch.float16)
        bg = [torch.zeros(w, device=DEV) for w in units]
        bm, bv = torch.zeros(a, device=DEV), torch.zeros(1, device=DEV)
        pb(gmu, gval, acts, gz, g16, v16, bg, bm, bv)
        outs.append([mu, v, x0] + acts + gz + [g16, v16] + bg)
    assert float(outs[0][0].abs().max()) > 0 and float(outs[0][6].abs().max()) > 0
    for t0, t1 in zip(outs[0], outs[1]):
        assert torch.equal(t0, t1)


def test_adaptive_lr_kernel_matches_the_scheduler_rule():
    from bez_isaacgym_amd.ppo import fused as F
    from bez_isaacgym_amd.ppo.a2c_continuous import AdaptiveScheduler
    sch = AdaptiveScheduler(0.008)
    for lr0, kl in ((3e-4, 0.02), (3e-4, 0.001), (3e-4, 0.008), (1e-6, 0.5), (1e-2, 0.0), (9e-3, 0.0039), (1.2e-6, 0.0161)):
        lr = torch.tensor([lr0], device=DEV)
        F.adaptive_lr(lr, torch.tensor([kl], device=DEV), sch.kl_threshold, sch.min_lr, sch.max_lr)
        np.testing.assert_allclose(float(lr), sch.update(lr0, kl), rtol=1e-6)


def test_gae_kernel_matches_discount_values():
    from bez_isaacgym_amd.ppo import fused as F
    from bez_isaacgym_amd.ppo.a2c_continuous import discount_values
    torch.manual_seed(6)
    for h, n in ((32, 4096), (5, 333)):
        rew, val = torch.randn(h, n, 1, device=DEV), torch.randn(h, n, 1, device=DEV)
        mbd, d = (torch.rand(h, n, device=DEV) < 0.1).float(), (torch.rand(n, device=DEV) < 0.1).float()
        last = torch.randn(n, 1, device=DEV)
        want = discount_values(0.99, 0.95, d, last, mbd, val, rew)
        advs, rets = torch.full((h, n, 1), 7.0, device=DEV), torch.full((h, n, 1), 7.0, device=DEV)
        F.gae(rew, val, mbd, d, last, 0.99, 0.95, advs, rets)
        np.testing.assert_allclose(advs.cpu(), want.cpu(), rtol=1e-5, atol=1e-5)   # same operations; the compiler contracts a*b+c (1.4e-6 over 32 steps)
        np.testing.assert_allclose(rets.cpu(), (want + val).cpu(), rtol=1e-5, atol=1e-5)
        # the bootstrap values as the network's normalised outputs, de-normalised inside the launch (RunningMeanStd(unnorm=True))
        from bez_isaacgym_amd.ppo.a2c_continuous import RunningMeanStd
        rms = RunningMeanStd((1,)).to(DEV)
        rms.running_mean.fill_(0.37); rms.running_var.fill_(2.5)
        raw = torch.randn(n, 1, device=DEV) * 3.0     # (some beyond the +-5 clamp)
        want2 = discount_values(0.99, 0.95, d, rms(raw, unnorm=True), mbd, val, rew)
        advs2 = torch.full((h, n, 1), 7.0, device=DEV)
        F.gae(rew, val, mbd, d, raw, 0.99, 0.95, advs2, None, unnorm=rms)
        np.testing.assert_allclose(advs2.cpu(), want2.cpu(), rtol=1e-5, atol=1e-5)


def test_head_grads_kernel():
    from bez_isaacgym_amd.ppo import fused as F
    torch.manual_seed(4)
    for b, a in ((32768, 18), (1000, 7)):
        gmu, gval = torch.randn(b, a, device=DEV) * 1e-3, torch.randn(b, 1, device=DEV) * 1e-3
        g16, v16 = torch.zeros(b, a, device=DEV, dtype=torch.float16), torch.zeros(b, 1, device=DEV, dtype=torch.float16)
        bm, bv = torch.full((a,), 0.5, device=DEV), torch.full((1,), -0.25, device=DEV)
        F.head_grads_f16(gmu, gval, g16, v16, bm, bv)
        assert torch.equal(g16, gmu.half()) and torch.equal(v16, gval.half())
        np.testing.assert_allclose(bm.cpu(), (0.5 + gmu.half().double().sum(0)).float().cpu(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(bv.cpu(), (-0.25 + gval.half().double().sum(0)).float().cpu(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("fused_backward", [False, True])
def test_train_forward_kernel_and_manual_backward_match_autograd_path(fused_backward):
    """A PPO minibatch's gradient through (a) the one-launch MFMA forward that keeps the ELU outputs + the backward chain called
    directly (its input-gradient half per layer, or as the one-launch MFMA backward kernel) and (b) the torch GEMM forward + autograd through _HalfLinearEluFn / _HalfLinearFn: same dataset, same weights.
    The forwards differ by fp16 roundings of the ELU (exp in fp32 either way) -- the flat gradients agree to that level."""
    from tests.test_gpu_round2 import _agent
    a = _agent(512, 4096, hip_graphs=False, fused_train_forward=False)
    b = _agent(512, 4096, hip_graphs=False, fused_train_forward=True, fused_policy_backward=fused_backward)
    assert (b._policy_bwd is not None) == fused_backward
    b.model.load_state_dict(a.model.state_dict())
    a.obs = a.env_reset()
    a.play_steps()
    b._alloc_static()
    for k in a.dataset:
        b.dataset[k].copy_(a.dataset[k])
    b._mom_pack.copy_(a._mom_pack)
    for ra, rb in ((a.running_mean_std, b.running_mean_std), (a.value_mean_std, b.value_mean_std)):
        rb.load_state_dict(ra.state_dict())
    if getattr(b, "_hflat", None) is not None:   # fp16 working copies follow the loaded masters
        b.model.a2c_network.refresh_half()
    grads = []
    for ag in (a, b):
        mb = ag._minibatch(0)
        ag._f_obs_rms.moments(mb["obs"], out=ag._obs_mom[0])   # (play_steps does this for every minibatch of the epoch)
        assert ag._train_fwd_ok(mb["obs"]) == (ag is b)
        ag._phase_b(mb)
        grads.append(ag._flat.detach().float().cpu().numpy().copy())
    ga, gb = grads
    scale = np.abs(ga).max()
    assert scale > 0 and np.isfinite(gb).all()
    np.testing.assert_allclose(gb, ga, atol=2e-3 * scale, rtol=2e-2)
    cos = float((ga * gb).sum() / np.sqrt((ga * ga).sum() * (gb * gb).sum()))
    assert cos > 0.9999, cos


def test_segmented_graphs_equal_the_monolithic_update():
    """World > 1 replays collective-free graph SEGMENTS with the RCCL calls in between.  Forced on one GPU (world = 2 pretended,
    no process group): from identical weights and an identical dataset the segmented update must leave the same parameters
    as the monolithic captured update."""
    from tests.test_gpu_round2 import _agent
    # constant learning rate: with the per-step adaptive rule a last-bit difference in a KL near a threshold (the loss kernel's sums are
    # float atomics) moves one agent's lr by 1.5x and the comparison would measure the schedule, not the graph segmentation
    mono, segm = _agent(512, 4096, lr_schedule="constant"), _agent(512, 4096, lr_schedule="constant")
    for ag in (mono, segm):
        ag.obs = ag.env_reset()
        for _ in range(2):
            ag.train_epoch()            # eager warm-up epochs (each agent draws its own exploration noise)
    import copy
    # deep copy: Optimizer.load_state_dict keeps same-device tensors BY REFERENCE, the two agents must not share Adam moments
    segm.set_full_state_weights(copy.deepcopy(mono.get_full_state_weights()))   # weights, normalisers, Adam moments, lr
    segm.scaler.load_state_dict(mono.scaler.state_dict())
    for a, b in zip(mono.model.parameters(), segm.model.parameters()):
        assert torch.equal(a, b)
    segm._segmented = True               # -> play_steps / run_update take the data-parallel (segmented) route
    for ag in (mono, segm):
        ag.play_steps()
    for k in mono.dataset:
        segm.dataset[k].copy_(mono.dataset[k])
    segm._mom_pack.copy_(mono._mom_pack)
    mono.run_update()                    # captures the whole update, then replays it: one real update
    segm.run_update()                    # captures the segments, then replays them: one real update
    torch.cuda.synchronize()
    assert segm._seg is not None and len(segm._seg["cb"]) == segm.num_minibatches and mono._g_update is not None
    # 20 optimiser steps on: the first mini-epoch's KL agrees to rounding, the later ones to the drift 20 Adam steps make of last-bit
    # differences in the atomically summed sigma / head-bias gradients (measured: <= 0.5 %)
    np.testing.assert_allclose(segm.kl_acc.cpu()[:1], mono.kl_acc.cpu()[:1], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(segm.kl_acc.cpu(), mono.kl_acc.cpu(), rtol=3e-2, atol=1e-5)
    for (name, pa), pb in zip(mono.model.named_parameters(), segm.model.parameters()):
        d = (pb.detach() - pa.detach()).abs().cpu().numpy()
        assert d.max() < 2e-3 and (d < 1e-4).mean() > 0.99, (name, d.max())
    segm.run_update()                    # replay works and stays finite
    assert all(torch.isfinite(p).all() for p in segm.model.parameters())


def test_data_parallel_path_with_real_rccl_calls_on_one_rank(tmp_path):
    """The data-parallel code path end to end on the 1-GPU box: a 1-rank RCCL ("nccl") process group, BEZ_PPO_FORCE_DIST=1, so
    every all-reduce is a real RCCL call issued between the graph replays; 5 epochs (2 eager + capture + replays)."""
    import os
    import subprocess
    import sys
    code = '''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import bez_isaacgym_amd   # before anything initialises HIP: graph replay is only safe with the runtime's packet capture off (DESIGN.md 6.2)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="%d", RANK="0", WORLD_SIZE="1", BEZ_PPO_FORCE_DIST="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from tests.test_gpu_round2 import _agent
a = _agent(512, 4096)
assert a._segmented and a.world == 1
a.obs = a.env_reset()
st = [a.train_epoch() for _ in range(5)]
assert a._seg is not None and a._g_rollout is not None
assert all(np.isfinite([s["kl"], s["a_loss"], s["c_loss"]]).all() for s in st), st
assert all(torch.isfinite(p).all() for p in a.model.parameters())
assert a._rccl is not None and a._rccl.world == 1          # the package's own communicator: ncclAllReduce on the training stream (ppo/rccl_direct.py)
seen, real = [], dist.all_reduce
dist.all_reduce = lambda t, *ar, **k: (seen.append(int(t.numel())), real(t, *ar, **k))[1]
del a.collective_sizes[:]; n0 = a._rccl.calls
a.train_epoch()
dist.all_reduce = real
calls = list(a.collective_sizes)
steps = a.mini_epochs * a.num_minibatches
assert len(calls) == steps + 2 and calls[2:] == [a._flat.numel()] * steps, calls   # SURVEY.md 5.8: ONE collective per optimiser step (+ 2 per epoch)
assert a._rccl.calls - n0 == steps + 2 and not seen, (a._rccl.calls - n0, seen)   # ... every one of them through the direct communicator, none through torch.distributed
# the direct all-reduce really sums on the stream it is given: a side stream's buffer, stream-ordered behind a fill
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    buf = torch.full((1000,), 3.0, device="cuda")
    a._rccl.all_reduce_(buf)
side.synchronize()
assert float(buf.sum()) == 3000.0
dist.destroy_process_group()
print("DP_OK", st[-1]["kl"])
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), __import__("tests.conftest", fromlist=["free_port"]).free_port())
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "DP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_training_is_bit_reproducible():
    """Same seed, same build, same GPU -> the same weights: two agents trained independently for 2 eager + 1 capturing + 2 replayed
    epochs end bit-identical (parameters, Adam moments, loss scale, normalisers).  What makes this hold: no float atomic on the path
    from the rollout to the weights -- loss sums, bias gradients, weight gradients and the gradient norm are fixed-order reductions."""
    from tests.test_gpu_round2 import _agent
    snaps = []
    for _ in range(2):
        torch.manual_seed(123); torch.cuda.manual_seed_all(123)
        a = _agent(512, 4096)
        a.obs = a.env_reset()
        for _e in range(5):
            a.train_epoch()
        torch.cuda.synchronize()
        assert a._g_update is not None
        snaps.append([p.detach().clone() for p in a.model.parameters()] + [a._mflat.clone(), a._vflat.clone(), a.scaler._scale.clone(),
                                                                          a.running_mean_std.running_mean.clone(), a.value_mean_std.running_var.clone()])
        del a
    assert all(torch.equal(x, y) for x, y in zip(*snaps)), [i for i, (x, y) in enumerate(zip(*snaps)) if not torch.equal(x, y)]


def test_two_ranks_on_the_gpu_stay_bit_identical(tmp_path):
    """World size 2 with the real HIP kernels: two processes share the 1-GPU box's card, each with its own shard of envs (global env
    ids rank * N + e), segmented HIP graphs, the step's one all-reduce between the replays.  RCCL refuses two ranks on one device, so
    the transport here is gloo (staged through the host when gloo has no device support); what is under test is everything around
    the collective: after 2 eager + 1 capturing + 2 replayed epochs both replicas hold bit-identical parameters, Adam moments,
    learning rate and loss scale, their rollouts differ (different env shards), and an epoch issues steps + 2 collectives."""
    import os
    import subprocess
    import sys
    code = '''
import os, sys, hashlib, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import bez_isaacgym_amd   # before anything initialises HIP (DESIGN.md 6.2)
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=2)
real = dist.all_reduce
calls = []
def staged(t, *a, **k):
    calls.append(int(t.numel()))
    if t.is_cuda:
        h = t.detach().cpu()
        real(h, *a, **k)
        t.copy_(h)
        return None
    return real(t, *a, **k)
dist.all_reduce = staged
real_bc = dist.broadcast
def staged_bc(t, src=0, *a, **k):
    if t.is_cuda:
        h = t.detach().cpu()
        real_bc(h, src, *a, **k)
        t.copy_(h)
        return None
    return real_bc(t, src, *a, **k)
dist.broadcast = staged_bc
from bez_isaacgym_amd.ppo.a2c_continuous import A2CAgent
from bez_isaacgym_amd.utils.config import load_config
from bez_isaacgym_amd.utils.rlgames_utils import RLGPUEnv, get_rlgames_env_creator
N = 512
cfg = load_config(["task=bez_kick", "num_envs=%%d" %% N, "headless=True"])
cfg["task"]["seed"] = 42
cfg["task"]["env_id_offset"] = rank * N          # what create_rlgpu_env records for a rank (rlgames_utils.py:71-81 + global env ids)
venv = RLGPUEnv("rlgpu", N, env_creator=get_rlgames_env_creator(cfg["task"], "bez_kick", "cuda:0", "cuda:0", 0, True))
params = cfg["train"]["params"]
params["config"].update(minibatch_size=4096, save_frequency=0, save_best_after=10 ** 9)
a = A2CAgent(params, venv, "cuda:0", rank=rank, world=2)
a.obs = a.env_reset()
st = [a.train_epoch() for _ in range(5)]
torch.cuda.synchronize()
assert a._seg is not None and a._g_rollout is not None, "segmented graphs were not captured"
assert all(np.isfinite([s["kl"], s["a_loss"], s["c_loss"]]).all() for s in st), st
calls.clear()
a.train_epoch()
steps = a.mini_epochs * a.num_minibatches
assert len(calls) == steps + 2 and calls[2:] == [a._flat.numel()] * steps, calls
def digest(ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()
mine = [digest(list(a.model.parameters())), digest([a._mflat, a._vflat, a._steps]), digest([a.lr_t.reshape(1), a.scaler._scale.reshape(1)]),
        digest([a.running_mean_std.running_mean, a.running_mean_std.running_var, a.value_mean_std.running_mean]),
        digest([a.mb["obs"]])]
both = [None, None]
dist.all_gather_object(both, mine)
assert both[0][:4] == both[1][:4], ("replicas diverged", both)
assert both[0][4] != both[1][4], "both ranks rolled out the same envs"
dist.barrier()
dist.destroy_process_group()
print("DP2_OK", rank, st[-1]["kl"])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(__import__("tests.conftest", fromlist=["free_port"]).free_port()), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=420))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (so, se) in enumerate(outs):
        assert "DP2_OK %d" % r in so, so[-2000:] + se[-3000:]


@pytest.mark.parametrize("rows,shapes", [
    (32768, [(400, 54), (200, 400), (100, 200), (18, 100), (1, 100)]),     # bez_kickPPO.yaml: 54-400-200-100 + the mu and value heads
    (4096, [(64, 32), (30, 416), (136, 72), (4, 50)]),                      # partial tiles, a block of 8 columns, 2- and 4-half load units
    (2048, [(416, 416)]),                                                  # the widest layer the policy kernels take
])
def test_wgrad_mfma_matches_fp32_reference(rows, shapes):
    """bez_ppo_wgrad_plan / _run (csrc/bez_wgrad.hip): dW_L += dY_L^T X_L for all layers in one split-K MFMA launch + one fixed-order
    reduction, against the fp32 product of the same fp16 operands; asymmetric integer-valued data first (any row / column or
    k-order mix-up of the transposed LDS reads shows as an exact mismatch), then random data; twice the same bits (deterministic)."""
    from bez_isaacgym_amd.ppo import fused as F
    g = torch.Generator(device=DEV); g.manual_seed(3)
    for mode in ("integer", "random"):
        dys, xs, grads, refs = [], [], [], []
        for (o, i) in shapes:
            if mode == "integer":   # small integers: every partial sum is exact in fp32, the result must match bit for bit
                dy = torch.randint(-2, 3, (rows, o), device=DEV, generator=g).to(torch.float16)
                x = (torch.randint(-2, 3, (rows, i), device=DEV, generator=g) + (torch.arange(i, device=DEV) % 3 == 0)).to(torch.float16)
            else:
                dy = (torch.randn(rows, o, device=DEV, generator=g) * 1e-2).to(torch.float16)
                x = torch.randn(rows, i, device=DEV, generator=g).to(torch.float16)
            # integer mode: the accumulator's start value is an integer too, so base + product is exact in fp32 whatever the order
            base = torch.randint(-8, 9, (o, i), device=DEV, generator=g).float() if mode == "integer" else torch.randn(o, i, device=DEV, generator=g)
            dys.append(dy); xs.append(x); grads.append(base.clone()); refs.append(base.double() + dy.double().t() @ x.double())
        wg = F.WgradMfma(dys, xs, grads)
        assert wg.ok and wg(accumulate=True)
        torch.cuda.synchronize()
        for gr, ref, (o, i) in zip(grads, refs, shapes):
            if mode == "integer":   # |sums| <= 32768 * 12 + 8 < 2^24: every partial sum is an exactly representable integer -> bit for bit
                assert torch.equal(gr.double(), ref), (o, i, float((gr.double() - ref).abs().max()))
            else:
                err = (gr.double() - ref).abs().max()
                assert float(err) < 2e-3 * float(ref.abs().max() + 1), (o, i, float(err))
        first = [gr.clone() for gr in grads]
        for gr, b in zip(grads, refs):
            gr.zero_()
        assert wg(accumulate=False)
        again = [gr.clone() for gr in grads]
        assert wg(accumulate=False)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(again, grads))    # fixed-order reduction: bit-identical run to run
    # what the kernel refuses (bez_ppo_wgrad_plan returns -3, the caller keeps its GEMM path): X rows of an odd number of halfs
    # (no aligned load unit), and a dY of odd width that needs more than one tile row (single-half loads are for a head's few columns)
    for (o, i) in ((8, 101), (45, 64)):
        odd = F.WgradMfma([torch.zeros(2048, o, device=DEV, dtype=torch.float16)], [torch.zeros(2048, i, device=DEV, dtype=torch.float16)],
                          [torch.zeros(o, i, device=DEV)])
        assert not odd.ok and odd(accumulate=False) is False
