"""PPO (rl_games `a2c_continuous` semantics) for the bez_kick env, in PyTorch-ROCm.

The reference trains through the un-vendored rl_games package (call sites bez_isaacgym/train.py:89-113; hyper-parameters
bez_isaacgym/cfg/train/bez_kickPPO.yaml:1-79).  This module is this build's consumer loop with the same semantics
(rl_games 1.1.x [ext]; its source is not under /root/reference, so the algorithm is restated from its published
behaviour and anchored on the YAML keys and on the shipped checkpoint's tensor names/shapes):

  network   actor_critic, separate: False -> shared MLP 54-400-200-100 (ELU), mu Linear(100,18) without activation,
            fixed sigma = nn.Parameter(zeros(18)) used as log-std, value Linear(100,1)        bez_kickPPO.yaml:10-27
  inputs    RunningMeanStd on observations (clamp +-5) and on values (normalize_value)         :51-52
  rollout   horizon_length 32, reward * scale_value 0.01, value bootstrap on time-outs        :53-56,72
  returns   GAE(gamma .99, tau .95)                                                           :58-59
  update    5 mini-epochs x (N*32 / minibatch_size) minibatches, PPO clip .2, clipped value loss, critic_coef 2 (x0.5),
            entropy_coef 0, bounds loss .001, advantage normalisation, grad-norm clip 1.0, AMP fp16 + GradScaler,
            Adam, adaptive LR on KL (threshold .008, x/ 1.5, clamp [1e-6, 1e-2])               :57-79
  multi-GPU one process per GPU; one fused all-reduce of the flat gradient (124 237 fp32) per optimiser step over
            RCCL (torch.distributed backend "nccl"), KL and running-moment sums averaged the same way; parameters
            broadcast from rank 0 at start (reference: Horovod, utils/rlgames_utils.py:71-81, config.yaml:40).
Checkpoints keep rl_games' key layout (model / running_mean_std / reward_mean_std / optimizer / epoch / frame /
last_mean_rewards) as evidenced by results/Bez_Kick/Normal/Bez_Kick_33.pth.
"""
import math
import os
import time

import torch
import torch.distributed as dist
import torch.nn as nn


# measurement only: BEZ_PPO_MEASURE_NO_COLLECTIVE=1 leaves the per-step all-reduce call out of the segmented update, to separate what
# the graph boundaries cost from what the collective call costs on the 1-GPU box.  Never set it in a real data-parallel run.
_MEASURE_NO_COLLECTIVE = os.environ.get("BEZ_PPO_MEASURE_NO_COLLECTIVE") == "1"


def _skip_collective_for_measurement():
    """the knob is honoured on a 1-rank group only: dropped on real ranks it would let the replicas diverge without an error (round-5 advisor finding)"""
    if not _MEASURE_NO_COLLECTIVE:
        return False
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        raise RuntimeError("BEZ_PPO_MEASURE_NO_COLLECTIVE=1 is a 1-rank measurement knob; unset it for a %d-rank job" % dist.get_world_size())
    return True


def _make_all_reduce(cfg, device):
    """The in-place sum over the ranks the data-parallel paths call.  With an RCCL process group on a GPU: a communicator of the package's own,
    whose ncclAllReduce is enqueued on the TRAINING stream (ppo/rccl_direct.py; `dp_direct_rccl: False` keeps torch.distributed) -- the
    all-reduce between two graph replays is then stream-ordered, without torch.distributed's two cross-stream event waits.  Otherwise
    (gloo on CPU, the world-2 tests): torch.distributed."""
    from . import rccl_direct
    if cfg.get("dp_direct_rccl", True) and rccl_direct.available(device):
        comm = rccl_direct.RcclComm(device)
        return (lambda t: comm.all_reduce_(t)), comm
    return (lambda t: dist.all_reduce(t)), None


def _dist_on():
    # BEZ_PPO_FORCE_DIST=1 (tests): treat a 1-rank process group as data parallel, so the real RCCL calls run on a 1-GPU box
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("BEZ_PPO_FORCE_DIST") == "1")


class RunningMeanStd(nn.Module):
    """rl_games RunningMeanStd: parallel-variance update of (mean, var, count); fp64 buffers like the checkpoint."""

    def __init__(self, shape, epsilon=1e-5):
        super().__init__()
        self.epsilon = epsilon
        self.register_buffer("running_mean", torch.zeros(shape, dtype=torch.float64))
        self.register_buffer("running_var", torch.ones(shape, dtype=torch.float64))
        self.register_buffer("count", torch.ones((), dtype=torch.float64))

    @torch.no_grad()
    def moments(self, x):
        """[column sums | column sums of squares | rows] in fp64: what an update needs to know about a batch.  Additive over
        ranks, so a data-parallel job all-reduces THESE (for all minibatches of an epoch at once) and not the statistics."""
        x = x.reshape(-1, *self.running_mean.shape).double()
        k = self.running_mean.numel()
        # NO torch.cat / torch.stack in anything a HIP graph may capture (see clip_grad_norm_capturable below): slice writes instead
        out = torch.empty(2 * k + 1, dtype=torch.float64, device=x.device)
        out[:k].copy_(x.sum(0).reshape(-1)); out[k:2 * k].copy_((x * x).sum(0).reshape(-1)); out[2 * k:].fill_(float(x.shape[0]))
        return out

    @torch.no_grad()
    def update_from_moments(self, flat):
        k = self.running_mean.numel()
        s1, s2, n = flat[:k].reshape(self.running_mean.shape), flat[k:2 * k].reshape(self.running_mean.shape), flat[2 * k]
        b_mean = s1 / n
        b_var = (s2 / n - b_mean * b_mean).clamp_min(0.0) * (n / (n - 1).clamp_min(1.0))  # unbiased, as torch.var
        delta = b_mean - self.running_mean
        tot = self.count + n
        self.running_mean += delta * n / tot
        m2 = self.running_var * self.count + b_var * n + delta * delta * self.count * n / tot
        self.running_var.copy_(m2 / tot)
        self.count.copy_(tot)

    @torch.no_grad()
    def update(self, x):
        flat = self.moments(x)
        if _dist_on():  # moments of the GLOBAL batch: every rank ends with identical statistics
            dist.all_reduce(flat)
        self.update_from_moments(flat)

    def forward(self, x, unnorm=False):
        if self.training and not unnorm:
            self.update(x)
        mean, var = self.running_mean.float(), self.running_var.float()
        if unnorm:
            y = torch.clamp(x, min=-5.0, max=5.0)
            return torch.sqrt(var + self.epsilon) * y + mean
        y = (x - mean) / torch.sqrt(var + self.epsilon)
        return torch.clamp(y, min=-5.0, max=5.0)


class _HalfLinearFn(torch.autograd.Function):
    """y = x @ W^T + b on explicit fp16 operands (the AMP arithmetic of nn.Linear under autocast), with the weight gradient
    computed split-K: for the PPO minibatch (32768 rows) dW = dY^T X has a tiny output (e.g. 400 x 54) and a huge reduction
    dimension, for which the GEMM library launches a handful of workgroups on 256 CUs (108 us measured per call); as a batched
    GEMM over `splits` row chunks plus one fp32 sum it fills the chip.  Gradients are returned in fp32 for the fp32 master
    parameters, exactly what autocast's cast nodes would hand back."""

    @staticmethod
    def forward(ctx, x, w32, b32, w16, b16, splits):
        ctx.save_for_backward(x, w16)
        ctx.splits = splits
        ctx.master = (w32, b32)
        return torch.addmm(b16, x, w16.t())

    @staticmethod
    def backward(ctx, gy):
        x, w16 = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy @ w16 if ctx.needs_input_grad[0] else None
        k, s = x.shape[0], ctx.splits
        split = s > 1 and k % s == 0 and k // s >= 64
        w32, b32 = ctx.master
        if split and gy.is_cuda and w32.grad is not None and b32.grad is not None and w32.grad.is_contiguous() and b32.grad.is_contiguous():
            # the agent's static flat gradient: reduce straight into the master .grad views (HIP kernels, csrc/bez_ppo.hip) and hand
            # autograd nothing to accumulate -- no fp32 copies of dY / of the partial products, no sum(0), no AccumulateGrad add
            from . import fused as F
            part = torch.bmm(gy.view(s, k // s, -1).transpose(1, 2), x.view(s, k // s, -1))
            F.wgrad_sum(part, w32.grad, accumulate=True)
            F.colsum_f16(gy, b32.grad, accumulate=True)
            return gx, None, None, None, None, None
        if split:
            gw = torch.bmm(gy.view(s, k // s, -1).transpose(1, 2), x.view(s, k // s, -1)).float().sum(0)
        else:
            gw = (gy.t() @ x).float()
        return gx, gw, gy.float().sum(0), None, None, None


class _HalfLinearEluFn(torch.autograd.Function):
    """elu(x @ W^T + b) of a hidden layer on explicit fp16 operands.  Forward = torch (GEMM with bias epilogue + ELU); backward
    = ONE HIP pass over dY for the ELU derivative and the bias gradient (csrc/bez_ppo.hip elu_bwd_colsum_kernel), then the same
    input-gradient GEMM and split-K weight gradient as _HalfLinearFn, reduced straight into the master .grad views."""

    @staticmethod
    def forward(ctx, x, w32, b32, w16, b16, splits):
        y = torch.nn.functional.elu(torch.addmm(b16, x, w16.t()))
        ctx.save_for_backward(x, w16, y)
        ctx.splits = splits
        ctx.master = (w32, b32)
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import fused as F
        x, w16, y = ctx.saved_tensors
        w32, b32 = ctx.master
        gz = torch.empty_like(y)
        F.elu_bwd_colsum_f16(gy.contiguous(), y, gz, b32.grad, accumulate=True)
        gx = gz @ w16 if ctx.needs_input_grad[0] else None
        k, s = x.shape[0], ctx.splits
        part = torch.bmm(gz.view(s, k // s, -1).transpose(1, 2), x.view(s, k // s, -1))
        F.wgrad_sum(part, w32.grad, accumulate=True)
        return gx, None, None, None, None, None

    @staticmethod
    def usable(x, w32, b32, splits):
        k = x.shape[0]
        return (x.is_cuda and splits > 1 and k % splits == 0 and k // splits >= 64 and w32.grad is not None and b32.grad is not None
                and w32.grad.is_contiguous() and b32.grad.is_contiguous() and torch.is_grad_enabled())


class A2CNetwork(nn.Module):
    """Parameter names match rl_games' a2c_network.* so state dicts interoperate."""

    def __init__(self, obs_dim, act_dim, units=(400, 200, 100)):
        super().__init__()
        layers, d = [], obs_dim
        for u in units:
            layers += [nn.Linear(d, u), nn.ELU()]
            d = u
        self.actor_mlp = nn.Sequential(*layers)
        self.value = nn.Linear(d, 1)
        self.mu = nn.Linear(d, act_dim)
        self.sigma = nn.Parameter(torch.zeros(act_dim))  # const_initializer val 0, fixed_sigma True

    # ---- explicit-fp16 fast path (GPU, mixed precision): fp16 copies of the fp32 master weights refreshed once per optimiser
    # step by one multi-tensor copy instead of autocast's per-use casts; split-K weight gradients (_HalfLinearFn)
    def enable_half_path(self, splits=64):
        self._lin = [m for m in self.actor_mlp if isinstance(m, nn.Linear)] + [self.mu, self.value]
        self._p32 = [p for m in self._lin for p in (m.weight, m.bias)]
        self._p16 = [p.detach().half() for p in self._p32]
        self._splits = int(splits)

    def refresh_half(self):
        torch._foreach_copy_(self._p16, [p.detach() for p in self._p32])

    def _half_linear(self, i, x):
        m = self._lin[i]
        return _HalfLinearFn.apply(x, m.weight, m.bias, self._p16[2 * i], self._p16[2 * i + 1], self._splits)

    def forward(self, obs):
        if obs.dtype == torch.float16 and getattr(self, "_p16", None) is not None:
            h = obs
            n_hidden = len(self._lin) - 2
            for i in range(n_hidden):
                m = self._lin[i]
                if _HalfLinearEluFn.usable(h, m.weight, m.bias, self._splits):
                    h = _HalfLinearEluFn.apply(h, m.weight, m.bias, self._p16[2 * i], self._p16[2 * i + 1], self._splits)
                else:
                    h = torch.nn.functional.elu(self._half_linear(i, h))
            return self._half_linear(n_hidden, h), self.sigma.unsqueeze(0).expand(obs.shape[0], -1), self._half_linear(n_hidden + 1, h)
        h = self.actor_mlp(obs)
        return self.mu(h), self.sigma.unsqueeze(0).expand(obs.shape[0], -1), self.value(h)


class ModelA2CContinuousLogStd(nn.Module):
    def __init__(self, obs_dim, act_dim, units):
        super().__init__()
        self.a2c_network = A2CNetwork(obs_dim, act_dim, units)

    @staticmethod
    def neglogp(x, mean, std, logstd):
        return 0.5 * (((x - mean) / std) ** 2).sum(-1) + 0.5 * math.log(2.0 * math.pi) * x.shape[-1] + logstd.sum(-1)

    def forward(self, obs, prev_actions=None):
        mu, logstd, value = self.a2c_network(obs)
        sigma = torch.exp(logstd)
        if prev_actions is not None:  # training pass
            entropy = (0.5 + 0.5 * math.log(2 * math.pi) + logstd).sum(-1)
            return dict(prev_neglogp=self.neglogp(prev_actions, mu, sigma, logstd), values=value, entropy=entropy,
                        mus=mu, sigmas=sigma)
        action = mu + sigma * torch.randn_like(mu)
        return dict(neglogpacs=self.neglogp(action, mu, sigma, logstd), values=value, actions=action, mus=mu, sigmas=sigma)


def policy_kl(p0_mu, p0_sigma, p1_mu, p1_sigma):
    c1 = torch.log(p1_sigma / p0_sigma + 1e-5)
    c2 = (p0_sigma ** 2 + (p1_mu - p0_mu) ** 2) / (2.0 * (p1_sigma ** 2 + 1e-5))
    return (c1 + c2 - 0.5).sum(-1).mean()


@torch.no_grad()
def clip_grad_norm_capturable(parameters, max_norm):
    """torch.nn.utils.clip_grad_norm_ (2-norm) without torch.stack: the squared norms are added with plain kernels instead of being
    concatenated first (11 tiny launches instead of a cat + a norm; on ROCm cat stages its input pointers through a pinned host buffer,
    one more thing a captured graph would have to keep alive).  NOT the cause of round 3's graph + AMP + save defect -- that was the HIP
    runtime's graph packet capture (bez_isaacgym_amd/__init__.py); both clips behave the same in profiles/r04_plain_graph_probe.txt."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.zeros(())
    norms = torch._foreach_norm(grads, 2.0)
    total_sq = norms[0] * norms[0]
    for nrm in norms[1:]:
        total_sq = torch.addcmul(total_sq, nrm, nrm)
    total = total_sq.sqrt()
    coef = torch.clamp(float(max_norm) / (total + 1e-6), max=1.0)
    torch._foreach_mul_(grads, coef)
    return total


class AdaptiveScheduler:
    """rl_games AdaptiveScheduler; `update_` is the same rule on a device-resident lr tensor (no host sync)."""

    def __init__(self, kl_threshold=0.008):
        self.min_lr, self.max_lr, self.kl_threshold = 1e-6, 1e-2, kl_threshold

    def update(self, lr, kl):
        if kl > 2.0 * self.kl_threshold:
            lr = max(lr / 1.5, self.min_lr)
        if kl < 0.5 * self.kl_threshold:
            lr = min(lr * 1.5, self.max_lr)
        return lr

    @torch.no_grad()
    def update_(self, lr_t, kl_t, scale=1.0):
        """scale: kl_t holds scale x the KL (a sum over rows instead of their mean): the thresholds are scaled instead."""
        thr = self.kl_threshold * scale
        if isinstance(lr_t, torch.Tensor) and lr_t.is_cuda and lr_t.dtype == torch.float32 and kl_t.dtype == torch.float32 and lr_t.is_contiguous():
            from . import fused as F   # one launch instead of ten elementwise ones
            return F.adaptive_lr(lr_t, kl_t, thr, self.min_lr, self.max_lr)
        cur = lr_t.value if hasattr(lr_t, "value") else lr_t
        kl_t = kl_t.reshape(())
        down = torch.clamp(cur / 1.5, min=self.min_lr)
        lr1 = torch.where(kl_t > 2.0 * thr, down, cur)
        up = torch.clamp(lr1 * 1.5, max=self.max_lr)
        lr_t.copy_(torch.where(kl_t < 0.5 * thr, up, lr1))


def discount_values(gamma, tau, fdones, last_values, mb_fdones, mb_values, mb_rewards):
    """GAE backward scan over the horizon (rl_games a2c_common.discount_values)."""
    horizon = mb_rewards.shape[0]
    advs = torch.zeros_like(mb_rewards)
    lastgaelam = torch.zeros_like(last_values)
    for t in reversed(range(horizon)):
        if t == horizon - 1:
            nextnonterminal, nextvalues = 1.0 - fdones, last_values
        else:
            nextnonterminal, nextvalues = 1.0 - mb_fdones[t + 1], mb_values[t + 1]
        nextnonterminal = nextnonterminal.unsqueeze(1)
        delta = mb_rewards[t] + gamma * nextvalues * nextnonterminal - mb_values[t]
        lastgaelam = delta + gamma * tau * nextnonterminal * lastgaelam
        advs[t] = lastgaelam
    return advs


def swap_and_flatten01(x):
    s = x.shape
    return x.transpose(0, 1).reshape(s[0] * s[1], *s[2:])


class _CpuLr:
    """CPU fallback of the device-resident lr tensor (tests): same interface, writes through to the param groups."""

    def __init__(self, optimizer, lr):
        self.optimizer, self.value = optimizer, torch.tensor(float(lr))

    def copy_(self, t):
        self.value = t.detach().clone().reshape(())
        for g in self.optimizer.param_groups:
            g["lr"] = float(self.value)

    def item(self):
        return float(self.value)


class A2CAgent:
    """Continuous-action PPO agent.  `vec_env` needs step(actions)->(obs_dict, rew, done, info), reset()->obs_dict."""

    def __init__(self, params, vec_env, device, writer=None, rank=0, world=1):
        c = params["config"]
        self.cfg, self.params = c, params
        self.vec_env, self.device = vec_env, torch.device(device)
        self.rank, self.world = rank, world
        # data parallel (more than one rank, or BEZ_PPO_FORCE_DIST=1 on a 1-rank group: the same code path with real RCCL calls on a
        # 1-GPU box -- tests, `bench.py --dp-path`): HIP graphs are captured in segments with the collectives between the replays
        # `dp_capture_collectives: True` (opt-in, EXPERIMENTAL): capture the RCCL calls into the same graphs as everything else -- the single-GPU
        # path's two replays per epoch, collectives included.  torch.distributed supports capturing NCCL / RCCL work; exercised here on a
        # 1-rank group only (a 1-GPU box cannot hold two RCCL ranks), hence off by default
        self._segmented = bool((world > 1 or _dist_on()) and not c.get("dp_capture_collectives", False))
        self.collective_sizes = []   # element count of every data-parallel sum this agent issued (tests: `steps + 2` per epoch), whatever the transport
        if _dist_on():
            fn, self._rccl = _make_all_reduce(c, self.device)
            self._all_reduce = lambda t: (self.collective_sizes.append(int(t.numel())), fn(t))[1]
        else:
            self._all_reduce, self._rccl = (lambda t: t), None
        self.num_actors = int(c["num_actors"])
        self.horizon = int(c["horizon_length"])
        self.gamma, self.tau = float(c["gamma"]), float(c["tau"])
        self.e_clip = float(c["e_clip"])
        self.critic_coef, self.entropy_coef = float(c["critic_coef"]), float(c["entropy_coef"])
        self.bounds_loss_coef = float(c.get("bounds_loss_coef", 0.0) or 0.0)
        self.grad_norm = float(c["grad_norm"])
        self.truncate_grads = bool(c.get("truncate_grads", False))
        self.clip_value = bool(c.get("clip_value", True))
        self.normalize_input = bool(c.get("normalize_input", False))
        self.normalize_value = bool(c.get("normalize_value", False))
        self.normalize_advantage = bool(c.get("normalize_advantage", True))
        self.value_bootstrap = bool(c.get("value_bootstrap", False))
        self.mixed_precision = bool(c.get("mixed_precision", False)) and self.device.type == "cuda"
        self.reward_scale = float(c.get("reward_shaper", {}).get("scale_value", 1.0))
        self.mini_epochs = int(c["mini_epochs"])
        self.batch_size = self.horizon * self.num_actors
        self.minibatch_size = min(int(c["minibatch_size"]), self.batch_size)
        assert self.batch_size % self.minibatch_size == 0, "batch must be a multiple of minibatch_size"
        self.num_minibatches = self.batch_size // self.minibatch_size
        self.max_epochs = int(c.get("max_epochs", 100000))
        self.save_frequency = int(c.get("save_frequency", 0) or 0)
        self.save_best_after = int(c.get("save_best_after", 100))
        self.score_to_win = float(c.get("score_to_win", float("inf")))
        self.last_lr = float(c["learning_rate"])
        self.is_adaptive_lr = c.get("lr_schedule") == "adaptive"
        self.scheduler = AdaptiveScheduler(float(c.get("kl_threshold", 0.008)))
        # rl_games 1.1.3 (setup.py:22) [ext]: config.get('schedule_type', 'legacy') -- 'legacy' moves the learning rate after EVERY
        # minibatch step on that step's KL, 'standard' once per mini-epoch on the mean; and PPODataset.update_mu_sigma stores the
        # policy's current mu / sigma over the minibatch's old ones after every step, so from the second mini-epoch on the KL is
        # measured against the previous pass, not against the rollout policy.  bez_kickPPO.yaml sets neither key.
        self.schedule_type = str(c.get("schedule_type", "legacy"))
        self.update_mu_sigma = bool(c.get("update_mu_sigma", True))
        self.name = c.get("full_experiment_name") or c.get("name", "bez_kick")
        self.writer = writer
        net = params["network"]
        assert net["name"] == "actor_critic" and not net.get("separate", False)
        units = tuple(net["mlp"]["units"])
        assert str(net["mlp"]["activation"]).lower() == "elu"
        obs_dim, act_dim = 54, 18
        info = getattr(vec_env, "get_env_info", None)
        if info is not None:
            ei = vec_env.get_env_info()
            obs_dim, act_dim = ei["observation_space"].shape[0], ei["action_space"].shape[0]
        self.obs_dim, self.act_dim = obs_dim, act_dim
        seed = params.get("seed", 42)
        torch.manual_seed(int(seed if seed not in ("", None) else 42))
        self.model = ModelA2CContinuousLogStd(obs_dim, act_dim, units).to(self.device)
        # (anything that loads weights into the model marks the derived fp16 / fragment-major copies stale: _refresh_weight_copies_if_dirty)
        self.model.register_load_state_dict_post_hook(lambda module, incompatible_keys: self._mark_weights_dirty())
        self.running_mean_std = RunningMeanStd((obs_dim,)).to(self.device) if self.normalize_input else None
        self.value_mean_std = RunningMeanStd((1,)).to(self.device) if self.normalize_value else None
        on_gpu = self.device.type == "cuda"
        # lr lives on the device so that the adaptive-KL rule needs no host sync and the update can be graph-captured
        # One 512-byte device buffer holds everything the host reads back per epoch -- episode statistics (3 fp64), learning rate, loss and KL
        # accumulators (fp32 views) -- so an epoch ends with ONE device-to-host copy instead of four reads with a host round trip each
        # (~30 us of idle GPU apiece between the replayed graphs)
        self._report = torch.zeros(64, device=self.device, dtype=torch.float64) if on_gpu else None
        if on_gpu:
            self.lr_t = self._report.view(torch.float32)[8]
            self.lr_t.fill_(self.last_lr)
        else:
            self.lr_t = torch.tensor(self.last_lr, device=self.device, dtype=torch.float32)
        # fused + capturable Adam: takes GradScaler's found_inf / scale as tensors (no .item()), so scaler.step() is graph-safe
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=self.lr_t if on_gpu else self.last_lr, eps=1e-8,
                                          capturable=on_gpu, fused=on_gpu)
        if not on_gpu:
            self.lr_t = _CpuLr(self.optimizer, self.last_lr)
        self.scaler = torch.amp.GradScaler("cuda", enabled=self.mixed_precision)
        g = c.get("hip_graphs", "auto")
        # an env whose step() syncs with the host or allocates (domain randomisation: vec_task.py:505-725) cannot be captured
        env_graph_safe = bool(getattr(getattr(vec_env, "env", vec_env), "graph_safe", True))
        # this trainer reads obs / reward / dones only: let the simulator skip the Isaac-visible extras (contact rows, feet, prev_lin_vel)
        # -- for as long as this agent trains: train() hands the env back with every tensor current (release_env)
        self._lean_env = getattr(vec_env, "env", vec_env) if c.get("lean_env_step", True) and hasattr(getattr(vec_env, "env", vec_env), "set_lean") else None
        if self._lean_env is not None:
            self._lean_env.set_lean(True)
        # world > 1: graphs are captured in SEGMENTS that contain no collective (the RCCL calls run eagerly between replays),
        # which needs the fused path's static flat gradient buffer -> decided below, once `self.fused` is known
        self.use_graphs = bool(on_gpu and env_graph_safe and (g is True or g == "auto"))
        self.graph_warmup_epochs = 2
        self._eager_epochs = 0  # epochs run eagerly IN THIS PROCESS (a restored epoch_num says nothing about warm-up)
        self._g_rollout = self._g_update = self._pool = None
        self.mb = None
        self._ep_hist = []
        self.epoch_num, self.frame = 0, 0
        self.games_to_track = 100
        self.game_rewards, self.game_lengths = [], []
        self.last_mean_rewards = -100500.0
        self.current_rewards = torch.zeros(self.num_actors, device=self.device)
        self.current_lengths = torch.zeros(self.num_actors, device=self.device)
        self.dones = torch.ones(self.num_actors, dtype=torch.float32, device=self.device)
        self._flat_grad = None
        # HIP glue kernels (csrc/bez_ppo.hip) for everything around the MLP: on by default on a GPU, `fused_ops: False` keeps
        # the plain torch formulation (the one the CPU path runs and the kernels are tested against)
        self._fused_opt = False
        self._policy_fwd = None
        self._policy_bwd = None
        self._packed = None
        self._packed_stale = True     # the fragment-major weight copies need a refresh() before their next use (set whenever weights change outside the fused optimiser)
        self._weights_dirty = True    # weights were written from outside the optimiser since the derived copies were last refreshed
        self._rms_preapplied = False  # the input normaliser already holds the coming minibatch's moments (folded into the previous optimiser launch)
        self.fused = bool(on_gpu and c.get("fused_ops", True))
        # Replaying ANY graph is only safe with the HIP runtime's graph packet capture off (bez_isaacgym_amd/__init__.py: root cause of
        # round 3's "plain AMP path stops learning once agent.save() runs between replays", profiles/r04_plain_graph_probe.txt).  The
        # package switches it off at import; if the process had initialised HIP before, or the user switched it back on, say so -- and
        # keep the plain torch path (thousands of nodes, clobbered after ~8 KB of eager kernel arguments) eager unless forced.
        from .. import GRAPH_REPLAY_SAFE
        if self.use_graphs and not GRAPH_REPLAY_SAFE:
            # a hard condition for EVERY path (round-4 advisor finding: the fused path only warned and kept replaying): without the
            # guarantee `hip_graphs: auto` means eager; only an explicit `hip_graphs: True` replays, at the caller's risk and with a warning
            import warnings
            if g is True:
                warnings.warn("HIP graphs are replayed with the runtime's graph packet capture ON (DEBUG_CLR_GRAPH_PACKET_CAPTURE != 0, or HIP "
                              "was initialised before bez_isaacgym_amd was imported): enough eager kernel launches between replays corrupt the "
                              "replayed kernels' arguments (DESIGN.md 6.2)", RuntimeWarning, stacklevel=2)
            else:
                warnings.warn("HIP graphs are OFF for this agent: the runtime's graph packet capture could not be switched off (HIP was initialised "
                              "before bez_isaacgym_amd was imported, or DEBUG_CLR_GRAPH_PACKET_CAPTURE is set to non-zero), and replaying graphs "
                              "in that state corrupts kernel arguments (DESIGN.md 6.2).  Import bez_isaacgym_amd before the first CUDA call to "
                              "get the replayed (faster) epochs", RuntimeWarning, stacklevel=2)
                self.use_graphs = False
        if self.fused:
            from . import fused as F
            self._F = F
            red = self._all_reduce if _dist_on() else None
            self._f_obs_rms = F.FusedRunningMeanStd(self.running_mean_std, red) if self.normalize_input else None
            self._f_val_rms = F.FusedRunningMeanStd(self.value_mean_std, red) if self.normalize_value else None
            self.half_path = bool(self.mixed_precision and self.normalize_input and c.get("half_path", True))
            if self.half_path:
                self.model.a2c_network.enable_half_path(int(c.get("wgrad_splits", 32)))
            self._bind_flat_grads()
            self._fused_opt = bool(c.get("fused_optimizer", True))
            if self._fused_opt:
                self._bind_flat_optimizer()
            # the rollout's policy forward as one MFMA kernel on the fp16 working weights (csrc/bez_policy.hip); training's forward /
            # backward stay torch GEMMs
            self._policy_fwd = None
            net = self.model.a2c_network
            if self.half_path and c.get("fused_policy_forward", True) and getattr(net, "_p16", None) is not None:
                nh = len(net._lin) - 2
                wb = [(net._p16[2 * i], net._p16[2 * i + 1]) for i in range(nh + 2)]
                if all(w.shape[0] <= 416 for w, _ in wb[:nh]) and obs_dim <= 416 and act_dim <= 31 and nh <= 6:
                    # fragment-major copies of the weights (coalesced MFMA operand loads), refreshed by one scatter of the fp16 working copy
                    hflat = getattr(self, "_hflat", None)
                    layout = None if hflat is None else [((w.data_ptr() - hflat.data_ptr()) // 2, w.shape[0], w.shape[1]) for w, _ in wb]
                    self._packed = F.PackedWeights(hflat, layout, act_dim) if (hflat is not None and c.get("packed_weights", True)) else None
                    self._policy_fwd = F.PolicyForward(wb[:nh], wb[nh], wb[nh + 1], self.running_mean_std if self.normalize_input else None, self._packed)
                    # ... and the input-gradient chain of the minibatch backward pass as one kernel on transposed copies
                    if (hflat is not None and c.get("fused_policy_backward", True) and all(32 <= w.shape[0] <= 416 and w.shape[0] % 2 == 0 for w, _ in wb[:nh])):
                        self._policy_bwd = F.PolicyBackward(hflat, layout, act_dim, self._packed)
        if (self._segmented or world > 1 or _dist_on()) and not self.fused:   # (also with dp_capture_collectives: the plain torch path is never captured across ranks)
            self.use_graphs = False  # the plain torch path has collectives in the middle of autograd-heavy code: eager only
        self._seg = None  # segmented graphs of the data-parallel update
        if _dist_on():  # identical replicas (hvd.broadcast_parameters equivalent)
            for p in self.model.parameters():
                dist.broadcast(p.data, src=0)
        self.obs = None

    # ------------------------------------------------------------------ rollout
    def _preproc_obs(self, obs):
        return self.running_mean_std(obs) if self.normalize_input else obs

    @torch.no_grad()
    def get_action_values(self, obs):
        self.model.eval()
        if self.normalize_input:
            self.running_mean_std.eval()
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.mixed_precision):
            res = self.model(self._preproc_obs(obs))
        res = {k: v.float() for k, v in res.items()}
        if self.normalize_value:
            self.value_mean_std.eval()
            res["values"] = self.value_mean_std(res["values"], True)
        return res

    @torch.no_grad()
    def get_values(self, obs):
        return self.get_action_values(obs)["values"]

    def env_reset(self):
        obs = self.vec_env.reset()["obs"].to(self.device)
        if self.obs is None or self.obs.shape != obs.shape:
            self.obs = obs.clone()
        else:
            self.obs.copy_(obs)
        return self.obs

    def _alloc_static(self):
        """Rollout / dataset storage allocated ONCE: fixed addresses are what lets the rollout and the minibatch update be
        captured as HIP graphs and replayed without a host round trip per op."""
        H, N, dev = self.horizon, self.num_actors, self.device
        z = lambda *s: torch.zeros(*s, device=dev)
        self.mb = dict(obs=z(H, N, self.obs_dim), act=z(H, N, self.act_dim), mu=z(H, N, self.act_dim), sigma=z(H, N, self.act_dim),
                       val=z(H, N, 1), rew=z(H, N, 1), neglogp=z(H, N), dones=z(H, N))
        B = self.batch_size
        self.dataset = dict(old_values=z(B, 1), old_logp=z(B), advantages=z(B), returns=z(B, 1), actions=z(B, self.act_dim),
                            obs=z(B, self.obs_dim), mu=z(B, self.act_dim), sigma=z(B, self.act_dim))
        if self.fused and self._policy_fwd is not None and self.cfg.get("rollout_into_dataset", True):
            # the rollout rows of obs / actions / mu / sigma / neglogp ARE the dataset's rows: rl_games flattens env-major (row of env e at
            # step n = e * H + n), so the (H, N, ...) rollout tensors become strided views of the (N * H, ...) dataset tensors and the
            # policy launch writes through them (BezPpoRolloutLayout) -- five transposing copies per epoch (28 MB for the observations)
            # and their temporaries are gone
            ds = self.dataset
            for k_mb, k_ds in (("obs", "obs"), ("act", "actions"), ("mu", "mu"), ("sigma", "sigma")):
                self.mb[k_mb] = ds[k_ds].view(N, H, -1).transpose(0, 1)
            self.mb["neglogp"] = ds["old_logp"].view(N, H).transpose(0, 1)
        # episode statistics accumulated on the device (no .nonzero()/.tolist() inside the rollout)
        if self._report is not None and 12 + self.mini_epochs <= 2 * self._report.numel():   # views of the epoch report (see __init__)
            f32 = self._report.view(torch.float32)
            self.ep_stats, self.loss_acc, self.kl_acc = self._report[0:3], f32[10:12], f32[12:12 + self.mini_epochs]
        else:
            self.ep_stats = torch.zeros(3, device=dev, dtype=torch.float64)  # [finished episodes, sum of returns, sum of lengths]
            self.kl_acc = torch.zeros(self.mini_epochs, device=dev)
            self.loss_acc = torch.zeros(2, device=dev)
        # the epoch's data-only moments in ONE fp64 buffer = one all-reduce per epoch when data parallel: the
        # observation moments of every minibatch (rl_games updates the input normaliser at every minibatch forward, and what
        # it adds depends on the dataset alone, not on the weights), then the moments of the values and of the returns
        w = 2 * self.obs_dim + 1
        self._mom_pack = torch.zeros(self.num_minibatches * w + 6, device=dev, dtype=torch.float64)
        self._obs_mom = self._mom_pack[:self.num_minibatches * w].view(self.num_minibatches, w)
        self._val_mom, self._ret_mom = self._mom_pack[-6:-3], self._mom_pack[-3:]
        self._adv_pack = torch.zeros(6, device=dev, dtype=torch.float64)  # second (and last) epoch collective: advantage moments + episode statistics
        if self.fused:
            hd = torch.float16 if self.mixed_precision else torch.float32
            A, D, MB = self.act_dim, self.obs_dim, self.minibatch_size
            self._fx = dict(obs_n=torch.zeros(N, D, device=dev, dtype=hd), env_act=z(N, A), noise=z(H, N, A),
                            mb_obs_n=torch.zeros(MB, D, device=dev, dtype=hd), gmu=z(MB, A), gval=z(MB, 1), glog=z(A), stats=z(5),
                            last_mu=z(N, A), last_v=z(N, 1), advs=z(H, N, 1), rets=z(H, N, 1), val_n=z(H * N, 1), ret_n=z(H * N, 1),
                            loss_scratch=self._F.loss_scratch(MB, A, dev))   # fixed-order sums in the loss kernel: bit-reproducible steps
            if self.cfg.get("episode_sum_slots", True):   # per-workgroup slots of the folded bookkeeping's episode sums (RolloutPost.ep_parts)
                self._ep_parts = torch.zeros(self._F.RolloutPost.parts_numel(N), device=dev, dtype=torch.float64)

    @torch.no_grad()
    def _rollout_steps_fused(self):
        """The horizon loop on HIP kernels: policy forward + sampling + neglogp + clamp + rollout rows (1 launch, bez_policy.hip;
        without it: normalise, torch MLP, rollout_pre), env step (1 launch), reward shaping + episode statistics (1 launch)."""
        mb, F, fx = self.mb, self._F, self._fx
        net = self.model.a2c_network
        self.model.eval()
        if not self._copies_kept_current():
            # (with the fused optimiser the Adam launch writes the fp16 working copy and the fragment-major copies at every step, and weights
            # changed from outside -- checkpoint restore, load_state_dict, the parameter broadcast -- are caught eagerly in play_steps: the
            # 26 us multi-tensor copy and the scatter at the head of every rollout were refreshing what was already current)
            if self.half_path:
                net.refresh_half()
            if self._packed is not None:
                self._packed.refresh()  # once per epoch, whoever changed the weights last
                self._packed_stale = False
        cur = self.obs  # step 0 reads the agent's copy; later steps read the env's own observation buffer (no per-step copy)
        vrms = self.value_mean_std if self.normalize_value else None
        if not torch.cuda.is_current_stream_capturing():
            fx["noise"].normal_()  # the whole horizon's action noise in one launch (replayed rollouts: drawn by play_steps in front of the replay)
        boot = self.value_bootstrap
        pending = None  # rollout_post arguments of the env step whose bookkeeping has not run yet
        # finished-episode sums of the folded bookkeeping: per-workgroup slots instead of 128 workgroups' fp64 atomics on one cache line per step
        # (1.7 us of a 14.8 us launch), folded into ep_stats once, behind the loop
        parts = getattr(self, "_ep_parts", None) if self._policy_fwd is not None else None   # (allocated with the rollout buffers: _alloc_static)
        fold = self._policy_fwd is not None and self.cfg.get("fold_rollout_post", True)
        # a domain-randomised env at full speed (BASELINE config 5): (a) its action-noise lambda is added by the policy launch itself
        # (the same bits: bez_sim_action_noise_source); (b) the randomisation of the coming env step runs as ONE EXTRA WORKGROUP of the
        # policy launch (bez_sim_dr_step_args: it touches nothing the forward pass reads) instead of a 6-7 us launch of its own in front of
        # the step; (c) `dr_prelaunch: True` would instead launch that kernel early on a side stream -- kept as an option, OFF: in the
        # replayed HIP graph each fork / join pair costs more (~14 us) than the kernel it hides (6.12 vs 5.68 ms per epoch)
        env = getattr(self.vec_env, "env", self.vec_env)
        act_noise = None
        if self._policy_fwd is not None and self.cfg.get("fold_action_noise", True) and hasattr(env, "action_noise_source"):
            src = env.action_noise_source()
            act_noise = None if src is None else F.ActionNoise(*src)
        fold_dr = bool(self._policy_fwd is not None and self.cfg.get("fold_dr_step", True) and hasattr(env, "dr_step_args") and getattr(env, "randomize", False)
                       and not self.cfg.get("dr_prelaunch", False))
        prelaunch = bool(self.cfg.get("dr_prelaunch", False) and hasattr(env, "dr_prelaunch") and getattr(env, "randomize", False)
                         and not getattr(env, "first_randomization", True))
        if prelaunch and getattr(self, "_side_stream", None) is None:
            assert not torch.cuda.is_current_stream_capturing()
            self._side_stream = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device) if prelaunch else None
        forked = False
        if act_noise is not None:
            env.external_action_noise = True
        try:
            for n in range(self.horizon):
                if self._policy_fwd is not None:
                    # normaliser + 5 Linear + 3 ELU + sampling + neglogp + clamp + the rollout-buffer rows: one launch -- which also does the
                    # PREVIOUS env step's bookkeeping (reward shaping, done flags, episode statistics): a rollout step is two launches
                    dr_blob = env.dr_step_args() if fold_dr else None
                    try:
                        self._policy_fwd.rollout_step(cur, net.sigma.detach(), fx["noise"][n], self.dones, vrms, mb["obs"][n], mb["dones"][n], mb["mu"][n],
                                                      mb["val"][n], mb["act"][n], fx["env_act"], mb["neglogp"][n], mb["sigma"][n],
                                                      prev_post=None if pending is None else F.RolloutPost.of(*pending, ep_parts=parts), action_noise=act_noise,
                                                      dr_step=dr_blob)
                    except BaseException:
                        # the launch that was to carry the coming step's randomisation did not run: hand it back, or the next env step
                        # would silently skip its randomisation pass (round-4 advisor finding)
                        if dr_blob is not None:
                            env.sim.dr_cancel()
                        raise
                    pending = None
                else:
                    x = self._f_obs_rms.normalize(cur, fx["obs_n"]) if self.normalize_input else cur
                    with torch.autocast("cuda", dtype=torch.float16, enabled=self.mixed_precision and not self.half_path):
                        mu, _logstd, value = net(x)
                    if mu.dtype not in (torch.float16, torch.float32):
                        mu, value = mu.float(), value.float()
                    # fp32 rows of obs / dones / mu / de-normalised value + sampling + neglogp + clamp: one launch
                    F.rollout_pre(mu.contiguous(), value.contiguous(), net.sigma.detach(), fx["noise"][n], cur, self.dones, vrms, mb["obs"][n], mb["dones"][n],
                                  mb["mu"][n], mb["val"][n], mb["act"][n], fx["env_act"], mb["neglogp"][n], mb["sigma"][n])
                if forked:
                    main.wait_stream(self._side_stream)   # join: the randomisation kernel of this step ran beside the policy launch
                    forked = False
                obs_dict, rew, dones, infos = self.vec_env.step(fx["env_act"])
                if prelaunch and n + 1 < self.horizon:
                    self._side_stream.wait_stream(main)   # fork behind this env step
                    with torch.cuda.stream(self._side_stream):
                        env.dr_prelaunch()
                    forked = True
                post = (rew, dones, infos["time_outs"] if "time_outs" in infos else dones, mb["val"][n], self.reward_scale, self.gamma,
                        boot and "time_outs" in infos, mb["rew"][n], self.dones, self.current_rewards, self.current_lengths, self.ep_stats)
                if fold and n + 1 < self.horizon and self._env_buffers_persist(rew, dones, infos):
                    pending = post   # rides in the next policy launch (the env's buffers keep this step's results until the next env step)
                else:
                    F.rollout_post(*post, ep_parts=parts if n + 1 == self.horizon else None)   # (the rollout's last launch folds the slots into ep_stats)
                o = obs_dict["obs"]
                if o.dtype == torch.float32 and o.is_contiguous() and o.device == self.obs.device:
                    # the env's persistent buffer (vec_task.py _clipped_obs): valid until the next step().  Under a HIP graph this
                    # branch is resolved ONCE, at capture: the replays read whatever address was seen then, so the env must hand back
                    # the same buffer on every step (host-side pointer compare, no sync)
                    if n == 0:
                        self._env_obs_ptr = o.data_ptr()
                    assert o.data_ptr() == self._env_obs_ptr, "the env returned a different observation buffer within one rollout"
                    cur = o
                else:
                    self.obs.copy_(o); cur = self.obs
            if cur is not self.obs:
                self.obs.copy_(cur)
        finally:
            if act_noise is not None:
                env.external_action_noise = False

    def _env_buffers_persist(self, rew, dones, infos):
        """The env hands back ITS OWN reward / reset / time-out buffers (fp32 / int64 on this device), the same ones on every step: then
        a later launch may still read this step's results from them.  Host-side pointer compares, no sync."""
        ts = (rew, dones) + ((infos["time_outs"],) if "time_outs" in infos else ())
        if not (rew.dtype == torch.float32 and dones.dtype == torch.int64 and all(t.dtype == torch.int64 for t in ts[2:])
                and all(t.is_contiguous() and t.device == self.obs.device for t in ts)):
            return False
        ptrs = tuple(t.data_ptr() for t in ts)
        seen = getattr(self, "_env_buf_ptrs", None)
        if seen is None:
            self._env_buf_ptrs = (ptrs, 1)
            return False   # first sight: nothing to compare with yet
        self._env_buf_ptrs = (ptrs, seen[1] + 1)
        return seen[0] == ptrs

    @torch.no_grad()
    def _rollout_impl(self, steps_done=False):
        """horizon_length env steps + GAE + dataset preparation; device ops only (capturable)."""
        mb, dev = self.mb, self.device
        for n in range(0 if self.fused else self.horizon):
            res = self.get_action_values(self.obs)
            mb["obs"][n].copy_(self.obs); mb["dones"][n].copy_(self.dones)
            mb["act"][n].copy_(res["actions"]); mb["mu"][n].copy_(res["mus"]); mb["sigma"][n].copy_(res["sigmas"])
            mb["val"][n].copy_(res["values"]); mb["neglogp"][n].copy_(res["neglogpacs"])
            obs_dict, rew, dones, infos = self.vec_env.step(torch.clamp(res["actions"], -1.0, 1.0))
            self.obs.copy_(obs_dict["obs"])
            rew = rew.to(dev).float().unsqueeze(1)
            shaped = rew * self.reward_scale
            if self.value_bootstrap and "time_outs" in infos:
                shaped = shaped + self.gamma * res["values"] * infos["time_outs"].to(dev).unsqueeze(1).float()
            mb["rew"][n].copy_(shaped)
            self.dones.copy_(dones.to(dev).float())
            self.current_rewards += rew.squeeze(1)
            self.current_lengths += 1
            self.ep_stats[0] += self.dones.sum()
            self.ep_stats[1] += (self.current_rewards * self.dones).sum()
            self.ep_stats[2] += (self.current_lengths * self.dones).sum()
            not_done = 1.0 - self.dones
            self.current_rewards *= not_done
            self.current_lengths *= not_done
        if self.fused and not steps_done:
            self._rollout_steps_fused()
        if self.fused and self._policy_fwd is not None:
            # bootstrap values: the one-launch forward; GAE: one thread per env instead of eight elementwise launches per step
            fx = self._fx
            self._policy_fwd(self.obs, fx["last_mu"], fx["last_v"])
            advs, returns = fx["advs"], fx["rets"]
            self._F.gae(mb["rew"], mb["val"], mb["dones"], self.dones, fx["last_v"], self.gamma, self.tau, advs, returns,
                        unnorm=self.value_mean_std if self.normalize_value else None)   # (the bootstrap values are de-normalised inside the launch)
        else:
            last_values = self.get_values(self.obs)
            advs = discount_values(self.gamma, self.tau, self.dones, last_values, mb["dones"], mb["val"], mb["rew"])
            returns = advs + mb["val"]
        # ---- prepare_dataset (rl_games a2c_continuous.prepare_dataset)
        ds = self.dataset
        aliased = mb["obs"].data_ptr() == ds["obs"].data_ptr()   # the rollout wrote the dataset's rows itself (_alloc_static)
        if not aliased:
            ds["obs"].copy_(swap_and_flatten01(mb["obs"]))
        fused_v = self.normalize_value and self.fused and getattr(self, "_f_val_rms", None) is not None
        if (self.fused and self._policy_fwd is not None and self.normalize_input and (fused_v or not self.normalize_value)
                and self.cfg.get("fused_dataset_prep", True)):
            # one rank: everything from here to the dataset's old_values / returns / advantages -- the per-minibatch observation moments, the
            # value / return moments, both value-normaliser updates, the two normalisations, the advantage and its normalisation, the
            # transposes into env-major rows -- in FOUR launches (csrc/bez_ppo.hip, bez_ppo_dataset_prep) instead of ~30
            sc = getattr(self, "_prep_scratch", None)
            if sc is None:
                assert not torch.cuda.is_current_stream_capturing()
                sc = self._prep_scratch = self._F.dataset_prep_scratch(self.num_minibatches, self.horizon, self.num_actors, dev)
            prep = lambda stages, sums=None: self._F.dataset_prep(ds["obs"], self.minibatch_size, self.num_minibatches, self._obs_mom, mb["val"], fx["rets"],
                                                                  self.value_mean_std if fused_v else None, self._val_mom, self._ret_mom, ds["old_values"],
                                                                  ds["returns"], ds["advantages"], self.normalize_advantage, sc, stages=stages, adv_sums=sums)
            if not _dist_on():
                done = prep(7)
            else:
                # data parallel: the same launches with the epoch's two collectives between them (the moments are plain sums, so the
                # all-reduced buffers hold the global batch's); the episode statistics ride in the second one
                st = self._adv_pack
                done = prep(1)
                if done:
                    self._all_reduce(self._mom_pack)
                    prep(2, st)
                    st[3:6] = self.ep_stats
                    self._all_reduce(st)
                    self.ep_stats.copy_(st[3:6])
                    prep(4, st)
            if done:
                if not aliased:
                    ds["old_logp"].copy_(swap_and_flatten01(mb["neglogp"])); ds["actions"].copy_(swap_and_flatten01(mb["act"]))
                    ds["mu"].copy_(swap_and_flatten01(mb["mu"]))
                    ds["sigma"].copy_(swap_and_flatten01(mb["sigma"]))
                return
        values, returns = swap_and_flatten01(mb["val"]), swap_and_flatten01(returns)
        # epoch collective 1 of 2: every moment that depends on the data alone, for the whole epoch
        if self.normalize_input:
            for i in range(self.num_minibatches):
                x = ds["obs"][i * self.minibatch_size:(i + 1) * self.minibatch_size]
                if self.fused:
                    self._f_obs_rms.moments(x, out=self._obs_mom[i])
                else:
                    self._obs_mom[i].copy_(self.running_mean_std.moments(x))
        if fused_v:
            values, returns = values.contiguous(), returns.contiguous()
            self._f_val_rms.moments(values, out=self._val_mom); self._f_val_rms.moments(returns, out=self._ret_mom)
        elif self.normalize_value:
            self._val_mom.copy_(self.value_mean_std.moments(values)); self._ret_mom.copy_(self.value_mean_std.moments(returns))
        if _dist_on():
            self._all_reduce(self._mom_pack)
        if fused_v:
            # RunningMeanStd.forward in train mode, twice (values, then returns): update -> normalise, two launches each
            fx, vr = self._fx, self._f_val_rms
            vr.apply(self._val_mom); values = vr.normalize(values, fx["val_n"])
            vr.apply(self._ret_mom); returns = vr.normalize(returns, fx["ret_n"])
        elif self.normalize_value:
            self.value_mean_std.eval()
            self.value_mean_std.update_from_moments(self._val_mom); values = self.value_mean_std(values)
            self.value_mean_std.update_from_moments(self._ret_mom); returns = self.value_mean_std(returns)
        adv = (returns - values).sum(dim=1)
        if self.normalize_advantage:
            if _dist_on():
                # epoch collective 2 of 2: the advantage moments need the normalised values, i.e. the first collective's result; the
                # episode statistics ride along (every rank then reports the job's mean reward, not its shard's)
                st = self._adv_pack
                st[0] = adv.sum(); st[1] = (adv * adv).sum(); st[2] = float(adv.numel()); st[3:6] = self.ep_stats
                self._all_reduce(st)
                self.ep_stats.copy_(st[3:6])
                mean = st[0] / st[2]
                std = torch.sqrt(((st[1] / st[2] - mean * mean) * st[2] / (st[2] - 1)).clamp_min(0))
                adv = (adv - mean.float()) / (std.float() + 1e-8)
            else:
                adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        ds["old_values"].copy_(values); ds["returns"].copy_(returns); ds["advantages"].copy_(adv)
        if not aliased:
            ds["old_logp"].copy_(swap_and_flatten01(mb["neglogp"])); ds["actions"].copy_(swap_and_flatten01(mb["act"]))
            ds["mu"].copy_(swap_and_flatten01(mb["mu"]))
            ds["sigma"].copy_(swap_and_flatten01(mb["sigma"]))

    def _mark_weights_dirty(self):
        self._weights_dirty = True
        self._packed_stale = True

    def mark_weights_dirty(self):
        """Public: call after writing the fp32 master weights by any route this agent cannot see (a raw-pointer kernel, `param.data` writes).
        `load_state_dict` and in-place tensor writes are noticed without it (post-hook / `_weights_signature`)."""
        self._mark_weights_dirty()

    def _weights_signature(self):
        """Host-only fingerprint of the master weights' identity: in-place writes bump a tensor's `_version`, re-pointing `param.data`
        changes its storage.  (The fused optimiser writes through raw pointers and changes neither: what it writes, it also mirrors.)"""
        return tuple((p._version, p.data_ptr()) for p in self.model.parameters())

    def _notice_external_weight_writes(self):
        sig = self._weights_signature()
        if sig != getattr(self, "_weights_sig", None):
            if getattr(self, "_weights_sig", None) is not None:
                self._mark_weights_dirty()
            self._weights_sig = sig

    def _copies_kept_current(self):
        """True where the fused optimiser launch maintains every derived weight copy (fp16 working copy, fragment-major forward / backward
        copies): the rollout then needs no refresh of its own."""
        return bool(self._fused_opt and self.half_path and getattr(self, "_hflat", None) is not None)

    def _refresh_weight_copies_if_dirty(self):
        """Weights written from outside the optimiser (load_state_dict on the model -- a post-hook marks it --, set_full_state_weights, the
        initial parameter broadcast): bring the derived copies up to date, eagerly, never inside a captured graph."""
        if self._weights_dirty and self._copies_kept_current():
            net = self.model.a2c_network
            net.refresh_half()
            if self._packed is not None:
                self._packed.refresh()
                self._packed_stale = False
        self._weights_dirty = False

    def play_steps(self):
        """Rollout + dataset.  With HIP graphs enabled the first call after warm-up captures, later calls replay."""
        if self.mb is None:
            self._alloc_static()
        self._notice_external_weight_writes()   # round-4 advisor finding: p.copy_(), EMA / perturbation tools, a late broadcast
        self._refresh_weight_copies_if_dirty()
        if self._g_rollout is not None and self._lean_env is not None and not getattr(self._lean_env, "_lean", True):
            self._lean_env.set_lean(True)       # the captured rollout steps lean whatever release_env() set in between
        if self.use_graphs and self._segmented and self._eager_epochs >= self.graph_warmup_epochs:
            # data parallel: the horizon loop has no collective and is replayed; GAE + dataset (two all-reduces) stay eager
            if self._g_rollout is None:
                torch.cuda.synchronize()
                self._g_rollout = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._g_rollout, pool=self._graph_pool()):
                    self._rollout_steps_fused()
            self._draw_rollout_noise()
            self._g_rollout.replay()  # (capture only records: the freshly captured graph is replayed like any later one)
            self._rollout_impl(steps_done=True)
        elif not self.use_graphs or self._eager_epochs < self.graph_warmup_epochs:
            self._rollout_impl()
        elif self._g_rollout is None:
            torch.cuda.synchronize()
            self._g_rollout = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_rollout, pool=self._graph_pool()):
                self._rollout_impl()
            self._draw_rollout_noise()
            self._g_rollout.replay()  # capture only records: the epoch that captures still has to run its rollout
        else:
            self._draw_rollout_noise()
            self._g_rollout.replay()
        return self.dataset

    def _draw_rollout_noise(self):
        """The horizon's action noise, drawn eagerly in front of a rollout-graph replay: a graph that uses torch's generator makes every replay
        fill the generator's seed / offset tensors first (two launches, ~9 us of the epoch); drawn outside, the captured rollout has no generator use."""
        fx = getattr(self, "_fx", None)
        if self.fused and fx is not None and "noise" in fx:
            fx["noise"].normal_()

    def _graph_pool(self):
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        return self._pool

    # ------------------------------------------------------------------ update
    def _bind_flat_grads(self):
        """Every parameter's .grad becomes a fixed view of ONE static fp32 buffer (124 237 + 8 floats; the tail carries the loss kernel's
        five minibatch sums -- a_loss, c_loss, b_loss, KL, entropy -- so that ONE clear per step zeroes gradients and sums alike and the
        KL travels with the gradient): autograd accumulates in place, the data-parallel all-reduce runs on the buffer itself (no
        flatten / unflatten copies), and graphs captured separately (forward+backward | optimiser) see the same addresses."""
        params = list(self.model.parameters())
        n = sum(p.numel() for p in params)
        self._nparam = n
        self._flat = torch.zeros(n + 8, device=self.device, dtype=torch.float32)
        off = 0
        for p in params:
            p.grad = self._flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self._flat_stats = self._flat[n:n + 5]   # SUMS over the minibatch's rows (after the all-reduce + division: the ranks' mean)
        self._flat_kl = self._flat[n + 3:n + 4]

    def _bind_flat_optimizer(self):
        """Parameters and Adam's moments become fixed views of three static fp32 buffers laid out like the flat gradient, so the
        optimiser tail (unscale, clip, Adam, scaler update) is ONE pass over them (fused.adam_step) instead of torch's
        dozen foreach launches.  `self.optimizer` stays a real torch.optim.Adam whose state tensors ARE those views: state_dict /
        load_state_dict and the checkpoint format are unchanged."""
        params = list(self.model.parameters())
        n = sum(p.numel() for p in params)
        dev = self.device
        self._pflat = torch.empty(n, device=dev, dtype=torch.float32)
        self._mflat = torch.zeros(n, device=dev, dtype=torch.float32)
        self._vflat = torch.zeros(n, device=dev, dtype=torch.float32)
        self._steps = torch.zeros(len(params), device=dev, dtype=torch.float32)
        self._opt_work = torch.zeros(self._F.ADAM_WORK_FLOATS, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for k, p in enumerate(params):
                sl = slice(off, off + p.numel())
                self._pflat[sl].copy_(p.data.reshape(-1))
                p.data = self._pflat[sl].view_as(p)
                self.optimizer.state[p] = {"step": self._steps[k], "exp_avg": self._mflat[sl].view_as(p), "exp_avg_sq": self._vflat[sl].view_as(p)}
                off += p.numel()
        if self.scaler.is_enabled() and self.scaler._scale is None:
            self.scaler._lazy_init_scale_growth_tracker(dev)
        # the fp16 working copies of the half path become views of one flat fp16 buffer with the same layout: the Adam kernel
        # writes them in the pass that updates the masters (no per-step multi-tensor cast)
        self._hflat = None
        net = self.model.a2c_network
        if getattr(self, "half_path", False) and getattr(net, "_p16", None) is not None:
            self._hflat = torch.empty(n, device=dev, dtype=torch.float16)
            offs, off = {}, 0
            for p in params:
                offs[id(p)] = off
                off += p.numel()
            with torch.no_grad():
                for j, p32 in enumerate(net._p32):
                    o = offs[id(p32)]
                    net._p16[j] = self._hflat[o:o + p32.numel()].view_as(p32)
            net.refresh_half()

    def _rebind_optimizer_state(self):
        """After Optimizer.load_state_dict (which installs fresh tensors): copy the loaded moments / step counts into the flat
        buffers and make the state entries views of them again."""
        off = 0
        with torch.no_grad():
            for k, p in enumerate(self.model.parameters()):
                sl = slice(off, off + p.numel())
                st = self.optimizer.state.get(p)
                if st:
                    self._mflat[sl].copy_(st["exp_avg"].reshape(-1)); self._vflat[sl].copy_(st["exp_avg_sq"].reshape(-1))
                    self._steps[k].copy_(torch.as_tensor(st["step"], dtype=torch.float32))
                self.optimizer.state[p] = {"step": self._steps[k], "exp_avg": self._mflat[sl].view_as(p), "exp_avg_sq": self._vflat[sl].view_as(p)}
                off += p.numel()

    def _allreduce_grads(self, kl):
        """ONE fused all-reduce per optimiser step: the flat fp32 gradient (124 237 elements = 497 KB) and the minibatch KL in
        its last slot.  The message is latency-bound on xGMI, so bucketing per parameter would only multiply the latency.
        Returns the mean KL."""
        params = [p for p in self.model.parameters() if p.grad is not None]
        n = sum(p.numel() for p in params)
        if self._flat_grad is None or self._flat_grad.numel() != n + 1:
            self._flat_grad = torch.empty(n + 1, device=self.device, dtype=torch.float32)
        off = 0
        for p in params:
            self._flat_grad[off:off + p.numel()].copy_(p.grad.reshape(-1))
            off += p.numel()
        self._flat_grad[n] = kl
        self._all_reduce(self._flat_grad)
        self._flat_grad.div_(dist.get_world_size())
        off = 0
        for p in params:
            p.grad.copy_(self._flat_grad[off:off + p.numel()].view_as(p.grad))
            off += p.numel()
        return self._flat_grad[n].clone()

    # ---- the fused optimiser step in two collective-free phases around its ONE collective (B | all-reduce grads + KL | C); the
    # observation moments the input normaliser absorbs at this step were computed -- and all-reduced -- once per epoch, after the rollout
    def _phase_b(self, mb):
        F, fx, net = self._F, self._fx, self.model.a2c_network
        self.model.train()
        obs = mb["obs"]
        if self.normalize_input:
            if self._rms_preapplied:   # the previous step's optimiser launch already absorbed this minibatch's moments
                self._rms_preapplied = False
            else:
                self._f_obs_rms.apply(self._obs_mom[mb["_i"]])
        if self.half_path and getattr(self, "_hflat", None) is None:
            net.refresh_half()  # (with the fused optimiser the Adam kernel keeps the fp16 copies current)
        elif self._weights_dirty and not torch.cuda.is_current_stream_capturing():
            self._refresh_weight_copies_if_dirty()   # an update without a rollout in front of it (tests, tools) after weights were loaded
        manual = self._train_fwd_ok(obs)
        wg = None
        if manual:
            # forward of the whole MLP as one MFMA kernel that keeps the ELU outputs (csrc/bez_policy.hip, mode 2); the backward pass
            # below is the chain autograd would run through _HalfLinearEluFn / _HalfLinearFn, called directly
            tf = self._train_bufs(obs.shape[0])
            if self._packed is not None and (self._packed_stale or not self._fused_opt):
                self._packed.refresh()  # forward and backward copies of this step's weights (the fused optimiser writes them itself)
                self._packed_stale = False
            self._policy_fwd.train_forward(obs, tf["x0"], tf["act"], tf["mu"], tf["v"])
            mu32, v32 = tf["mu"], tf["v"]
            wg = self._wgrad_plan(tf)
        else:
            if self.normalize_input:
                obs = self._f_obs_rms.normalize(obs, fx["mb_obs_n"])
            with torch.autocast("cuda", dtype=torch.float16, enabled=self.mixed_precision and not self.half_path):
                mu, _logstd, value = net(obs)
            mu32, v32 = mu.float().contiguous(), value.float().contiguous()
        scale = None
        if self.scaler.is_enabled():
            if self.scaler._scale is None:
                self.scaler._lazy_init_scale_growth_tracker(self.device)
            scale = self.scaler._scale
        # one launch for every second-stage reduction of the step (weights, biases, log-sigma, statistics), which WRITES the whole flat
        # gradient: no clear in front of the step (a minibatch step loses the fill and two of its three reduction launches)
        one_reduce = manual and wg is not None and self._policy_bwd is not None and self.cfg.get("fused_grad_reduce", True)
        if not one_reduce:
            self._flat.zero_()
        fused_loss = None
        if one_reduce and self.cfg.get("fused_loss_backward", True):
            # the loss rides in front of the backward chain (one launch; d loss / d mu, d loss / d value stay on the chip)
            fused_loss = F.LossOperands.of(mu32, net.sigma.detach(), v32, mb, self.e_clip, self.critic_coef, self.entropy_coef, self.bounds_loss_coef,
                                           self.clip_value, scale, self.update_mu_sigma, fx["loss_scratch"])
            lin = net._lin
            nh = len(lin) - 2
            if not self._policy_bwd.with_loss(fused_loss, tf["act"], tf["gz"], tf["gmu16"], tf["gv16"], [lin[L].bias.grad for L in range(nh)],
                                              lin[nh].bias.grad, lin[nh + 1].bias.grad):
                fused_loss = None
        if fused_loss is None:
            F.loss(mu32.detach(), net.sigma.detach(), v32.detach(), mb, self.e_clip, self.critic_coef, self.entropy_coef, self.bounds_loss_coef,
                   self.clip_value, scale, fx["gmu"], fx["gval"], net.sigma.grad, self._flat_stats, zero_glog=False, zero_stats=False,
                   update_mu_sigma=self.update_mu_sigma, scratch=fx["loss_scratch"], defer_reduce=one_reduce)
        if manual:
            self._manual_backward(tf, fx["gmu"], fx["gval"], wg, one_reduce, chain_done=fused_loss is not None)
        else:
            torch.autograd.backward([mu32, v32], [fx["gmu"], fx["gval"]])

    def _wgrad_plan(self, tf):
        """The split-K MFMA weight-gradient plan for the training buffers `tf` (None: switched off or shapes not taken).  Made (and
        uploaded, synchronously) in the first eager epoch, never inside a graph capture."""
        if self._policy_bwd is None or not self.cfg.get("fused_wgrad", True):
            return None
        F, net = self._F, self.model.a2c_network
        lin = net._lin
        nh = len(lin) - 2
        h_last = tf["act"][nh - 1]
        dys = [tf["gz"][L] for L in range(nh)] + [tf["gmu16"], tf["gv16"]]
        xs = [tf["x0"]] + [tf["act"][L] for L in range(nh - 1)] + [h_last, h_last]
        grads = [lin[L].weight.grad for L in range(nh + 2)]
        wg = getattr(self, "_wgrad_mfma", None)
        if wg is None or not wg.matches(dys, xs, grads):
            assert not torch.cuda.is_current_stream_capturing(), "the weight-gradient plan must exist before the update is captured"
            wg = self._wgrad_mfma = F.WgradMfma(dys, xs, grads)
        return wg if wg.ok else None

    def _train_fwd_ok(self, obs):
        net = self.model.a2c_network
        if self._policy_fwd is None or not self.normalize_input or not self.cfg.get("fused_train_forward", True):
            return False
        k, s = obs.shape[0], net._splits
        lin = net._lin
        return (obs.dtype == torch.float32 and obs.is_contiguous() and s > 1 and k % s == 0 and k // s >= 64 and obs.shape[1] % 2 == 0
                and all(m.weight.shape[0] % 2 == 0 for m in lin[:-2])
                and all(m.weight.grad is not None and m.bias.grad is not None and m.weight.grad.is_contiguous() and m.bias.grad.is_contiguous() for m in lin))

    def _train_bufs(self, k):
        tf = getattr(self, "_tf", None)
        if tf is None or tf["k"] != k:
            net, dev = self.model.a2c_network, self.device
            widths = [m.weight.shape[0] for m in net._lin[:-2]]
            h = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float16)
            tf = dict(k=k, x0=h(k, self.obs_dim), act=[h(k, w) for w in widths], gz=[h(k, w) for w in widths], g=[h(k, w) for w in widths],
                      mu=torch.empty(k, self.act_dim, device=dev), v=torch.empty(k, 1, device=dev),
                      gmu16=h(k, self.act_dim), gv16=h(k, 1))
            self._tf = tf
        return tf

    def _manual_backward(self, tf, gmu, gval, wg=None, one_reduce=False, chain_done=False):
        """d(loss)/d(parameters) from the loss kernel's d/d(mu), d/d(value): per layer one input-gradient GEMM, the split-K weight
        gradient reduced straight into the fp32 master .grad views, and the fused ELU-derivative / bias-gradient pass -- the same
        launches _HalfLinearFn / _HalfLinearEluFn issue under autograd (tests hold the two against each other)."""
        F, net = self._F, self.model.a2c_network
        lin, p16, s = net._lin, net._p16, net._splits
        nh = len(lin) - 2
        k = gmu.shape[0]
        ks = k // s

        def wgrad(g16, x16, m):
            part = torch.bmm(g16.view(s, ks, -1).transpose(1, 2), x16.view(s, ks, -1))
            F.wgrad_sum(part, m.weight.grad, accumulate=True)
        h_last = tf["act"][nh - 1]
        if self._policy_bwd is not None:
            # the whole input-gradient chain (head casts + bias sums, per layer ELU derivative + bias sum + dgrad GEMM) in one launch on
            # the transposed weight copies (refreshed by one scatter of the fp16 working copy); the weight gradients follow as GEMMs
            if self._packed is None:
                self._policy_bwd.refresh()
            bias_grads = [lin[L].bias.grad for L in range(nh)]
            if not chain_done:   # (chain_done: PolicyBackward.with_loss already ran it, behind the loss)
                self._policy_bwd(gmu, gval, tf["act"], tf["gz"], tf["gmu16"], tf["gv16"], bias_grads, lin[nh].bias.grad, lin[nh + 1].bias.grad,
                                 defer_reduce=one_reduce)
            # all five weight gradients: one split-K MFMA launch over the output blocks of every layer (csrc/bez_wgrad.hip) instead of a
            # batched GEMM and a sum per layer ...
            if one_reduce:
                # ... and ONE fixed-order reduction launch for its partial images, the bias column sums and the loss kernel's sums
                wg(reduce=False)
                np_ = getattr(self, "_norm_parts", None)
                nb = F.grad_reduce_blocks(wg, self._policy_bwd)
                if np_ is None or np_.shape[0] != nb:
                    assert not torch.cuda.is_current_stream_capturing()
                    np_ = self._norm_parts = torch.zeros(nb, 2, device=self.device, dtype=torch.float32)
                F.grad_reduce_all(wg, self._policy_bwd, bias_grads, lin[nh].bias.grad, lin[nh + 1].bias.grad, gmu.shape[0], self._fx["loss_scratch"],
                                  net.sigma.grad, self._flat_stats, accumulate=False, norm_parts=np_)
                self._norm_parts_fresh = True   # they describe the gradient in the flat buffer (until a collective changes it)
                return
            if wg is not None and wg(accumulate=True):
                return
            wgrad(tf["gmu16"], h_last, lin[nh])
            wgrad(tf["gv16"], h_last, lin[nh + 1])
            for L in range(nh - 1, -1, -1):
                wgrad(tf["gz"][L], tf["act"][L - 1] if L > 0 else tf["x0"], lin[L])
            return
        # the cast nodes' backward (fp32 -> fp16) and both head bias gradients: one launch
        F.head_grads_f16(gmu, gval, tf["gmu16"], tf["gv16"], lin[nh].bias.grad, lin[nh + 1].bias.grad)
        g = tf["g"][nh - 1]
        torch.mm(tf["gmu16"], p16[2 * nh], out=g)                 # d/d(h): mu head ...
        g.addmm_(tf["gv16"], p16[2 * (nh + 1)])                   # ... + value head (autograd sums the two branches in fp16 as well)
        wgrad(tf["gmu16"], h_last, lin[nh])
        wgrad(tf["gv16"], h_last, lin[nh + 1])
        for L in range(nh - 1, -1, -1):
            gz = tf["gz"][L]
            F.elu_bwd_colsum_f16(g, tf["act"][L], gz, lin[L].bias.grad, accumulate=True)
            x = tf["act"][L - 1] if L > 0 else tf["x0"]
            if L > 0:
                g = tf["g"][L - 1]
                torch.mm(gz, p16[2 * L], out=g)
            wgrad(gz, x, lin[L])

    def _phase_c(self, kl_out, loss_out, next_i=None):
        """next_i: row of the epoch's observation moments the NEXT minibatch step absorbs (None: that step applies them itself)"""
        fresh, self._norm_parts_fresh = getattr(self, "_norm_parts_fresh", False), False
        wdiv = 1.0
        if _dist_on():
            wdiv = float(dist.get_world_size())
            fresh = False                            # (the all-reduce changed the gradient its producer left the norm shares for)
            if not self._fused_opt:
                self._flat.div_(wdiv)                # mean of the (still scaled) gradients and of the KL
        if self._fused_opt:
            g0 = self.optimizer.param_groups[0]
            amp = self.scaler.is_enabled()
            # (the epoch's KL / loss accumulators ride in the optimiser's last launch: the sums of the loss kernel -> means)
            # data parallel: the flat buffer holds the SUM over the ranks (gradient and statistics alike); the mean is never formed by a
            # pass of its own -- the division rides in the optimiser launch's unscale factor, in the tail scales and in the KL threshold
            rows = float(self.minibatch_size) * wdiv
            st = self._flat_stats
            tail = ((kl_out, st[3:4], 1.0 / (rows * self.num_minibatches)), (loss_out[0:1], st[0:1], 1.0 / rows), (loss_out[1:2], st[1:2], 1.0 / rows))
            # 'legacy' schedule: the lr moves after every step, in the same launch (the step's KL is a SUM over the rows: threshold scaled)
            sc = self.scheduler
            adapt = (self._flat_kl, sc.kl_threshold * rows, sc.min_lr, sc.max_lr) if self.is_adaptive_lr and self.schedule_type == "legacy" else None
            parts, gridn = (self._norm_parts if fresh else None), None
            if wdiv != 1.0 or _dist_on():
                # ... and the norm of the all-reduced buffer is formed INSIDE the optimiser launch: every workgroup sums its own slice, the
                # workgroups meet at a counter (`dp_grid_norm`, default).  Off: one small launch re-forms the shares (bez_ppo_grad_norm_parts).
                # (Every workgroup reading the whole gradient instead costs the launch 18 us against 9.)
                if getattr(self, "_grid_fits", None) is None:   # (host query, once: can every workgroup of the launch be resident at the same time?)
                    self._grid_fits = bool(self._F.adam_grid_fits(self._nparam))
                if self.cfg.get("dp_grid_norm", True) and self._grid_fits:
                    gridn = getattr(self, "_grid_norm", None)
                    if gridn is None:
                        assert not torch.cuda.is_current_stream_capturing()
                        gridn = self._grid_norm = torch.zeros(self._F.ADAM_GRIDNORM_FLOATS, device=self.device, dtype=torch.float32)
                else:
                    npd = getattr(self, "_norm_parts_dp", None)
                    if npd is None:
                        assert not torch.cuda.is_current_stream_capturing()
                        npd = self._norm_parts_dp = torch.zeros((self._nparam // 4 + 3 + 1023) // 1024 + 1, 2, device=self.device, dtype=torch.float32)
                    parts = self._F.grad_norm_parts(self._flat[:self._nparam], npd)
            self._F.adam_step(self._pflat, self._flat[:self._nparam], self._mflat, self._vflat, self._steps, self.lr_t, g0["betas"], g0["eps"],
                              g0["weight_decay"], self.grad_norm if self.truncate_grads else 0.0, self.scaler._scale if amp else None,
                              self.scaler._growth_tracker if amp else None, self.scaler.get_growth_factor(), self.scaler.get_backoff_factor(),
                              self.scaler.get_growth_interval(), self._opt_work, self._hflat, tail=tail, adapt=adapt,
                              packed=self._packed if (self._packed is not None and self._hflat is not None and not self._packed_stale) else None,
                              next_rms=(self._f_obs_rms, self._obs_mom[next_i]) if (next_i is not None and self.normalize_input) else None,
                              norm_parts=parts, grad_div=wdiv, grid_norm=gridn)
            if next_i is not None and self.normalize_input:
                self._rms_preapplied = True
            return
        else:
            if self.truncate_grads:
                self.scaler.unscale_(self.optimizer)
                clip_grad_norm_capturable(self.model.parameters(), self.grad_norm)
            self.scaler.step(self.optimizer)
            self.scaler.update()
        with torch.no_grad():
            # (one launch each: add with a scalar multiplier)
            kl_out.add_(self._flat_kl[0], alpha=1.0 / (self.num_minibatches * float(self.minibatch_size)))
            loss_out.add_(self._flat_stats[0:2], alpha=1.0 / float(self.minibatch_size))
            if self.is_adaptive_lr and self.schedule_type == "legacy":
                self.scheduler.update_(self.lr_t, self._flat_kl, scale=float(self.minibatch_size))

    def _calc_gradients_fused(self, mb, kl_out, loss_out, next_i=None):
        """calc_gradients with the HIP glue kernels: running update of the input normaliser from the epoch's precomputed moments, MLP
        forward (torch), the whole loss and its gradient w.r.t. mu / value / log-std (1 launch), MLP backward (torch, into the
        static flat gradient), then the all-reduce / unscale / clip / Adam / scaler tail."""
        self._phase_b(mb)
        if _dist_on():
            # ONE all-reduce of the flat STILL-SCALED gradient + KL (124 238 fp32 = 497 KB, latency-bound on xGMI): an fp16
            # overflow on any rank reaches every rank, so unscale_ records the same found_inf everywhere (as DDP does)
            self._all_reduce(self._flat)
        self._phase_c(kl_out, loss_out, next_i)

    def calc_gradients(self, mb, kl_out, loss_out, next_i=None):
        """One optimiser step on minibatch `mb`; device ops only.  KL is written to kl_out (0-dim view), losses added to loss_out.
        next_i: index of the minibatch the following step will train on (None: this is the update's last step)."""
        if self.fused:
            return self._calc_gradients_fused(mb, kl_out, loss_out, next_i)
        self.model.train()
        if self.normalize_input:
            # rl_games runs the input normaliser in train mode here (every minibatch forward updates it); what it absorbs are this
            # minibatch's moments, computed and all-reduced once per epoch after the rollout
            self.running_mean_std.eval()
            self.running_mean_std.update_from_moments(self._obs_mom[mb["_i"]])
        obs = self._preproc_obs(mb["obs"])
        e = self.e_clip
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.mixed_precision):
            res = self.model(obs, mb["actions"])
            neglogp, values, entropy, mu, sigma = (res[k].float() for k in ("prev_neglogp", "values", "entropy", "mus", "sigmas"))
            ratio = torch.exp(mb["old_logp"] - neglogp)
            surr1 = mb["advantages"] * ratio
            surr2 = mb["advantages"] * torch.clamp(ratio, 1.0 - e, 1.0 + e)
            a_loss = torch.max(-surr1, -surr2)
            if self.clip_value:
                vclip = mb["old_values"] + (values - mb["old_values"]).clamp(-e, e)
                c_loss = torch.max((values - mb["returns"]) ** 2, (vclip - mb["returns"]) ** 2)
            else:
                c_loss = (mb["returns"] - values) ** 2
            if self.bounds_loss_coef > 0:
                soft = 1.1
                b_loss = (torch.clamp_min(mu - soft, 0.0) ** 2 + torch.clamp_max(mu + soft, 0.0) ** 2).sum(-1)
            else:
                b_loss = torch.zeros_like(a_loss)
            a_l, c_l, ent, b_l = a_loss.mean(), c_loss.mean(), entropy.mean(), b_loss.mean()
            loss = a_l + 0.5 * c_l * self.critic_coef - ent * self.entropy_coef + b_l * self.bounds_loss_coef
        self.optimizer.zero_grad(set_to_none=True)
        self.scaler.scale(loss).backward()
        kl = policy_kl(mu.detach(), sigma.detach(), mb["mu"], mb["sigma"])
        if _dist_on():
            # the step's ONE collective: the STILL-SCALED gradients (as DDP does: an fp16 overflow on any rank reaches every rank, so
            # unscale_ below records the same found_inf everywhere and all replicas skip or take the step together) + the KL
            kl = self._allreduce_grads(kl)
        if self.truncate_grads:
            self.scaler.unscale_(self.optimizer)
            clip_grad_norm_capturable(self.model.parameters(), self.grad_norm)
        self.scaler.step(self.optimizer)
        self.scaler.update()
        with torch.no_grad():
            kl_out.add_(kl / self.num_minibatches)
            loss_out[0] += a_l.detach(); loss_out[1] += c_l.detach()
            if self.update_mu_sigma:   # PPODataset.update_mu_sigma [ext]
                mb["mu"].copy_(mu.detach()); mb["sigma"].copy_(sigma.detach())
            if self.is_adaptive_lr and self.schedule_type == "legacy":
                self.scheduler.update_(self.lr_t, kl.detach().float())

    def _minibatch(self, i):
        sl = slice(i * self.minibatch_size, (i + 1) * self.minibatch_size)
        mb = {k: v[sl] for k, v in self.dataset.items()}
        mb["_i"] = i  # row of the epoch's precomputed observation moments
        return mb

    def _zero_update_sums(self):
        acc = getattr(self, "_upd_sums", None)
        if acc is None and self._report is not None and self.loss_acc.data_ptr() + 8 == self.kl_acc.data_ptr():
            acc = self._upd_sums = self._report.view(torch.float32)[10:12 + self.mini_epochs]   # loss_acc | kl_acc: neighbours in the epoch report, one fill
        if acc is not None:
            acc.zero_()
        else:
            self.kl_acc.zero_(); self.loss_acc.zero_()

    def _update_impl(self):
        """mini_epochs x num_minibatches optimiser steps + the adaptive LR rule, all on the device."""
        self._zero_update_sums()
        self._rms_preapplied = False
        last = self.mini_epochs * self.num_minibatches - 1
        for ep in range(self.mini_epochs):
            for i in range(self.num_minibatches):
                step = ep * self.num_minibatches + i
                self.calc_gradients(self._minibatch(i), self.kl_acc[ep], self.loss_acc, None if step == last else (i + 1) % self.num_minibatches)
            if self.is_adaptive_lr and self.schedule_type != "legacy":
                self.scheduler.update_(self.lr_t, self.kl_acc[ep])

    def _update_segmented(self):
        """Data-parallel update with HIP graphs; the step's ONE RCCL all-reduce (gradient + KL) runs eagerly between replays (no collective is
        ever captured).  Segments: the forward/backward half of a step ("B", per minibatch) and the optimiser half ("C") -- and because the C
        of step s and the B of step s + 1 have no collective between them, they are ONE graph: an update of S steps is S + 1 replays
        (B0 | CB ... CB | C) instead of 2 S, and the C inside a CB graph absorbs the next minibatch's observation moments as the one-rank
        path's does.  The first call captures the segments (capture records, it does not execute) and then replays them like every later call."""
        self._rms_preapplied = False
        nm = self.num_minibatches
        if self._seg is None:
            torch.cuda.synchronize()
            seg = dict(b0=torch.cuda.CUDAGraph(), cb=[], c=torch.cuda.CUDAGraph(), kl=torch.zeros((), device=self.device))
            with torch.cuda.graph(seg["b0"], pool=self._graph_pool()):
                self._phase_b(self._minibatch(0))
            for i in range(nm):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self._graph_pool()):
                    self._phase_c(seg["kl"], self.loss_acc, i)
                    self._phase_b(self._minibatch(i))
                seg["cb"].append(g)
            with torch.cuda.graph(seg["c"], pool=self._graph_pool()):
                self._phase_c(seg["kl"], self.loss_acc)
            self._rms_preapplied = False
            self._seg = seg
        seg = self._seg
        self._zero_update_sums()
        seg["kl"].zero_()
        steps = self.mini_epochs * nm

        def close_mini_epoch(ep):   # the C that completed mini-epoch `ep` has just been replayed
            self.kl_acc[ep].copy_(seg["kl"])
            seg["kl"].zero_()
            if self.is_adaptive_lr and self.schedule_type != "legacy":
                self.scheduler.update_(self.lr_t, self.kl_acc[ep])
        for s_ in range(steps):
            if s_ == 0:
                seg["b0"].replay()
            else:
                seg["cb"][s_ % nm].replay()
                if s_ % nm == 0:
                    close_mini_epoch(s_ // nm - 1)
            if _dist_on() and not _skip_collective_for_measurement():
                self._all_reduce(self._flat)
        seg["c"].replay()
        close_mini_epoch(self.mini_epochs - 1)

    def run_update(self):
        if self.use_graphs and self._segmented and self._eager_epochs >= self.graph_warmup_epochs:
            if self.cfg.get("dp_eager_update", False):   # A/B: the update's ~120 launches eagerly instead of S + 1 graph segments (the rollout stays a graph)
                return self._update_impl()
            return self._update_segmented()
        if not self.use_graphs or self._eager_epochs < self.graph_warmup_epochs:
            self._update_impl()
        elif self._g_update is None:
            torch.cuda.synchronize()
            self._g_update = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_update, pool=self._graph_pool()):
                self._update_impl()
            self._g_update.replay()  # (same: the capturing epoch performs its update by replaying the fresh graph)
        else:
            self._g_update.replay()

    def _drain_episode_stats(self, host=None, zero=True):
        """Finished-episode count / return / length sums of this epoch's rollout (host = the values already read back with the epoch
        report; None: one device->host read of its own).  zero False: the caller has cleared the device sums already (pipelined epochs)."""
        cnt, rsum, lsum = self.ep_stats.tolist() if host is None else host
        if zero:
            self.ep_stats.zero_()
        if cnt > 0:
            self._ep_hist.append((cnt, rsum, lsum))
            while len(self._ep_hist) > 1 and sum(c for c, _, _ in self._ep_hist[1:]) >= self.games_to_track:
                self._ep_hist.pop(0)
        tot = sum(c for c, _, _ in self._ep_hist)
        if tot > 0:  # mean over (at least) the last games_to_track finished episodes
            self.game_rewards = [sum(r for _, r, _ in self._ep_hist) / tot]
            self.game_lengths = [sum(l for _, _, l in self._ep_hist) / tot]

    def train_epoch(self):
        t0 = time.perf_counter()
        packed = self._report is not None and self.mb is not None and self.ep_stats.data_ptr() == self._report.data_ptr()
        if packed:
            # no host synchronisation between the rollout and the update: their shares of the epoch come from HIP events, resolved at
            # the epoch's ONE device-to-host copy
            ev = self._epoch_events = getattr(self, "_epoch_events", None) or [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
            self.play_steps()
            ev[1].record()
            self.run_update()
            ev[2].record()
            rep = self._report.cpu()     # the epoch's host sync
            f32 = rep.view(torch.float32)
            self.last_lr = float(f32[8])
            kls = f32[12:12 + self.mini_epochs].tolist()
            a_l, c_l = (f32[10:12] / (self.mini_epochs * self.num_minibatches)).tolist()
            self._drain_episode_stats(rep[0:3].tolist())
            t_total = time.perf_counter() - t0
            dev_play, dev_upd = ev[0].elapsed_time(ev[1]) * 1e-3, ev[1].elapsed_time(ev[2]) * 1e-3
            t_play = t_total * dev_play / max(dev_play + dev_upd, 1e-9)   # the wall time split as the device time was
        else:
            self.play_steps()
            if self.device.type == "cuda":
                torch.cuda.synchronize()
            t_play = time.perf_counter() - t0
            self.run_update()
            self.last_lr = float(self.lr_t.item())  # the epoch's only other host sync
            kls = self.kl_acc.tolist()
            a_l, c_l = (self.loss_acc / (self.mini_epochs * self.num_minibatches)).tolist()
            self._drain_episode_stats()
            t_total = time.perf_counter() - t0
        self._eager_epochs += 1
        self.epoch_num += 1
        self.frame += self.batch_size * self.world
        self._weights_sig = self._weights_signature()   # whatever this epoch's own optimiser steps did to the versions is not "external"
        return dict(play_time=t_play, update_time=t_total - t_play, total_time=t_total, kl=sum(kls) / len(kls),
                    a_loss=a_l, c_loss=c_l, lr=self.last_lr)

    # ---- pipelined epochs: the host reads epoch k's report while epoch k + 1 is already queued.  train_epoch() ends in the epoch's one
    # device-to-host copy and only then launches the next rollout: between the two the GPU waits for the host (wake-up from the copy, the
    # report, the next epoch's fills / noise / graph launch) -- ~120 us of a 3.9 ms epoch in the kernel trace.  Here the report goes to a pinned
    # buffer behind the epoch's kernels (two slots), the device sums are cleared in stream order, and the caller collects the report one epoch late.
    def train_epoch_launch(self):
        """Queues one epoch (rollout + update + the copy of its report) and returns a ticket for train_epoch_finish(); None where the epoch
        report is not packed (CPU, unfused paths): the caller then uses train_epoch()."""
        packed = self._report is not None and self.mb is not None and self.ep_stats.data_ptr() == self._report.data_ptr()
        if not packed or self.device.type != "cuda":
            return None
        pipe = getattr(self, "_pipe", None)
        if pipe is None:
            pipe = self._pipe = [dict(ev=[torch.cuda.Event(enable_timing=True) for _ in range(3)], done=torch.cuda.Event(),
                                      host=torch.zeros(self._report.numel(), dtype=self._report.dtype).pin_memory(), busy=False) for _ in range(2)]
            self._pipe_next = 0
        slot = self._pipe_next
        st = pipe[slot]
        assert not st["busy"], "train_epoch_finish() the ticket of two epochs ago first"
        self._pipe_next = 1 - slot
        st["t0"] = time.perf_counter()
        ev = st["ev"]
        # the rollout / update split of an epoch comes from three timing events, and each costs the GPU ~5 us where it is recorded (kernel trace):
        # taken on every 8th epoch, the share carried over in between
        timed = st["timed"] = (self.epoch_num % 8 == 0) or not hasattr(self, "_play_share")
        if timed:
            ev[0].record()
        self.play_steps()
        if timed:
            ev[1].record()
        self.run_update()
        if timed:
            ev[2].record()
        st["host"].copy_(self._report, non_blocking=True)
        self.ep_stats.zero_()            # (stream order: behind the copy, in front of the next rollout's first count)
        st["done"].record()
        st["busy"] = True
        self._eager_epochs += 1
        self.epoch_num += 1
        self.frame += self.batch_size * self.world
        st["epoch"], st["frame"] = self.epoch_num, self.frame
        self._weights_sig = self._weights_signature()
        return slot

    def train_epoch_finish(self, ticket):
        """Waits for the epoch of `ticket` and returns what train_epoch() returns for it (+ "epoch", "frame"); total_time is the wall time since
        the previous report (or since the epoch's launch, for the first)."""
        st = self._pipe[ticket]
        assert st["busy"]
        st["done"].synchronize()
        st["busy"] = False
        rep = st["host"]
        f32 = rep.view(torch.float32)
        self.last_lr = float(f32[8])
        kls = f32[12:12 + self.mini_epochs].tolist()
        a_l, c_l = (f32[10:12] / (self.mini_epochs * self.num_minibatches)).tolist()
        self._drain_episode_stats(rep[0:3].tolist(), zero=False)
        now = time.perf_counter()
        t_total = now - max(st["t0"], getattr(self, "_pipe_last_report", 0.0))
        self._pipe_last_report = now
        ev = st["ev"]
        if st["timed"]:
            dev_play, dev_upd = ev[0].elapsed_time(ev[1]) * 1e-3, ev[1].elapsed_time(ev[2]) * 1e-3
            self._play_share = dev_play / max(dev_play + dev_upd, 1e-9)
        t_play = t_total * getattr(self, "_play_share", 0.35)
        return dict(play_time=t_play, update_time=t_total - t_play, total_time=t_total, kl=sum(kls) / len(kls), a_loss=a_l, c_loss=c_l,
                    lr=self.last_lr, epoch=st["epoch"], frame=st["frame"])

    def release_env(self):
        """Hands the env back to other consumers: lean stepping off, so env.net_contact_forces / feet / prev_lin_vel are
        current again after its next step (the agent switched them off for its rollouts)."""
        if getattr(self, "_lean_env", None) is not None:
            self._lean_env.set_lean(False)

    def __del__(self):
        try:
            self.release_env()
        except Exception:
            pass

    def train(self, max_epochs=None, log=print):
        if self._lean_env is not None:
            self._lean_env.set_lean(True)  # a second train() after release_env(): the captured rollout graphs step lean
        try:
            return self._train(max_epochs, log)
        finally:
            self.release_env()

    def _train(self, max_epochs=None, log=print):
        self.obs = self.env_reset()
        max_epochs = max_epochs or self.max_epochs
        total_time = 0.0
        won = False

        def report(st):   # one epoch's report: log line, observer, checkpoints, the stop rule
            nonlocal total_time, won
            total_time += st["total_time"]
            epoch, frame = st.get("epoch", self.epoch_num), st.get("frame", self.frame)
            mean_rew = sum(self.game_rewards) / len(self.game_rewards) if self.game_rewards else float("nan")
            if self.rank == 0:
                fps = self.batch_size * self.world / st["total_time"]
                log("epoch %d frames %d fps total %.0f (step %.0f) kl %.5f lr %.2e a_loss %.4f c_loss %.4f mean_reward %.3f" % (
                    epoch, frame, fps, self.batch_size * self.world / st["play_time"], st["kl"], st["lr"],
                    st["a_loss"], st["c_loss"], mean_rew))
                if self.writer is not None:
                    self.writer.add(dict(st, epoch=epoch, frame=frame, time=total_time, fps=fps, mean_reward=mean_rew))
                if self.save_frequency and epoch % self.save_frequency == 0:
                    self.save(os.path.join("runs", str(self.name), "nn", "last_%s_ep_%d.pth" % (self.name, epoch)))
                if self.game_rewards and mean_rew > self.last_mean_rewards and epoch >= self.save_best_after:
                    # (pipelined epochs: the weights saved are those behind the epoch in flight, one update further than the report that chose them)
                    self.last_mean_rewards = mean_rew
                    self.save(os.path.join("runs", str(self.name), "nn", "%s.pth" % self.name))
            if self.game_rewards and mean_rew > self.score_to_win:
                won = True
        pending = None
        pipelined = bool(self.cfg.get("pipeline_epochs", True))
        while self.epoch_num < max_epochs and not won:
            ticket = self.train_epoch_launch() if pipelined else None
            if ticket is None:
                report(self.train_epoch())
                continue
            if pending is not None:
                report(self.train_epoch_finish(pending))    # the epoch BEFORE the one just queued
            pending = ticket
            if self.save_frequency and self.epoch_num % self.save_frequency == 0:   # a periodic checkpoint holds exactly its epoch's weights
                report(self.train_epoch_finish(pending))
                pending = None
        if pending is not None:
            report(self.train_epoch_finish(pending))
        return self.last_mean_rewards, self.epoch_num

    # ------------------------------------------------------------------ checkpoints (rl_games key layout)
    def get_full_state_weights(self):
        state = {"model": self.model.state_dict(), "epoch": self.epoch_num, "frame": self.frame,
                 "optimizer": self.optimizer.state_dict(), "last_mean_rewards": self.last_mean_rewards,
                 "scaler": self.scaler.state_dict(), "env_state": None}
        if self.normalize_input:
            state["running_mean_std"] = self.running_mean_std.state_dict()
        if self.normalize_value:
            state["reward_mean_std"] = self.value_mean_std.state_dict()
        return state

    def set_full_state_weights(self, state):
        self.model.load_state_dict(state["model"])
        if self.normalize_input and "running_mean_std" in state:
            self.running_mean_std.load_state_dict(state["running_mean_std"])
        if self.normalize_value and "reward_mean_std" in state:
            self.value_mean_std.load_state_dict(state["reward_mean_std"])
        if "optimizer" in state:
            self.optimizer.load_state_dict(state["optimizer"])
            # load_state_dict replaces param_groups[..]['lr'] by a copy: re-point every group at the live device tensor the
            # adaptive-KL rule updates, and carry the checkpoint's learning rate over into it
            lr = self.optimizer.param_groups[0]["lr"]
            lr = float(lr.item() if torch.is_tensor(lr) else lr)
            if isinstance(self.lr_t, _CpuLr):
                self.lr_t.copy_(torch.tensor(lr))
            else:
                self.lr_t.fill_(lr)
                for g in self.optimizer.param_groups:
                    g["lr"] = self.lr_t
            self.last_lr = lr
            if getattr(self, "_fused_opt", False):
                self._rebind_optimizer_state()
        self.epoch_num = int(state.get("epoch", 0))
        self.frame = int(state.get("frame", 0))
        self.last_mean_rewards = float(state.get("last_mean_rewards", -100500.0))

    def save(self, path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(self.get_full_state_weights(), path)

    def restore(self, path):
        """Load a checkpoint WITHOUT ever unpickling arbitrary objects: this build's own files pass torch's weights_only
        loader; anything else (e.g. the reference's rl_games .pth, whose pickle references numpy scalars) is read by
        utils/rlg_checkpoint.py, which only disassembles the pickle -- model and normaliser statistics are restored,
        optimiser state is not."""
        try:
            state = torch.load(path, map_location=self.device, weights_only=True)
        except Exception:
            from ..utils.rlg_checkpoint import load_into_agent_modules, read_rlgames_checkpoint
            ck = read_rlgames_checkpoint(path)
            load_into_agent_modules(ck, self.model, self.running_mean_std, self.value_mean_std)
            self.epoch_num = int(ck.get("epoch", 0) or 0)
            self.frame = int(ck.get("frame", 0) or 0)
            self.last_mean_rewards = float(ck.get("last_mean_rewards", -100500.0) or -100500.0)
            return
        self.set_full_state_weights(state)
