"""ctypes bindings of the PPO glue kernels (include/bez_sim.h "bez_ppo_*", csrc/bez_ppo.hip) and the torch-side plumbing
around them: device pointers, the current HIP stream, and one autograd bridge (the fused loss returns d loss / d mu,
d loss / d value and d loss / d log-std; the MLP's own backward stays PyTorch's).  HIP only -- the CPU path of the agent
keeps the plain torch formulation these kernels are tested against."""
import ctypes as C

import numpy as np
import torch

from ..sim import load_library

_vp, _i32, _i64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_SIGS = {
    "bez_ppo_rms_moments": [_vp, _i64, _i32, _vp, _vp, _vp],
    "bez_ppo_rms_apply": [_vp, _i32, _vp, _vp, _vp, _vp],
    "bez_ppo_rms_normalize": [_vp, _i64, _i32, _vp, _vp, _f, _vp, _i32, _vp],
    "bez_ppo_sample": [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp],
    "bez_ppo_rollout_post": [_vp, _vp, _vp, _vp, _i64, _f, _f, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "bez_ppo_rollout_post_fold": [_vp, _vp, _vp, _vp, _i64, _f, _f, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp],
    "bez_ppo_loss": [_vp] * 10 + [_i64, _i32, _f, _f, _f, _f, _i32] + [_vp] * 7,
    "bez_ppo_rollout_pre": [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i64, _i32, _i32] + [_vp] * 9,
    "bez_ppo_policy_forward": [_vp, _i64, _i32, _vp, _vp, _f, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp],
    "bez_ppo_policy_rollout_step": [_vp, _i64, _i32, _vp, _vp, _f, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f] + [_vp] * 8 + [_i32, _vp, _vp, _vp, _vp, _vp],
    "bez_ppo_policy_forward_train": [_vp, _i64, _i32, _vp, _vp, _f, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp],
    "bez_ppo_policy_backward": [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp],
    "bez_ppo_scatter_f16": [_vp, _vp, _i64, _vp, _vp],
    "bez_ppo_scatter2_f16": [_vp, _vp, _vp, _i64, _vp, _vp],
    "bez_ppo_adaptive_lr": [_vp, _vp, _f, _f, _f, _vp],
    "bez_ppo_gae": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _f, _f, _vp, _vp, _vp, _vp, _f, _vp],
    "bez_ppo_head_grads_f16": [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp],
    "bez_ppo_policy_backward_with_loss": [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp],
    "bez_ppo_dataset_prep": [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _vp],
    "bez_ppo_dataset_prep_staged": [_i32, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _vp],
    "bez_ppo_wgrad_sum": [_vp, _i32, _i64, _vp, _i32, _vp],
    "bez_ppo_wgrad_plan": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp],
    "bez_ppo_wgrad_run": [_vp, _vp, _i32, _vp],
    "bez_ppo_colsum_f16": [_vp, _i64, _i32, _vp, _i32, _vp],
    "bez_ppo_elu_bwd_colsum_f16": [_vp, _vp, _vp, _i64, _i32, _vp, _i32, _vp],
    "bez_ppo_adam_step": [_vp, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _f, _f, _f, _f, _f, _vp, _vp, _f, _f, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp],
    "bez_ppo_grad_reduce_all": [_vp, _vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _vp],
    "bez_ppo_grad_reduce_blocks": [_vp, _i32, _vp, _i32],
    "bez_ppo_grad_norm_parts": [_vp, _i64, _vp, _i32, _vp],
    "bez_ppo_adam_grid_capacity": [_i64, _vp, _vp],
}
_lib = None
PPO_ABI_VERSION = 9   # BEZ_PPO_ABI_VERSION (include/bez_sim.h)


def lib():
    global _lib
    if _lib is None:
        l = load_library()
        l.bez_ppo_abi_version.restype, l.bez_ppo_abi_version.argtypes = C.c_int32, []
        got = int(l.bez_ppo_abi_version())
        if got != PPO_ABI_VERSION:
            raise RuntimeError("libbez_sim.so speaks PPO ABI %d, this binding %d: rebuild (python -m bez_isaacgym_amd.build)" % (got, PPO_ABI_VERSION))
        for name, args in _SIGS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = C.c_int, args
        _lib = l
    return _lib


def _p(t, dtype=torch.float32):
    assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return C.c_void_p(t.data_ptr())


def _pv(t, dtype=torch.float32):
    """device pointer of a tensor that may be a strided view (the callee is told the strides)"""
    assert t.is_cuda and t.dtype == dtype, t.dtype
    return C.c_void_p(t.data_ptr())


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d)" % (what, rc))


class FusedRunningMeanStd:
    """The three kernels behind RunningMeanStd.forward (train: moments -> [all-reduce] -> apply -> normalise)."""

    def __init__(self, rms, reduce_fn=None):
        self.rms = rms
        d = rms.running_mean.numel()
        self.d = d
        self.mom = torch.zeros(2 * d + 1, dtype=torch.float64, device=rms.running_mean.device)
        # per-workgroup partials + ticket counter of the fixed-order (bit-reproducible) column sums: 1 + 1024 * 2 * d doubles, zeroed once
        self.scratch = torch.zeros(1 + 1024 * 2 * d, dtype=torch.float64, device=rms.running_mean.device)
        self.reduce_fn = reduce_fn  # e.g. dist.all_reduce for data-parallel training

    def moments(self, x, out=None):
        """[column sums | sums of squares | rows] of x into `out` (default: the object's own buffer)"""
        rows = x.numel() // self.d
        out = self.mom if out is None else out
        assert out.numel() == 2 * self.d + 1 and out.is_contiguous()
        _chk(lib().bez_ppo_rms_moments(_p(x), rows, self.d, _p(out, torch.float64), _p(self.scratch, torch.float64), _stream(x)), "bez_ppo_rms_moments")

    def apply(self, mom=None):
        r = self.rms
        mom = self.mom if mom is None else mom
        _chk(lib().bez_ppo_rms_apply(_p(mom, torch.float64), self.d, _p(r.running_mean, torch.float64), _p(r.running_var, torch.float64),
                                     _p(r.count.view(1), torch.float64), _stream(mom)), "bez_ppo_rms_apply")

    def update(self, x):
        self.moments(x)
        if self.reduce_fn is not None:
            self.reduce_fn(self.mom)
        self.apply()

    def normalize(self, x, out):
        rows = x.numel() // self.d
        r = self.rms
        _chk(lib().bez_ppo_rms_normalize(_p(x), rows, self.d, _p(r.running_mean, torch.float64), _p(r.running_var, torch.float64),
                                         float(r.epsilon), C.c_void_p(out.data_ptr()), 1 if out.dtype == torch.float16 else 0, _stream(x)),
             "bez_ppo_rms_normalize")
        return out


def sample(mu, logstd, noise, actions, env_actions, neglogp, sigma):
    n, a = mu.shape
    _chk(lib().bez_ppo_sample(_p(mu), _p(logstd), _p(noise), n, a, _p(actions), _p(env_actions), _p(neglogp), _p(sigma), _stream(mu)), "bez_ppo_sample")


class RolloutLayout(C.Structure):
    """BezPpoRolloutLayout (include/bez_sim.h): row strides (floats) of the rollout rows PolicyForward.rollout_step writes"""
    _fields_ = [("obs_row_stride", C.c_int64), ("act_row_stride", C.c_int64), ("scalar_stride", C.c_int64)]


class ActionNoise(C.Structure):
    """BezPpoActionNoise (include/bez_sim.h): env.action_noise_source() for the launch that adds the env's action noise itself"""
    _fields_ = [("snap_dev", C.c_void_p), ("seed", C.c_uint64), ("env_id_offset", C.c_int64)]


class RolloutPost(C.Structure):
    """BezPpoRolloutPost (include/bez_sim.h): the arguments of rollout_post() for the launch that carries them (PolicyForward.rollout_step)"""
    _fields_ = [("rew", C.c_void_p), ("reset", C.c_void_p), ("timeouts", C.c_void_p), ("prev_values", C.c_void_p), ("reward_scale", C.c_float),
                ("gamma", C.c_float), ("bootstrap", C.c_int32), ("shaped", C.c_void_p), ("dones_f", C.c_void_p), ("cur_rew", C.c_void_p),
                ("cur_len", C.c_void_p), ("ep_stats", C.c_void_p), ("ep_parts", C.c_void_p)]

    @staticmethod
    def parts_numel(n):
        """doubles of the per-workgroup slots (ep_parts) for launches of up to n rows: 4 per workgroup, the smallest tile has 32 rows"""
        return 4 * ((n + 31) // 32 + 1)

    @classmethod
    def of(cls, rew, dones, timeouts, values, reward_scale, gamma, bootstrap, shaped, dones_f, cur_rew, cur_len, ep_stats, ep_parts=None):
        """ep_parts (fp64, parts_numel(n) zero-initialised elements): the launch adds each workgroup's finished-episode sums to that workgroup's own
        slot instead of adding to ep_stats with atomics; the caller folds the slots into ep_stats (fold_episode_parts) once per rollout."""
        n = rew.numel()
        assert dones.numel() == timeouts.numel() == values.numel() == shaped.numel() == dones_f.numel() == cur_rew.numel() == cur_len.numel() == n
        if ep_parts is not None:
            assert ep_parts.dtype == torch.float64 and ep_parts.is_contiguous() and ep_parts.numel() >= cls.parts_numel(n)
        return cls(_p(rew).value, _p(dones, torch.int64).value, _p(timeouts, torch.int64).value, _p(values).value, float(reward_scale), float(gamma),
                   1 if bootstrap else 0, _p(shaped).value, _p(dones_f).value, _p(cur_rew).value, _p(cur_len).value, _p(ep_stats, torch.float64).value,
                   None if ep_parts is None else _p(ep_parts, torch.float64).value)


def fold_episode_parts(ep_stats, ep_parts):
    """ep_stats[0:3] += the per-workgroup slots' column sums (fixed order), slots cleared: once per rollout, behind its last policy launch"""
    ep_stats.add_(ep_parts.view(-1, 4)[:, :3].sum(0))
    ep_parts.zero_()


def rollout_post(rew, dones, timeouts, values, reward_scale, gamma, bootstrap, shaped, dones_f, cur_rew, cur_len, ep_stats, ep_parts=None):
    """ep_parts: the slots RolloutPost.of(ep_parts=) launches filled -- one extra workgroup of this launch adds them to ep_stats and clears them"""
    n = rew.numel()
    if ep_parts is not None:
        _chk(lib().bez_ppo_rollout_post_fold(_p(rew), _p(dones, torch.int64), _p(timeouts, torch.int64), _p(values), n, float(reward_scale), float(gamma),
                                             1 if bootstrap else 0, _p(shaped), _p(dones_f), _p(cur_rew), _p(cur_len), _p(ep_stats, torch.float64),
                                             _p(ep_parts, torch.float64), ep_parts.numel() // 4, _stream(rew)), "bez_ppo_rollout_post_fold")
        return
    _chk(lib().bez_ppo_rollout_post(_p(rew), _p(dones, torch.int64), _p(timeouts, torch.int64), _p(values), n, float(reward_scale), float(gamma),
                                    1 if bootstrap else 0, _p(shaped), _p(dones_f), _p(cur_rew), _p(cur_len), _p(ep_stats, torch.float64), _stream(rew)),
         "bez_ppo_rollout_post")


def loss(mu, logstd, value, mb, e_clip, critic_coef, entropy_coef, bounds_coef, clip_value, scale, gmu, gval, glog, stats, zero_glog=True,
         zero_stats=True, update_mu_sigma=False, scratch=None, defer_reduce=False):
    """stats[5] <- sums of a_loss, c_loss, b_loss, kl, entropy (zero_stats False: the caller cleared them); gmu / gval <- gradient of the
    mean loss (x loss scale); glog is ACCUMULATED into (zero_glog: cleared first); update_mu_sigma: mb["mu"] / mb["sigma"] are
    overwritten with the current mu / exp(logstd) once the KL against the old ones is taken; scratch (loss_scratch(b, a), zeroed once):
    the sums are added in a fixed order instead of with float atomics (bit-reproducible).  defer_reduce: only the per-workgroup partials
    are written (into scratch); grad_reduce_all() then WRITES glog / stats together with the step's other gradient reductions."""
    b, a = mu.shape
    assert scratch is not None or not defer_reduce
    _chk(lib().bez_ppo_loss(_p(mu), _p(logstd), _p(value), _p(mb["actions"]), _p(mb["old_logp"]), _p(mb["advantages"]), _p(mb["old_values"]),
                            _p(mb["returns"]), _p(mb["mu"]), _p(mb["sigma"]), b, a, float(e_clip), float(critic_coef), float(entropy_coef),
                            float(bounds_coef), (1 if clip_value else 0) | (0 if zero_glog else 2) | (0 if zero_stats else 4) | (8 if update_mu_sigma else 0) | (16 if defer_reduce else 0), None if scale is None else _p(scale), _p(gmu),
                            _p(gval), _p(glog), _p(stats), None if scratch is None else _p(scratch), _stream(mu)), "bez_ppo_loss")


class LossOperands(C.Structure):
    """BezPpoLossOperands (include/bez_sim.h): the operands of loss() for PolicyBackward.with_loss()"""
    _fields_ = [(k, C.c_void_p) for k in ("mu_dev", "logstd_dev", "value_dev", "actions_dev", "old_logp_dev", "adv_dev", "old_value_dev", "returns_dev",
                                           "old_mu_dev", "old_sigma_dev")] + \
               [("e_clip", C.c_float), ("critic_coef", C.c_float), ("entropy_coef", C.c_float), ("bounds_coef", C.c_float), ("clip_value", C.c_int32),
                ("loss_scale_dev", C.c_void_p), ("scratch_dev", C.c_void_p)]

    @classmethod
    def of(cls, mu, logstd, value, mb, e_clip, critic_coef, entropy_coef, bounds_coef, clip_value, scale, update_mu_sigma, scratch):
        o = cls()
        for k, t in (("mu_dev", mu), ("logstd_dev", logstd), ("value_dev", value), ("actions_dev", mb["actions"]), ("old_logp_dev", mb["old_logp"]),
                     ("adv_dev", mb["advantages"]), ("old_value_dev", mb["old_values"]), ("returns_dev", mb["returns"]), ("old_mu_dev", mb["mu"]),
                     ("old_sigma_dev", mb["sigma"]), ("scratch_dev", scratch)):
            setattr(o, k, _p(t).value)
        o.loss_scale_dev = None if scale is None else _p(scale).value
        o.e_clip, o.critic_coef, o.entropy_coef, o.bounds_coef = float(e_clip), float(critic_coef), float(entropy_coef), float(bounds_coef)
        o.clip_value = (1 if clip_value else 0) | (8 if update_mu_sigma else 0)
        return o


def loss_scratch(b, a, device):
    """zeroed scratch for loss(..., scratch=): 2 + ceil(b / 64) * (a + 5) floats"""
    return torch.zeros(2 + ((b + 63) // 64) * (a + 5), device=device, dtype=torch.float32)


ADAM_WORK_FLOATS = 258   # BEZ_PPO_ADAM_WORK_FLOATS
ADAM_GRIDNORM_FLOATS = 516   # BEZ_PPO_ADAM_GRIDNORM_FLOATS


def adam_grid_fits(n):
    """True when every workgroup of the optimiser launch for n parameters is resident at once on the current device -- the condition of the
    in-launch gradient norm (AdamExtra.grid_norm_dev: the workgroups meet at a counter); bez_ppo_adam_step refuses the launch (-6) otherwise."""
    cap, g = C.c_int32(0), C.c_int32(0)
    rc = lib().bez_ppo_adam_grid_capacity(int(n), C.addressof(cap), C.addressof(g))
    if rc != 0:
        raise RuntimeError("bez_ppo_adam_grid_capacity failed (%d)" % rc)
    return g.value <= cap.value


def adam_step(params, grads, exp_avg, exp_avg_sq, steps, lr, betas, eps, weight_decay, max_norm, scale, growth_tracker, growth_factor,
              backoff_factor, growth_interval, work, params_f16=None, tail=(), adapt=None, packed=None, next_rms=None, norm_parts=None, grad_div=1.0,
              grid_norm=None):
    """unscale + clip + Adam + scaler update on the flat buffers in ONE launch (csrc/bez_ppo.hip adam_fused_kernel); scale / growth_tracker
    None = no AMP; params_f16 (flat fp16, same layout) receives the updated parameters in the same pass.  `work` must be zero on entry and
    is zero again afterwards.  tail: up to 4 (dst, src, scale) with one-element fp32 tensors: dst += src * scale.
    adapt = (kl, threshold, min_lr, max_lr): the AdaptiveScheduler rule moves `lr` on that one-element KL after the step.
    packed (PackedWeights): its fragment-major copies are written in the same pass (instead of a refresh() launch before the next forward);
    next_rms = (FusedRunningMeanStd, moments): the input normaliser absorbs the NEXT minibatch's moments here (instead of an apply() launch);
    norm_parts: the (blocks, 2) tensor grad_reduce_all() / grad_norm_parts() filled for THIS gradient (norm and non-finite count are then not
    re-derived); grad_div: the buffer holds the all-reduced SUM over that many ranks (the division rides in the unscale factor)."""
    n = params.numel()
    extra = None
    if packed is not None or next_rms is not None or norm_parts is not None or grad_div != 1.0 or grid_norm is not None:
        extra = AdamExtra()
        extra.grad_div = float(grad_div)
        if grid_norm is not None:   # (ADAM_GRIDNORM_FLOATS zero-initialised floats: the launch forms the norm itself, workgroups meeting at a counter)
            assert norm_parts is None and grid_norm.dtype == torch.float32 and grid_norm.is_contiguous() and grid_norm.numel() >= ADAM_GRIDNORM_FLOATS
            extra.grid_norm_dev = grid_norm.data_ptr()
        if norm_parts is not None:
            assert norm_parts.dtype == torch.float32 and norm_parts.is_contiguous() and norm_parts.shape[1] == 2
            extra.norm_parts_dev, extra.norm_parts = norm_parts.data_ptr(), norm_parts.shape[0]
        if packed is not None:
            assert params_f16 is not None and packed.hflat.data_ptr() == params_f16.data_ptr()
            extra.map_a_dev, extra.map_b_dev, extra.packed_f16_dev = packed.map_a.data_ptr(), packed.map_b.data_ptr(), packed.flat.data_ptr()
        if next_rms is not None:
            f, mom = next_rms
            r = f.rms
            assert mom.dtype == torch.float64 and mom.numel() == 2 * f.d + 1 and mom.is_contiguous()
            extra.rms_moments_dev, extra.rms_cols = mom.data_ptr(), f.d
            extra.rms_mean_dev, extra.rms_var_dev, extra.rms_count_dev = r.running_mean.data_ptr(), r.running_var.data_ptr(), r.count.data_ptr()
    assert work.numel() >= ADAM_WORK_FLOATS, "work: BEZ_PPO_ADAM_WORK_FLOATS zero-initialised floats"
    nt = len(tail)
    assert nt <= 4 and all(d.numel() == 1 and x.numel() == 1 for d, x, _ in tail)
    td = (C.c_void_p * 4)(*[_p(d).value for d, _, _ in tail])
    ts = (C.c_void_p * 4)(*[_p(x).value for _, x, _ in tail])
    tsc = (C.c_float * 4)(*[float(a) for _, _, a in tail])
    _chk(lib().bez_ppo_adam_step(_p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), n, _p(steps), steps.numel(), _p(lr), float(betas[0]),
                                 float(betas[1]), float(eps), float(weight_decay), float(max_norm), None if scale is None else _p(scale),
                                 None if growth_tracker is None else _p(growth_tracker, torch.int32), float(growth_factor), float(backoff_factor),
                                 int(growth_interval), _p(work), None if params_f16 is None else _p(params_f16, torch.float16), nt, td, ts, tsc,
                                 None if adapt is None else _p(adapt[0]), float(adapt[1]) if adapt else 0.0, float(adapt[2]) if adapt else 0.0,
                                 float(adapt[3]) if adapt else 0.0, None if extra is None else C.byref(extra), _stream(params)),
         "bez_ppo_adam_step")


def grad_norm_parts(grads, parts):
    """Per-workgroup (sum g^2, non-finite count) of the flat gradient `grads` into `parts` ((>= blocks, 2) fp32): what grad_reduce_all(norm_parts=)
    leaves, re-formed after an all-reduce replaced the gradient.  Returns the view of the rows written (adam_step(norm_parts=))."""
    assert parts.dtype == torch.float32 and parts.is_contiguous() and parts.dim() == 2 and parts.shape[1] == 2
    rc = lib().bez_ppo_grad_norm_parts(_p(grads), grads.numel(), _p(parts), parts.shape[0], _stream(grads))
    if rc <= 0:
        raise RuntimeError("bez_ppo_grad_norm_parts failed (%d)" % rc)
    return parts[:rc]


class AdamExtra(C.Structure):
    """BezPpoAdamExtra (include/bez_sim.h)"""
    _fields_ = [("norm_parts_dev", C.c_void_p), ("norm_parts", C.c_int32), ("grad_div", C.c_float), ("grid_norm_dev", C.c_void_p), ("map_a_dev", C.c_void_p), ("map_b_dev", C.c_void_p), ("packed_f16_dev", C.c_void_p), ("rms_moments_dev", C.c_void_p),
                ("rms_cols", C.c_int32), ("rms_mean_dev", C.c_void_p), ("rms_var_dev", C.c_void_p), ("rms_count_dev", C.c_void_p)]


def grad_reduce_blocks(wgrad, bwd):
    """rows of the (blocks, 2) fp32 tensor grad_reduce_all(..., norm_parts=) fills"""
    nb = lib().bez_ppo_grad_reduce_blocks(wgrad.plan_host, bwd.nh, C.cast(bwd.c_widths, C.c_void_p), bwd.A)
    assert nb > 0
    return nb


def grad_reduce_all(wgrad, bwd, bias_grads, mu_bias_grad, value_bias_grad, rows, loss_scratch_buf, glog, stats, accumulate=False, norm_parts=None):
    """The step's three second-stage reductions in one launch: the split-K images of `wgrad` (WgradMfma called with reduce=False), the bias
    column sums of `bwd` (PolicyBackward called with defer_reduce=True) and the loss kernel's per-workgroup sums (loss(defer_reduce=True)).
    accumulate False: every weight / bias / log-sigma gradient and the five statistics are WRITTEN (no clear needed in front of the step);
    norm_parts (grad_reduce_blocks(...), 2) then receives each workgroup's share of sum g^2 / non-finite count for adam_step(norm_parts=)."""
    assert norm_parts is None or (not accumulate and norm_parts.shape == (grad_reduce_blocks(wgrad, bwd), 2) and norm_parts.is_contiguous())
    nh = bwd.nh
    t_b = (C.c_void_p * nh)(*[b.data_ptr() for b in bias_grads])
    _chk(lib().bez_ppo_grad_reduce_all(wgrad.plan_host, C.c_void_p(wgrad.plan_dev.data_ptr()), _p(bwd._partial), rows, nh, C.cast(bwd.c_widths, C.c_void_p),
                                       bwd.A, C.cast(t_b, C.c_void_p), _p(mu_bias_grad), _p(value_bias_grad), _p(loss_scratch_buf), rows, _p(glog), _p(stats),
                                       1 if accumulate else 0, None if norm_parts is None else _p(norm_parts), _stream(glog)), "bez_ppo_grad_reduce_all")


def wgrad_sum(partials, out, accumulate=False):
    """out (fp32, any shape with n elements) <- (+=) sum over dim 0 of partials (fp16, [S, ...n elements])."""
    s_, n = partials.shape[0], out.numel()
    assert partials.numel() == s_ * n and out.is_contiguous()
    _chk(lib().bez_ppo_wgrad_sum(_p(partials, torch.float16), s_, n, _p(out), 1 if accumulate else 0, _stream(out)), "bez_ppo_wgrad_sum")


class WgradMfma:
    """dW_L (+)= dY_L^T X_L for all Linear layers of the MLP in ONE split-K MFMA launch + one deterministic reduction
    (csrc/bez_wgrad.hip).  The work is laid out once per set of tensors (bez_ppo_wgrad_plan), the plan copied to the device once;
    `ok` is False when the kernel does not take the shapes (the caller then keeps its GEMM path)."""
    NSPLIT = int(__import__("os").environ.get("BEZ_WGRAD_NSPLIT", "40"))   # images of every gradient the scratch holds = the most K-splits a block may get
    PLAN_BYTES = 3072

    def __init__(self, dys, xs, grads):
        n = len(dys)
        rows = dys[0].shape[0]
        assert all(t.dtype == torch.float16 and t.is_contiguous() and t.shape[0] == rows for t in list(dys) + list(xs))
        assert all(g.dtype == torch.float32 and g.is_contiguous() and g.shape == (dy.shape[1], x.shape[1]) for g, dy, x in zip(grads, dys, xs))
        self.keep = (list(dys), list(xs), list(grads))
        dy = (C.c_void_p * n)(*[t.data_ptr() for t in dys]); x = (C.c_void_p * n)(*[t.data_ptr() for t in xs])
        dw = (C.c_void_p * n)(*[t.data_ptr() for t in grads])
        of = (C.c_int32 * n)(*[t.shape[1] for t in dys]); inf = (C.c_int32 * n)(*[t.shape[1] for t in xs])
        total = sum(g.numel() for g in grads)
        dev = dys[0].device
        self.partial = torch.empty(self.NSPLIT * total, device=dev, dtype=torch.float32)
        self.plan_host = (C.c_uint8 * self.PLAN_BYTES)()
        rc = lib().bez_ppo_wgrad_plan(dy, x, of, inf, dw, n, rows, self.NSPLIT, _p(self.partial), self.plan_host) if rows % 64 == 0 else -3
        self.ok = rc == 0
        if rc not in (0, -3):
            _chk(rc, "bez_ppo_wgrad_plan")
        if self.ok:   # one synchronous upload, outside any graph capture
            self.plan_dev = torch.frombuffer(bytearray(self.plan_host), dtype=torch.uint8).to(dev)
        else:
            self.partial = None

    def matches(self, dys, xs, grads):
        # the plan bakes raw device pointers AND the geometry (rows, widths, strides): all of it must still hold
        k = self.keep
        new, old = list(dys) + list(xs) + list(grads), k[0] + k[1] + k[2]
        return len(new) == len(old) and all(a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride() and a.dtype == b.dtype
                                            for a, b in zip(new, old))

    def __call__(self, accumulate=True, reduce=True):
        """reduce False: the split-K partial images only; grad_reduce_all() adds them (with the step's other reductions)"""
        if not self.ok:
            return False
        _chk(lib().bez_ppo_wgrad_run(self.plan_host, C.c_void_p(self.plan_dev.data_ptr()), (1 if accumulate else 0) if reduce else 2, _stream(self.partial)),
             "bez_ppo_wgrad_run")
        return True


def colsum_f16(y, out, accumulate=False):
    """out (fp32, D) <- (+=) column sums of y (fp16, (B, D))."""
    b, d = y.shape
    assert out.numel() == d and out.is_contiguous()
    _chk(lib().bez_ppo_colsum_f16(_p(y, torch.float16), b, d, _p(out), 1 if accumulate else 0, _stream(out)), "bez_ppo_colsum_f16")


def rollout_pre(mu, value, logstd, noise, obs, dones, value_rms, mb_obs, mb_dones, mb_mu, mb_val, act, env_act, neglogp, sigma):
    """fp32 rows of the rollout buffers (obs, dones, mu, de-normalised value) + action sampling, one launch; mu / value fp16 or fp32."""
    n, a = mu.shape
    half = mu.dtype == torch.float16
    assert value.dtype == mu.dtype and value.numel() == n and mu.is_contiguous() and value.is_contiguous()
    vm = None if value_rms is None else _p(value_rms.running_mean, torch.float64)
    vv = None if value_rms is None else _p(value_rms.running_var, torch.float64)
    _chk(lib().bez_ppo_rollout_pre(C.c_void_p(mu.data_ptr()), C.c_void_p(value.data_ptr()), 1 if half else 0, _p(logstd), _p(noise), _p(obs), _p(dones), vm, vv,
                                   0.0 if value_rms is None else float(value_rms.epsilon), n, a, obs.shape[1], _p(mb_obs), _p(mb_dones), _p(mb_mu), _p(mb_val),
                                   _p(act), _p(env_act), _p(neglogp), _p(sigma), _stream(mu)), "bez_ppo_rollout_pre")


class PolicyBackward:
    """One-launch input-gradient chain of the MLP's backward pass (csrc/bez_policy.hip policy_backward_kernel) + the transposed fp16
    weight copies it reads, refreshed from the flat fp16 working copy `hflat` by one scatter launch.  `layout` = [(offset in hflat,
    out, in)] of the weight of every Linear in network order (hidden layers, mu head, value head)."""

    def __init__(self, hflat, layout, num_actions, packed=None):
        dev = hflat.device
        nh = len(layout) - 2
        self.nh, self.A, self.hflat = nh, num_actions, hflat
        self.widths = [o for _, o, _ in layout[:nh]]
        assert all(32 <= w <= 416 and w % 2 == 0 for w in self.widths) and num_actions <= 31
        self.c_widths = (C.c_int32 * nh)(*self.widths)
        self.packed = packed
        if packed is not None:  # fragment-major transposed copies owned (and refreshed) by the shared PackedWeights
            self.flag, self.wht = 1, packed.bwd_heads
            self.c_wt = (C.c_void_p * nh)(*[None if w is None else w.data_ptr() for w in packed.bwd])
            return
        self.flag = 0
        last = self.widths[-1]
        sizes = [0] + [layout[i][1] * layout[i][2] for i in range(1, nh)]
        offs = np.cumsum([0] + sizes).tolist()           # wt[i] at offs[i] (i >= 1), the heads matrix behind them
        total = offs[nh] + last * 32
        self.flat_t = torch.zeros(total, device=dev, dtype=torch.float16)
        m = np.full(hflat.numel(), -1, dtype=np.int32)
        for i in range(1, nh):                               # W_i (out, in) -> W_i^T (in, out)
            off, o, k = layout[i]
            src = off + np.arange(o * k).reshape(o, k)
            m[src] = offs[i] + (np.arange(k)[None, :] * o + np.arange(o)[:, None])
        off_mu, a, k = layout[nh]
        assert a == num_actions and k == last and layout[nh + 1][1] == 1 and layout[nh + 1][2] == last
        m[off_mu + np.arange(a * k).reshape(a, k)] = offs[nh] + (np.arange(k)[None, :] * 32 + np.arange(a)[:, None])
        m[layout[nh + 1][0] + np.arange(k)] = offs[nh] + np.arange(k) * 32 + a
        self.map = torch.from_numpy(m).to(dev)
        self.wt = [None] + [self.flat_t[offs[i]:offs[i] + sizes[i]] for i in range(1, nh)]
        self.wht = self.flat_t[offs[nh]:]
        self.c_widths = (C.c_int32 * nh)(*self.widths)
        self.c_wt = (C.c_void_p * nh)(*[None if w is None else w.data_ptr() for w in self.wt])

    def refresh(self):
        if self.packed is not None:
            return self.packed.refresh()
        _chk(lib().bez_ppo_scatter_f16(_p(self.hflat, torch.float16), C.c_void_p(self.map.data_ptr()), self.hflat.numel(), _p(self.flat_t, torch.float16),
                                       _stream(self.hflat)), "bez_ppo_scatter_f16")

    def __call__(self, gmu, gval, acts, gz, gmu16, gv16, bias_grads, mu_bias_grad, value_bias_grad, defer_reduce=False):
        """defer_reduce: the per-workgroup column sums only (self._partial); grad_reduce_all() writes the bias gradients"""
        n = gmu.shape[0]
        assert gmu.shape == (n, self.A) and gval.numel() == n and len(acts) == len(gz) == len(bias_grads) == self.nh
        for a, z, b, w in zip(acts, gz, bias_grads, self.widths):
            assert a.shape == (n, w) and z.shape == (n, w) and a.dtype == z.dtype == torch.float16 and a.is_contiguous() and z.is_contiguous()
            assert b.dtype == torch.float32 and b.numel() == w and b.is_contiguous()
        t_act = (C.c_void_p * self.nh)(*[a.data_ptr() for a in acts])
        t_gz = (C.c_void_p * self.nh)(*[z.data_ptr() for z in gz])
        t_b = (C.c_void_p * self.nh)(*[b.data_ptr() for b in bias_grads])
        need = ((n + 63) // 64) * (sum(self.widths) + 32)
        if getattr(self, "_partial", None) is None or self._partial.numel() < need:
            self._partial = torch.empty(need, device=gmu.device, dtype=torch.float32)
        _chk(lib().bez_ppo_policy_backward(_p(gmu), _p(gval), n, self.nh, C.cast(self.c_widths, C.c_void_p), self.A, C.cast(t_act, C.c_void_p),
                                           C.cast(self.c_wt, C.c_void_p), C.c_void_p(self.wht.data_ptr()), C.cast(t_gz, C.c_void_p), _p(gmu16, torch.float16),
                                           _p(gv16, torch.float16), C.cast(t_b, C.c_void_p), _p(mu_bias_grad), _p(value_bias_grad), _p(self._partial), self.flag | (2 if defer_reduce else 0),
                                           _stream(gmu)),
             "bez_ppo_policy_backward")


def _policy_backward_with_loss(self, loss_ops, acts, gz, gmu16, gv16, bias_grads, mu_bias_grad, value_bias_grad):
    """loss() + __call__(defer_reduce=True) as ONE launch (bez_ppo_policy_backward_with_loss): the tile's loss terms and d loss / d mu, d loss / d value
    are formed in front of the backward chain and never leave the chip.  False: a shape the fused kernel does not take (the caller runs the two)."""
    n = gmu16.shape[0]
    assert len(acts) == len(gz) == len(bias_grads) == self.nh and gmu16.shape == (n, self.A) and gv16.numel() == n
    t_act = (C.c_void_p * self.nh)(*[a.data_ptr() for a in acts])
    t_gz = (C.c_void_p * self.nh)(*[z.data_ptr() for z in gz])
    t_b = (C.c_void_p * self.nh)(*[b.data_ptr() for b in bias_grads])
    need = ((n + 63) // 64) * (sum(self.widths) + 32)
    if getattr(self, "_partial", None) is None or self._partial.numel() < need:
        self._partial = torch.empty(need, device=gmu16.device, dtype=torch.float32)
    rc = lib().bez_ppo_policy_backward_with_loss(C.byref(loss_ops), n, self.nh, C.cast(self.c_widths, C.c_void_p), self.A, C.cast(t_act, C.c_void_p),
                                                 C.cast(self.c_wt, C.c_void_p), C.c_void_p(self.wht.data_ptr()), C.cast(t_gz, C.c_void_p),
                                                 _p(gmu16, torch.float16), _p(gv16, torch.float16), C.cast(t_b, C.c_void_p), _p(mu_bias_grad),
                                                 _p(value_bias_grad), _p(self._partial), self.flag | 2, _stream(gmu16))
    if rc == -3:
        return False
    _chk(rc, "bez_ppo_policy_backward_with_loss")
    return True


PolicyBackward.with_loss = _policy_backward_with_loss


def adaptive_lr(lr, kl, kl_threshold, min_lr, max_lr):
    """AdaptiveScheduler.update on the device-resident lr (0-dim / 1-element fp32 tensors)."""
    assert lr.numel() == 1 and kl.numel() == 1
    _chk(lib().bez_ppo_adaptive_lr(_p(lr), _p(kl), float(kl_threshold), float(min_lr), float(max_lr), _stream(lr)), "bez_ppo_adaptive_lr")


def gae(rewards, values, mb_dones, dones, last_values, gamma, tau, advs, returns=None, unnorm=None):
    """GAE backward scan: rewards / values / mb_dones (H,N[,1]) fp32, dones / last_values (N[,1]); writes advs (and returns = advs + values).
    unnorm (a RunningMeanStd): last_values are the network's normalised outputs, de-normalised inside the launch."""
    h = rewards.shape[0]
    n = rewards.numel() // h
    assert values.numel() == h * n and mb_dones.numel() == h * n and dones.numel() == n and last_values.numel() == n and advs.numel() == h * n
    _chk(lib().bez_ppo_gae(_p(rewards), _p(values), _p(mb_dones), _p(dones), _p(last_values), h, n, float(gamma), float(tau), _p(advs),
                           None if returns is None else _p(returns), None if unnorm is None else _p(unnorm.running_mean, torch.float64),
                           None if unnorm is None else _p(unnorm.running_var, torch.float64), 0.0 if unnorm is None else float(unnorm.epsilon),
                           _stream(rewards)), "bez_ppo_gae")


def dataset_prep_scratch(num_minibatches, horizon, num_envs, device):
    return torch.zeros((num_minibatches + 2) * 256 * 128 + 2 * ((horizon * num_envs + 255) // 256), device=device, dtype=torch.float64)


def dataset_prep(obs, minibatch_rows, num_minibatches, obs_moments, values, returns, value_rms, value_moments, return_moments, old_values, ds_returns,
                 advantages, normalize_advantage, scratch, stages=7, adv_sums=None):
    """prepare_dataset + the per-minibatch observation moments in four launches (bez_ppo_dataset_prep).  obs: the dataset's (rows, D) fp32
    observations (None: no observation moments); values / returns (H, N[, 1]) fp32 in the rollout's layout; value_rms: the RunningMeanStd of
    the values (None: values / returns are used as they are); outputs env-major (N * H[, 1]).  False: shapes the kernel does not take.
    stages / adv_sums (data parallel, bez_ppo_dataset_prep_staged): 1 = local moments, 2 = values / returns / advantages + adv_sums (3 fp64:
    sum, sum of squares, count of this rank's advantages), 4 = advantage normalisation from adv_sums; the caller all-reduces between them."""
    h = values.shape[0]
    n = values.numel() // h
    assert returns.numel() == h * n and old_values.numel() == h * n and ds_returns.numel() == h * n and advantages.numel() == h * n
    assert values.is_contiguous() and returns.is_contiguous() and old_values.is_contiguous() and ds_returns.is_contiguous() and advantages.is_contiguous()
    d = 0
    if obs is not None and num_minibatches > 0:
        d = obs.shape[1]
        assert obs.is_contiguous() and obs.shape[0] >= num_minibatches * minibatch_rows and obs_moments.is_contiguous() and obs_moments.numel() == num_minibatches * (2 * d + 1)
    else:
        num_minibatches = 0
    r = value_rms
    assert adv_sums is None or (adv_sums.is_contiguous() and adv_sums.numel() >= 3)
    rc = lib().bez_ppo_dataset_prep_staged(int(stages), None if not num_minibatches else _p(obs), minibatch_rows, num_minibatches, d,
                                           None if not num_minibatches else _p(obs_moments, torch.float64), _p(values), _p(returns), h, n,
                                           None if r is None else _p(r.running_mean, torch.float64), None if r is None else _p(r.running_var, torch.float64),
                                           None if r is None else _p(r.count.view(1), torch.float64), 0.0 if r is None else float(r.epsilon),
                                           _p(value_moments, torch.float64), _p(return_moments, torch.float64), _p(old_values), _p(ds_returns), _p(advantages),
                                           1 if normalize_advantage else 0, None if adv_sums is None else _p(adv_sums, torch.float64),
                                           _p(scratch, torch.float64), scratch.numel(), _stream(values))
    if rc == -3:
        return False
    _chk(rc, "bez_ppo_dataset_prep")
    return True


def head_grads_f16(gmu, gval, gmu16, gv16, mu_bias_grad, value_bias_grad):
    """fp16 copies of the loss gradients w.r.t. mu (B,A) / value (B,1) + their column sums added to the head bias gradients."""
    b, a = gmu.shape
    assert gval.numel() == b and gmu16.shape == (b, a) and gv16.numel() == b and mu_bias_grad.numel() == a and value_bias_grad.numel() == 1
    _chk(lib().bez_ppo_head_grads_f16(_p(gmu), _p(gval), b, a, _p(gmu16, torch.float16), _p(gv16, torch.float16), _p(mu_bias_grad), _p(value_bias_grad),
                                      _stream(gmu)), "bez_ppo_head_grads_f16")


def elu_bwd_colsum_f16(gy, y, gz, bias_grad, accumulate=False):
    """gz (fp16) <- gy * elu'(y) (y = the ELU output); bias_grad (fp32, D) <- (+=) column sums of gz."""
    b, d = y.shape
    assert gy.shape == y.shape == gz.shape and gy.is_contiguous() and y.is_contiguous() and gz.is_contiguous() and bias_grad.numel() == d
    _chk(lib().bez_ppo_elu_bwd_colsum_f16(_p(gy, torch.float16), _p(y, torch.float16), _p(gz, torch.float16), b, d, _p(bias_grad), 1 if accumulate else 0,
                                          _stream(y)), "bez_ppo_elu_bwd_colsum_f16")


def _pack_index(out, k):
    """position of element (n, c) of a Linear (out, k) in its fragment-major copy (include/bez_sim.h, weights_packed) and the copy's size"""
    ks = (k + 15) // 16
    n, c = np.arange(out)[:, None], np.arange(k)[None, :]
    idx = (((n // 32) * ks + c // 16) * 64 + ((c // 8) % 2) * 32 + n % 32) * 8 + c % 8
    return idx.astype(np.int64), ((out + 31) // 32) * ks * 512


class PackedWeights:
    """Fragment-major fp16 copies of every weight the MFMA policy kernels read, kept current from the flat fp16 working copy `hflat` by
    ONE scatter launch (bez_ppo_scatter2_f16: each source element has a destination in the forward set and one in the backward set).
    `layout` = [(offset in hflat, out, in)] of the Linears in network order: hidden layers, mu head, value head."""

    def __init__(self, hflat, layout, num_actions):
        nh, a = len(layout) - 2, num_actions
        self.hflat, self.nh, self.A = hflat, nh, a
        last = layout[nh - 1][1]
        assert layout[nh][1] == a and layout[nh][2] == last and layout[nh + 1][1] == 1 and layout[nh + 1][2] == last and a <= 31
        ma = np.full(hflat.numel(), -1, dtype=np.int64)   # forward copies
        mb = np.full(hflat.numel(), -1, dtype=np.int64)   # backward (transposed) copies
        pos, fwd_off, bwd_off = 0, [], [None]
        for off, o, k in layout[:nh]:                    # forward: W_i as it is
            idx, size = _pack_index(o, k)
            ma[off + np.arange(o * k).reshape(o, k)] = pos + idx
            fwd_off.append(pos); pos += size
        idx, size = _pack_index(a + 1, last)             # forward heads: [W_mu; W_value]
        ma[layout[nh][0] + np.arange(a * last).reshape(a, last)] = pos + idx[:a]
        ma[layout[nh + 1][0] + np.arange(last)] = pos + idx[a]
        fwd_heads = pos; pos += size
        for off, o, k in layout[1:nh]:                   # backward: W_i^T = Linear (out' = in, in' = out)
            idx, size = _pack_index(k, o)                # idx[c, n] = position of element (c, n) of W^T, i.e. of W[n, c]
            mb[off + np.arange(o * k).reshape(o, k)] = pos + idx.T
            bwd_off.append(pos); pos += size
        idx, size = _pack_index(last, 32)                # backward heads: (out' = last, in' = 32) = [W_mu; W_value]^T
        mb[layout[nh][0] + np.arange(a * last).reshape(a, last)] = pos + idx[:, :a].T
        mb[layout[nh + 1][0] + np.arange(last)] = pos + idx[:, a]
        bwd_heads = pos; pos += size
        dev = hflat.device
        self.flat = torch.zeros(pos, device=dev, dtype=torch.float16)   # (padding stays zero)
        self.map_a = torch.from_numpy(ma.astype(np.int32)).to(dev)
        self.map_b = torch.from_numpy(mb.astype(np.int32)).to(dev)
        self.fwd = [self.flat[o:] for o in fwd_off]
        self.fwd_heads = self.flat[fwd_heads:]
        self.bwd = [None] + [self.flat[o:] for o in bwd_off[1:]]
        self.bwd_heads = self.flat[bwd_heads:]
        self.refresh()

    def refresh(self):
        _chk(lib().bez_ppo_scatter2_f16(_p(self.hflat, torch.float16), C.c_void_p(self.map_a.data_ptr()), C.c_void_p(self.map_b.data_ptr()),
                                        self.hflat.numel(), _p(self.flat, torch.float16), _stream(self.hflat)), "bez_ppo_scatter2_f16")


class PolicyForward:
    """One-launch rollout forward of the actor-critic MLP on its fp16 working weights (csrc/bez_policy.hip).  `hidden` = list of
    (w16, b16) of the ELU layers, then the mu and value heads; the pointer tables are built once (the tensors are static views)."""

    def __init__(self, hidden, mu_wb, value_wb, obs_rms, packed=None):
        self.keep = (hidden, mu_wb, value_wb, obs_rms)
        self.packed = packed  # PackedWeights: the kernels read the fragment-major copies (the caller keeps them refreshed)
        k = len(hidden)
        for w, b in list(hidden) + [mu_wb, value_wb]:
            assert w.dtype == torch.float16 and b.dtype == torch.float16 and w.is_contiguous() and b.is_contiguous() and w.is_cuda
        self.hw = (C.c_void_p * k)(*([w.data_ptr() for w, _ in hidden] if packed is None else [w.data_ptr() for w in packed.fwd]))
        self.flag = 0 if packed is None else 1
        self.mu_w = mu_wb[0] if packed is None else packed.fwd_heads
        self.hb = (C.c_void_p * k)(*[b.data_ptr() for _, b in hidden])
        self.widths = (C.c_int32 * k)(*[w.shape[0] for w, _ in hidden])
        self.k, self.d_in, self.num_actions = k, hidden[0][0].shape[1], mu_wb[0].shape[0]
        assert value_wb[0].shape[0] == 1 and mu_wb[0].shape[1] == hidden[-1][0].shape[0] == value_wb[0].shape[1]

    def __call__(self, obs, mu_out, value_out):
        hidden, mu_wb, value_wb, rms = self.keep
        n = obs.shape[0]
        assert obs.shape[1] == self.d_in and mu_out.shape == (n, self.num_actions) and value_out.numel() == n
        _chk(lib().bez_ppo_policy_forward(_p(obs), n, self.d_in, None if rms is None else _p(rms.running_mean, torch.float64),
                                          None if rms is None else _p(rms.running_var, torch.float64), 0.0 if rms is None else float(rms.epsilon), self.k,
                                          C.cast(self.hw, C.c_void_p), C.cast(self.hb, C.c_void_p), C.cast(self.widths, C.c_void_p),
                                          C.c_void_p(self.mu_w.data_ptr()), _p(mu_wb[1], torch.float16), self.num_actions, _p(value_wb[0], torch.float16),
                                          _p(value_wb[1], torch.float16), _p(mu_out), _p(value_out), self.flag, _stream(obs)), "bez_ppo_policy_forward")

    def train_forward(self, obs, x0, acts, mu_out, value_out):
        """Forward of a training minibatch that keeps the backward pass's operands: x0 (n, d_in) fp16, acts[i] (n, width_i) fp16."""
        hidden, mu_wb, value_wb, rms = self.keep
        n = obs.shape[0]
        assert obs.shape[1] == self.d_in and x0.shape == (n, self.d_in) and len(acts) == self.k and mu_out.shape == (n, self.num_actions) and value_out.numel() == n
        for a, (w, _) in zip(acts, hidden):
            assert a.shape == (n, w.shape[0]) and a.dtype == torch.float16 and a.is_contiguous()
        tab = (C.c_void_p * self.k)(*[a.data_ptr() for a in acts])
        _chk(lib().bez_ppo_policy_forward_train(
            _p(obs), n, self.d_in, None if rms is None else _p(rms.running_mean, torch.float64), None if rms is None else _p(rms.running_var, torch.float64),
            0.0 if rms is None else float(rms.epsilon), self.k, C.cast(self.hw, C.c_void_p), C.cast(self.hb, C.c_void_p), C.cast(self.widths, C.c_void_p),
            C.c_void_p(self.mu_w.data_ptr()), _p(mu_wb[1], torch.float16), self.num_actions, _p(value_wb[0], torch.float16), _p(value_wb[1], torch.float16),
            _p(x0, torch.float16), C.cast(tab, C.c_void_p), _p(mu_out), _p(value_out), self.flag, _stream(obs)), "bez_ppo_policy_forward_train")

    def rollout_step(self, obs, logstd, noise, dones, value_rms, mb_obs, mb_dones, mb_mu, mb_val, actions, env_actions, neglogp, sigma, prev_post=None, action_noise=None, dr_step=None):
        """Forward + everything up to the env step in the same launch (bez_ppo_policy_rollout_step): same outputs as
        `self(obs, mu, v); rollout_pre(mu, v, ...)`.  prev_post (RolloutPost): rollout_post() of the PREVIOUS env step in the same launch;
        action_noise (ActionNoise): env_actions also receives the env's domain-randomisation action noise (the env must not add it again);
        dr_step (env.dr_step_args()): the coming env step's randomisation runs as one extra workgroup of this launch."""
        hidden, mu_wb, value_wb, rms = self.keep
        n, A = obs.shape[0], self.num_actions
        assert obs.shape[1] == self.d_in and noise.shape == (n, A) and logstd.numel() == A and dones.numel() == n
        assert mb_obs.shape == obs.shape and mb_mu.shape == (n, A) and actions.shape == (n, A) and env_actions.shape == (n, A) and sigma.shape == (n, A)
        assert mb_dones.numel() == n and mb_val.numel() == n and neglogp.numel() == n
        # the rows of mb_obs / (mb_mu, actions, sigma) / neglogp may be strided views (rows of the env-major dataset): unit stride inside a
        # row, one common row stride per group
        layout = None
        so, sa, s1 = mb_obs.stride(0), mb_mu.stride(0), (neglogp.stride(0) if neglogp.dim() else 1)
        assert mb_obs.stride(1) == 1 and mb_mu.stride(1) == 1 and actions.stride() == mb_mu.stride() and sigma.stride() == mb_mu.stride()
        if (so, sa, s1) != (self.d_in, A, 1):
            layout = RolloutLayout(so, sa, s1)
        _chk(lib().bez_ppo_policy_rollout_step(
            _p(obs), n, self.d_in, None if rms is None else _p(rms.running_mean, torch.float64), None if rms is None else _p(rms.running_var, torch.float64),
            0.0 if rms is None else float(rms.epsilon), self.k, C.cast(self.hw, C.c_void_p), C.cast(self.hb, C.c_void_p), C.cast(self.widths, C.c_void_p),
            C.c_void_p(self.mu_w.data_ptr()), _p(mu_wb[1], torch.float16), A, _p(value_wb[0], torch.float16), _p(value_wb[1], torch.float16), _p(logstd), _p(noise),
            _p(dones), None if value_rms is None else _p(value_rms.running_mean, torch.float64),
            None if value_rms is None else _p(value_rms.running_var, torch.float64), 0.0 if value_rms is None else float(value_rms.epsilon), _pv(mb_obs),
            _p(mb_dones), _pv(mb_mu), _p(mb_val), _pv(actions), _p(env_actions), _pv(neglogp), _pv(sigma), self.flag,
            None if prev_post is None else C.byref(prev_post), None if action_noise is None else C.byref(action_noise),
            None if dr_step is None else C.cast(dr_step, C.c_void_p), None if layout is None else C.byref(layout), _stream(obs)),
            "bez_ppo_policy_rollout_step")
