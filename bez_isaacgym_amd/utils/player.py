"""Deterministic continuous-action player: what the reference's play path does with a trained checkpoint
(bez_isaacgym/utils/players.py:46-66, play.py:125-236): obs -> RunningMeanStd (eval) -> MLP -> mu -> clamp(+-1) ->
rescale to the action space (Box(-1,1): identity).  Used by play mode and by the sim-to-sim check of the reference
policy; loads either this build's own checkpoints, a foreign rl_games .pth (read without unpickling), or the numeric
fixture tests/golden/bez_kick_33_policy.npz."""
import numpy as np
import torch

from ..ppo.a2c_continuous import ModelA2CContinuousLogStd, RunningMeanStd
from .rlg_checkpoint import load_into_agent_modules, read_rlgames_checkpoint


def checkpoint_from_npz(path):
    """The fixture layout ("model/<key>", "running_mean_std/<key>", scalars) -> the nested dict of read_rlgames_checkpoint."""
    z = np.load(path)
    ck = {"model": {}, "running_mean_std": {}, "reward_mean_std": {}}
    for k in z.files:
        if "/" in k:
            grp, name = k.split("/", 1)
            ck[grp][name] = z[k]
        else:
            ck[k] = z[k].item()
    return ck


class PpoPlayerContinuous:
    def __init__(self, checkpoint, device="cuda:0", obs_dim=54, act_dim=18, units=(400, 200, 100), deterministic=True):
        ck = checkpoint
        if isinstance(checkpoint, str):
            ck = checkpoint_from_npz(checkpoint) if checkpoint.endswith(".npz") else read_rlgames_checkpoint(checkpoint)
        self.device = torch.device(device)
        self.model = ModelA2CContinuousLogStd(obs_dim, act_dim, units).eval()
        self.running_mean_std = RunningMeanStd((obs_dim,)).eval()
        load_into_agent_modules(ck, self.model, self.running_mean_std)
        self.model.to(self.device); self.running_mean_std.to(self.device)
        self.is_deterministic = deterministic
        self.checkpoint = ck

    @torch.no_grad()
    def get_action(self, obs):
        x = self.running_mean_std(obs)
        mu, logstd, _ = self.model.a2c_network(x)
        act = mu if self.is_deterministic else mu + torch.exp(logstd) * torch.randn_like(mu)
        return torch.clamp(act, -1.0, 1.0)  # players.py:63-64; rescale_actions on Box(-1, 1) is the identity
