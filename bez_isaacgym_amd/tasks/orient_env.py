"""OrientEnv with the reference's surface (bez_isaacgym/tasks/orient_env.py): turn in place to env.goalState.goal_angle.
Same robot and simulator as bez_walk; the observation carries (cos, sin) of the heading error (compute_off_angle,
orient_env.py:719-735) where the other tasks carry off_orn; reward / reset conditions of orient_env.py:843-1018 run inside
the fused kernel (BezSimConfig.task = BEZ_TASK_ORIENT)."""
import torch

from .kick_env import KickEnv


class OrientEnv(KickEnv):
    TASK = "bez_orient"
    HAS_BALL = False

    def __init__(self, cfg, sim_device, graphics_device_id, headless):
        super().__init__(cfg, sim_device, graphics_device_id, headless)
        goal_angle = self.cfg["env"]["goalState"]["goal_angle"]  # orient_env.py:61,145
        self.goal_angle = torch.tensor([goal_angle], device=self.device, dtype=torch.float32).repeat((self.num_envs, 1))
