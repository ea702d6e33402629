"""WalkEnv with the reference's surface (bez_isaacgym/tasks/walk_env.py): the same robot and simulator as bez_kick without the
ball actor; 52 observations (walk_env.py:104,1033-1050); the goal xy is redrawn ~ U(-2,2)^2 by reset_idx and shared by every
env reset in that call (walk_env.py:570-575); reward / reset conditions of walk_env.py:826-1031 run inside the fused kernel
(BezSimConfig.task = BEZ_TASK_WALK)."""
from .kick_env import KickEnv


class WalkEnv(KickEnv):
    TASK = "bez_walk"
    HAS_BALL = False
