"""Task registry, same mapping as the reference (bez_isaacgym/tasks/__init__.py:10-16)."""
from .kick_env import KickEnv
from .orient_env import OrientEnv
from .walk_env import WalkEnv

isaacgym_task_map = {
    "bez_kick": KickEnv,
    "bez_walk": WalkEnv,
    "bez_orient": OrientEnv,
}
