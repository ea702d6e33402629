"""Task registry, same mapping name as the reference (bez_isaacgym/tasks/__init__.py:10-16).
Only `bez_kick` is in scope of this build; walk/orient are listed as 'next'."""
from .kick_env import KickEnv

isaacgym_task_map = {
    "bez_kick": KickEnv,
}
