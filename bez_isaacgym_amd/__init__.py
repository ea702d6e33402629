"""bez_isaacgym_amd -- MI355X-native `bez_kick` environment step behind the reference's VecTask surface.

Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI), abi.py / sim.py (ctypes binding),
tasks/ (VecTask / KickEnv mirror), utils/ (config loader, rl_games-style adapter), ppo/ (the consumer loop).
"""
__version__ = "0.1.0"
