"""bez_isaacgym_amd -- MI355X-native `bez_kick` environment step behind the reference's VecTask surface.

Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI), abi.py / sim.py (ctypes binding),
tasks/ (VecTask / KickEnv mirror), utils/ (config loader, rl_games-style adapter), ppo/ (the consumer loop).
"""
__version__ = "0.1.0"

import os as _os
import sys as _sys

# ---- HIP-graph replay safety (profiles/r04_plain_graph_probe.txt).  ROCm 7.2's HIP runtime pre-builds the AQL packets of
# an instantiated graph ("graph packet capture") and keeps their kernel arguments in memory that ordinary launches recycle: once
# enough kernel-argument bytes have been launched EAGERLY between two replays (~8 KB for a graph of thousands of nodes, a few
# hundred KB for this package's fused PPO graphs), replayed kernels run with clobbered arguments -- silently wrong numbers, spurious
# non-finite gradients, NaN weights.  The runtime's own switch turns the feature off; it must be in the environment before the first HIP
# call of the process (it still works after `import torch`).  Costs 2 % of the PPO leg (5.12 -> 5.22 ms per epoch).  A value
# already present in the environment is respected (DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 re-enables the feature at the user's risk).
_PACKET_CAPTURE = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_torch = _sys.modules.get("torch")
_hip_up = bool(_torch is not None and _torch.cuda.is_initialized())
GRAPH_REPLAY_SAFE = _os.environ.get(_PACKET_CAPTURE) == "0" or not _hip_up and _os.environ.get(_PACKET_CAPTURE) is None
if _os.environ.get(_PACKET_CAPTURE) is None:
    _os.environ[_PACKET_CAPTURE] = "0"
del _torch, _hip_up
