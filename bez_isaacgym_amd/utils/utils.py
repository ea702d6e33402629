"""set_seed with the reference's behaviour (bez_isaacgym/utils/utils.py:45-70)."""
import os
import random

import numpy as np
import torch


def set_np_formatting():
    np.set_printoptions(edgeitems=30, infstr="inf", linewidth=4000, nanstr="nan", precision=2, suppress=False,
                        threshold=10000, formatter=None)


def set_seed(seed, torch_deterministic=False):
    if seed == -1 and torch_deterministic:
        seed = 42
    elif seed == -1:
        seed = np.random.randint(0, 10000)
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if torch_deterministic:
        torch.use_deterministic_algorithms(True)
    return seed


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 on numpy uint64 arrays holding 32-bit words -- the same generator the kernels and the oracle use for every
    per-env draw (csrc/bez_kernels.h: philox4x32_10).  Returns the four output words."""
    import numpy as np
    M = np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (np.asarray(v, np.uint64) & M for v in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0) & M, np.uint64(k1) & M
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M, p1 & M, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M, p0 & M
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M, (k1 + np.uint64(0xBB67AE85)) & M
    return c0, c1, c2, c3


def per_env_uniform(seed, global_env_ids, tag, width):
    """(len(ids), width) float32 uniforms in [0, 1), element (e, k) = word k & 3 of Philox(counter = (env id lo, env id hi, tag, k >> 2),
    key = seed): a function of the GLOBAL env id alone, so a draw does not depend on how the envs are sharded over GPUs -- the keying of
    the reset noise (kick_env.py:786-791 restated in the kernels) applied to one-time per-env parameters."""
    import numpy as np
    g = np.asarray(global_env_ids, np.int64).astype(np.uint64)
    out = np.empty((g.size, width), np.float32)
    for blk in range((width + 3) // 4):
        w = philox4x32_10(g, g >> np.uint64(32), np.full_like(g, tag), np.full_like(g, blk), int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
        for j in range(4):
            if 4 * blk + j < width:
                out[:, 4 * blk + j] = (w[j] >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return out
