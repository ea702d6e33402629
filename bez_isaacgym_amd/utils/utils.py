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
