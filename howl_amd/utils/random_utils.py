"""``set_random_seed`` (``howl/utils/random_utils.py:7-17``)."""
import random

import numpy as np
import torch


def set_random_seed(seed: int = 0):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
