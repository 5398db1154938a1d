"""utils/fixseed.py:6-10 — seed python, numpy and torch. The on-device Philox seed of a sampling call is
drawn from torch's default generator, so fixing the seed here fixes the samples."""
import random

import numpy as np
import torch


def fixseed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
