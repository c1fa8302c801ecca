"""Seeding (tencentpretrain/utils/seed.py:6-12 of the reference)."""
import os
import random

import numpy as np
import torch


def set_seed(seed=7):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
