"""qdiff/utils.py of the reference: forward-hook capture and seeding."""
import os
import random

import numpy as np
import torch


class AttentionMap:
    """Keeps the last (input, output) of a module (utils.py:12-24)."""

    def __init__(self, module):
        self.hook = module.register_forward_hook(self.hook_fn)

    def hook_fn(self, module, input, output):
        self.out = output
        self.feature = input

    def remove(self):
        self.hook.remove()


def at(x):
    return x.view(x.size(0), -1)


def at_loss(x, y):
    return (at(x) - at(y)).pow(2).mean(1).sum()


def seed_everything(seed):
    from qdiff.quant_layer import seed_mask_rng
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    seed_mask_rng(seed)
