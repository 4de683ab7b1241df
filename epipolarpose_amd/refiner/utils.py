"""refiner/utils.py of the reference: exponential learning-rate decay, meters, checkpoints."""
import os

import torch

from ..utils.utils import AverageMeter  # noqa: F401  (refiner/utils.py:4-15: the same meter)


def lr_decay(optimizer, step, lr, decay_step, gamma):
    """refiner/utils.py:18-22: lr * gamma ** (step / decay_step), written into every parameter group."""
    lr = lr * gamma ** (step / decay_step)
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr
    return lr


step_decay = lr_decay            # refiner/utils.py:24-28 is the same expression


def save_ckpt(state, ckpt_path, is_best=True):
    """refiner/utils.py:30-35."""
    torch.save(state, os.path.join(ckpt_path, 'best.pth.tar' if is_best else 'last.pth.tar'))
