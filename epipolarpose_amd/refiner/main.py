"""Training / evaluation loop of the refinement MLP -- mirror of the reference's ``refiner/main.py:31-83`` (``train``, ``test``).
The reference script's command line, logging set-up and epoch / checkpoint loop (``refiner/main.py:17-29,84-175``) are control plane outside the
hot-path scope and are NOT mirrored here: drive ``train`` / ``test`` from your own script.

Same step: two-headed MSE (``criterion(p1, t) + criterion(p2, t)``), ``clip_grad_norm_(max_norm=1)``, Adam, exponential lr decay
every ``lr_decay`` steps.  Here the clip is folded into the fused Adam launch (``optim.FusedAdam(max_grad_norm=1.0)``) and the loss
is read back once per epoch instead of once per step.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import hip
from ..optim import FusedAdam
from .utils import AverageMeter, lr_decay


class TwoHeadMSE(nn.Module):
    """``nn.MSELoss(reduction='mean')`` applied to both heads and summed (refiner/main.py:49), value and gradient from
    ``epi_joint_loss`` (kind L2 with weights 1 / n: sum / B of w * d^2 = the mean)."""

    def forward(self, outputs, targets):
        n = targets.shape[1]
        w = torch.full_like(targets, 1.0 / n)
        return sum(_MSE.apply(o, targets, w) for o in outputs)


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight):
        loss, grad = hip.joint_loss(pred.float().contiguous(), target, weight, "l2", norm=False, size_average=True, need_grad=True)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def make_optimizer(model, lr=1e-3):
    """refiner/main.py:107 (Adam) + :53 (clip_grad_norm_ 1.0) as one fused launch pair."""
    return FusedAdam(model, lr=lr, low_precision_convs=False, max_grad_norm=1.0)


def train(model, train_dl, optimizer, glob_step, lr_now, criterion, args, logger=None):
    """refiner/main.py:31-60.  ``args`` needs ``lr``, ``lr_decay``, ``lr_gamma``."""
    losses = AverageMeter()
    model.train()
    pending, count = None, 0
    for inp, tar in train_dl:
        glob_step += 1
        if glob_step % args.lr_decay == 0 or glob_step == 1:
            lr_now = lr_decay(optimizer, glob_step, args.lr, args.lr_decay, args.lr_gamma)
        inputs, targets = inp.cuda(non_blocking=True), tar.cuda(non_blocking=True)
        outputs = model(inputs)
        optimizer.zero_grad()
        loss = criterion(outputs, targets)
        loss.backward()
        optimizer.step()
        pending = loss.detach() * targets.size(0) if pending is None else pending + loss.detach() * targets.size(0)
        count += targets.size(0)
    if pending is not None:
        losses.update(float(pending.item()) / count, count)
    if logger is not None:
        logger.info('Avg Loss: %.5f' % losses.avg)
    train.last_avg_loss = losses.avg          # (the reference only logs it; its return value is the 2-tuple below, refiner/main.py:60)
    return glob_step, lr_now


train.last_avg_loss = None


def test(model, test_dl):
    """refiner/main.py:62-83: second-head predictions of the whole set -> ``dataset.evaluate``."""
    model.eval()
    preds = []
    with torch.no_grad():
        for inp, _ in test_dl:
            preds.append(model(inp.cuda(non_blocking=True))[-1])
    preds = torch.cat(preds, dim=0).cpu().numpy()
    return test_dl.dataset.evaluate(np.asarray(preds))
