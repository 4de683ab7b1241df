"""Training / evaluation loop of the refinement MLP -- mirror of the reference's ``refiner/main.py:31-83`` (``train``, ``test``).

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


def parse_args(argv=None):
    """refiner/main.py:17-29."""
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument('--exp', type=str, default='test', help='ID of experiment')
    parser.add_argument('--load', type=str, default=None, help='path to load a pretrained checkpoint')
    parser.add_argument('--mode', type=str, default='train', help='mode: [train, test]')
    parser.add_argument('--num_epochs', type=int, default=200, help='num epochs')
    parser.add_argument('--lr', type=float, default=1e-3, help='learning rate')
    parser.add_argument('--lr_decay', type=int, default=100000, help='# steps of lr decay')
    parser.add_argument('--lr_gamma', type=float, default=0.96)
    return parser.parse_args(argv)


def main(argv=None):
    """refiner/main.py:84-175: logging, model, criterion, optimizer, optional checkpoint, the epoch loop with evaluation and checkpoints.
    The datasets are ``refiner.data.Human36M`` (the reference's pickles when present, synthetic pairs otherwise)."""
    import logging
    import os
    import time
    from .data import Human36M
    from .model import get_model, weight_init
    from .utils import save_ckpt
    args = parse_args(argv)
    err_best = 1000
    log_dir = os.path.join('refiner/experiments', args.exp)
    os.makedirs(log_dir, exist_ok=True)
    logging.basicConfig(filename=os.path.join(log_dir, '%s_log_%s.log' % (args.mode, time.strftime('%Y-%m-%d-%H-%M'))), format='%(asctime)-15s %(message)s')
    logger = logging.getLogger()
    logger.setLevel(logging.INFO)
    logging.getLogger('').addHandler(logging.StreamHandler())
    model = get_model(weights=None).cuda()
    model.apply(weight_init)
    criterion = TwoHeadMSE()
    optimizer = make_optimizer(model, lr=args.lr)
    glob_step, lr_now = 0, args.lr
    if args.load:
        logger.info(">>> loading ckpt from '{}'".format(args.load))
        ckpt = torch.load(args.load)
        err_best, glob_step, lr_now = ckpt['err'], ckpt['step'], ckpt['lr']
        model.load_state_dict(ckpt['state_dict'])
        optimizer.load_state_dict(ckpt['optimizer'])
        logger.info(">>> ckpt loaded (epoch: {} | err: {})".format(ckpt['epoch'], err_best))
    train_set = Human36M(is_train=True)
    train_dl = torch.utils.data.DataLoader(dataset=train_set, batch_size=64, shuffle=True, num_workers=0)
    test_dl = torch.utils.data.DataLoader(dataset=Human36M(is_train=False, norm=train_set.norm()), batch_size=64, shuffle=False, num_workers=0)
    logger.info("- done.")
    if args.mode == 'train':
        logger.info("Starting training for {} epoch(s)".format(args.num_epochs))
        for epoch in range(args.num_epochs):
            logger.info('%s | %s | lr: %.6f' % (epoch, args.num_epochs, lr_now))
            glob_step, lr_now = train(model, train_dl, optimizer, glob_step, lr_now, criterion, args, logger)
            logger.info('Evaluation')
            error = test(model, test_dl)
            error = error[0] if isinstance(error, tuple) else error
            is_best = error < err_best
            err_best = min(error, err_best)
            save_ckpt({'epoch': epoch + 1, 'lr': lr_now, 'step': glob_step, 'err': error, 'state_dict': model.state_dict(),
                       'optimizer': optimizer.state_dict()}, ckpt_path=log_dir, is_best=is_best)
            if is_best:
                logger.info('Found new best, error: %s' % error)
    elif args.mode == 'test':
        return test(model, test_dl)
    else:
        print('mode input error!')
    return err_best


if __name__ == '__main__':
    main()
