"""Training utilities -- mirror of the reference's ``lib/utils/utils.py`` (the parts the hot path uses)."""
import logging
import os
import time
from pathlib import Path

import torch
import torch.optim as optim

from ..core.config import get_model_name


def create_logger(cfg, cfg_name, phase='train'):
    """utils.py:13-42: output/<dataset>/<model>/<EXP_NAME>/<cfg>_<time>_<phase>.log + console."""
    root = Path(cfg.OUTPUT_DIR)
    root.mkdir(parents=True, exist_ok=True)
    dataset = cfg.DATASET.DATASET + ('_' + cfg.DATASET.HYBRID_JOINTS_TYPE if cfg.DATASET.HYBRID_JOINTS_TYPE else '')
    model, _ = get_model_name(cfg)
    out_dir = root / dataset.replace(':', '_') / model / cfg.EXP_NAME
    out_dir.mkdir(parents=True, exist_ok=True)
    stem = os.path.basename(cfg_name).split('.')[0]
    log_file = out_dir / '{}_{}_{}.log'.format(stem, time.strftime('%Y-%m-%d-%H-%M'), phase)
    logging.basicConfig(filename=str(log_file), format='%(asctime)-15s %(message)s')
    logger = logging.getLogger()
    logger.setLevel(logging.INFO)
    if not any(isinstance(h, logging.StreamHandler) and not isinstance(h, logging.FileHandler) for h in logger.handlers):
        logger.addHandler(logging.StreamHandler())
    return logger, str(out_dir)


def get_optimizer(cfg, model, capturable=False, fused=True):
    """utils.py:45-61.  Adam takes only the learning rate (the reference ignores TRAIN.WD for Adam).
    On the GPU the default is ``epipolarpose_amd.optim.FusedAdam`` (same arithmetic, one launch, bf16 training copies
    of the MIOpen convolution weights); ``fused=False`` or ``capturable=True`` (hipGraph capture keeps Adam's step
    counter on the device) selects ``torch.optim.Adam``."""
    if cfg.TRAIN.OPTIMIZER == 'sgd':
        return optim.SGD(model.parameters(), lr=cfg.TRAIN.LR, momentum=cfg.TRAIN.MOMENTUM, weight_decay=cfg.TRAIN.WD,
                         nesterov=cfg.TRAIN.NESTEROV)
    if cfg.TRAIN.OPTIMIZER == 'adam':
        params = [p for p in model.parameters() if p.requires_grad]
        if fused and not capturable and params and all(p.is_cuda for p in params):
            from ..optim import FusedAdam
            return FusedAdam(model, lr=cfg.TRAIN.LR)
        return optim.Adam(model.parameters(), lr=cfg.TRAIN.LR, capturable=capturable)
    return None


def save_checkpoint(states, is_best, output_dir, filename='checkpoint.pth.tar'):
    """utils.py:64-69: full dict every call; the best model keeps only its state_dict."""
    torch.save(states, os.path.join(output_dir, filename))
    if is_best and 'state_dict' in states:
        torch.save(states['state_dict'], os.path.join(output_dir, 'model_best.pth.tar'))


class AverageMeter(object):
    """utils.py:199-214."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count if self.count != 0 else 0
