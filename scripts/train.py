#!/usr/bin/env python
"""Training entry point -- the MI355X counterpart of the reference's ``scripts/train.py:66-188``.

    python scripts/train.py --cfg experiments/h36m/train.yaml [--gpus 0] [--workers 8] [--frequent 100]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train.py --cfg ...   (one rank per GPU)

Same command line, same YAML schema, same wiring as the reference: model factory, ``eval('loss.' + LOSS.FN)`` criterion,
``get_optimizer``, ``MultiStepLR(LR_STEP, LR_FACTOR)`` stepped BEFORE each epoch (train.py:158), resume from
``MODEL.RESUME`` (full checkpoint or bare state_dict, train.py:112-122), per-epoch train -> validate -> evaluate ->
``save_checkpoint`` (train.py:157-181) and the final ``final_state.pth.tar`` (train.py:183-187).  Differences:
* ``torch.nn.DataParallel`` (train.py:93-94) is replaced by one process per GPU with a bucketed RCCL gradient all-reduce
  (``epipolarpose_amd.distributed``); ``TRAIN.BATCH_SIZE`` keeps its meaning of images per GPU (train.py:143);
* checkpoints keep the reference's on-disk format: ``checkpoint.pth.tar`` / ``model_best.pth.tar`` carry the ``module.`` key
  prefix DataParallel gives them, ``final_state.pth.tar`` does not; either form is accepted on resume;
* resume keeps the reference's behaviour exactly: the scheduler is NOT restored, the learning rate comes back with the optimizer
  state and ``LR_STEP`` milestones count from the resume point (train.py:105-118 builds the scheduler before loading);
* ``DATASET.TRI`` batches are made view-major (``dataset.view_major_collate``) so that the self-supervision step pairs
  sample i with sample i + B/2 as ``img_utils.py:194-199`` expects.
"""
import argparse
import os
import pprint
import shutil
import sys

import torch
import torch.utils.data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import epipolarpose_amd  # noqa: E402

epipolarpose_amd.install_as_lib()            # the reference's import names (train.py:14-26) now resolve to this package

from lib.core.config import config, get_model_name, update_config  # noqa: E402
from lib.core.function import eval_integral, train_integral, validate_integral  # noqa: E402
from lib.utils.utils import create_logger, get_optimizer, save_checkpoint  # noqa: E402

import lib.core.integral_loss as loss  # noqa: E402
import lib.dataset as dataset  # noqa: E402
import lib.models as models  # noqa: E402


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Train keypoints network')
    parser.add_argument('--cfg', help='experiment configure file name', required=True, type=str)
    args, _ = parser.parse_known_args(argv)
    update_config(args.cfg)
    parser.add_argument('--frequent', help='frequency of logging', default=config.PRINT_FREQ, type=int)
    parser.add_argument('--gpus', help='gpus', type=str)
    parser.add_argument('--workers', help='num of dataloader workers', type=int, default=8)
    return parser.parse_args(argv)


def reset_config(cfg, args):
    if args.gpus:
        cfg.GPUS = args.gpus
    if args.workers is not None:
        cfg.WORKERS = args.workers
    if args.frequent:
        cfg.PRINT_FREQ = args.frequent


def strip_module_prefix(state_dict):
    if len(state_dict) and all(k.startswith('module.') for k in state_dict):
        return type(state_dict)((k[7:], v) for k, v in state_dict.items())
    return state_dict


def main(argv=None):
    from epipolarpose_amd import distributed as epd

    best_perf = 0.0
    args = parse_args(argv)
    reset_config(config, args)
    rank, world, local = epd.init_from_env()
    logger, final_output_dir = create_logger(config, args.cfg, 'train')
    if rank == 0:
        logger.info(pprint.pformat(args))
        logger.info(pprint.pformat(config))

    torch.backends.cudnn.benchmark = config.CUDNN.BENCHMARK              # MIOpen find mode
    torch.backends.cudnn.deterministic = config.CUDNN.DETERMINISTIC
    torch.backends.cudnn.enabled = config.CUDNN.ENABLED
    # GPUS (config / --gpus): the reference hands the id list to nn.DataParallel (train.py:93-94).  Here one PROCESS drives one GPU:
    # a single process honours a single id; several ids need `python -m torch.distributed.run --nproc-per-node N scripts/train.py ...`
    gpu_ids = [int(i) for i in str(getattr(config, 'GPUS', '') or '').replace(' ', '').split(',') if i != '']
    if world == 1 and gpu_ids:
        if len(gpu_ids) > 1:
            logger.warning('GPUS lists %d devices but this is a single process (WORLD_SIZE=1): training on GPU %d only, with a per-GPU batch of '
                           '%d images -- launch with torch.distributed.run --nproc-per-node %d for data parallelism', len(gpu_ids), gpu_ids[0],
                           config.TRAIN.BATCH_SIZE, len(gpu_ids))
        local = gpu_ids[0]
    torch.cuda.set_device(local)
    if config.CUDNN.DETERMINISTIC and torch.cuda.is_available():
        # (after set_device: the library keeps one partial-sum scratch per device, allocated on the device that is current at first use)
        from epipolarpose_amd import hip as _hip
        _hip.set_deterministic(True)          # ordered BatchNorm sums instead of atomics: bit-identical reruns (epipolar_hip.h)

    model = models.pose3d_resnet.get_pose_net(config, is_train=True).cuda()
    if rank == 0 and os.path.abspath(os.path.dirname(args.cfg)) != os.path.abspath(final_output_dir):
        shutil.copy2(args.cfg, final_output_dir)

    loss_fn = getattr(loss, config.LOSS.FN)                               # train.py:97 (`eval('loss.' + ...)`)
    criterion = loss_fn(num_joints=config.MODEL.NUM_JOINTS, norm=config.LOSS.NORM).cuda()
    train, validate, evaluate = train_integral, validate_integral, eval_integral

    epd.broadcast_module(model)                                           # identical start on every rank, BEFORE the optimizer
    optimizer = get_optimizer(config, model)                              # snapshots its bf16 training copies
    lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, config.TRAIN.LR_STEP, config.TRAIN.LR_FACTOR)
    grad_sync = epd.BucketedGradSync(model, optimizer=optimizer) if world > 1 else None
    from epipolarpose_amd.optim import enable_step_in_backward
    enable_step_in_backward(optimizer, model, grad_sync)                  # EPI_STEP_IN_BACKWARD=1 only (measured: < 1 %)

    if config.MODEL.RESUME != '':                                         # train.py:112-122
        checkpoint = torch.load(config.MODEL.RESUME, map_location='cpu')
        if 'epoch' in checkpoint.keys():
            config.TRAIN.BEGIN_EPOCH = checkpoint['epoch']
            best_perf = checkpoint['perf']
            model.load_state_dict(strip_module_prefix(checkpoint['state_dict']))
            optimizer.load_state_dict(checkpoint['optimizer'])            # carries the decayed learning rate
        else:
            model.load_state_dict(strip_module_prefix(checkpoint))
        if hasattr(optimizer, 'refresh_training_copies'):
            optimizer.refresh_training_copies()
        logger.info('=> resume from pretrained model {}'.format(config.MODEL.RESUME))

    ds = getattr(dataset, config.DATASET.DATASET)                         # train.py:125
    train_dataset = ds(cfg=config, root=config.DATASET.ROOT, image_set=config.DATASET.TRAIN_SET, is_train=True)
    valid_dataset = ds(cfg=config, root=config.DATASET.ROOT, image_set=config.DATASET.TEST_SET, is_train=False)

    tri = bool(config.DATASET.TRI)
    items_per_batch = config.TRAIN.BATCH_SIZE // 2 if tri else config.TRAIN.BATCH_SIZE      # a TRI item carries two images
    sampler = None
    if world > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(train_dataset, num_replicas=world, rank=rank,
                                                                  shuffle=config.TRAIN.SHUFFLE)
    # A data set whose items are made ON THE DEVICE (dataset/h36m.py H36M_Integral: the patch comes from the crop kernel) is iterated in this process:
    # a forked DataLoader worker cannot initialise HIP again behind model.cuda() ("Cannot re-initialize CUDA in forked subprocess"), and the work the
    # reference gives its 8 cv2 workers (train.py:142) is one kernel launch here.  Host-only data sets keep config.WORKERS.
    def workers_for(ds_obj):
        return 0 if getattr(ds_obj, "items_use_device", False) else config.WORKERS
    train_loader = torch.utils.data.DataLoader(train_dataset, batch_size=max(1, items_per_batch),
                                               shuffle=config.TRAIN.SHUFFLE and sampler is None, sampler=sampler,
                                               num_workers=workers_for(train_dataset), pin_memory=True, drop_last=world > 1,
                                               collate_fn=dataset.view_major_collate)
    valid_loader = torch.utils.data.DataLoader(valid_dataset, batch_size=config.TEST.BATCH_SIZE, shuffle=False,
                                               num_workers=workers_for(valid_dataset), pin_memory=True)

    best_model = False
    for epoch in range(config.TRAIN.BEGIN_EPOCH, config.TRAIN.END_EPOCH):
        lr_scheduler.step()                                               # train.py:158: before the epoch, as the reference
        if sampler is not None:
            sampler.set_epoch(epoch)
        train(config, train_loader, model, criterion, optimizer, epoch, grad_sync=grad_sync)
        if rank == 0:
            preds_in_patch_with_score = validate(valid_loader, model, num_joints=config.MODEL.NUM_JOINTS)
            acc = evaluate(epoch, preds_in_patch_with_score, valid_loader, final_output_dir, debug=config.DEBUG.DEBUG)
            perf_indicator = 500. - acc                                   # train.py:167 (the expression is always this branch)
            best_model = perf_indicator > best_perf
            if best_model:
                best_perf = perf_indicator
            logger.info('=> saving checkpoint to {}'.format(final_output_dir))
            state = {'module.' + k: v for k, v in model.state_dict().items()}       # DataParallel's key names, as the reference saves
            save_checkpoint({'epoch': epoch + 1, 'model': get_model_name(config), 'state_dict': state, 'perf': perf_indicator,
                             'optimizer': optimizer.state_dict()}, best_model, final_output_dir)
        if world > 1:
            torch.distributed.barrier()

    if rank == 0:
        final_model_state_file = os.path.join(final_output_dir, 'final_state.pth.tar')
        logger.info('saving final model state to {}'.format(final_model_state_file))
        torch.save(model.state_dict(), final_model_state_file)
    if world > 1:
        torch.distributed.destroy_process_group()
    return final_output_dir


if __name__ == '__main__':
    main()
