"""Experiment configuration -- mirror of the reference's ``lib/core/config.py`` (schema and behaviour).

Every key of the reference's global ``config`` (config.py:10-139) exists with the same default, YAML overlays raise
``ValueError`` on unknown keys (config.py:163-167,183-184) so ``experiments/*.yaml`` files written for the reference
load unchanged, and ``get_model_name`` yields the same strings.  No ``easydict`` dependency: ``AttrDict`` below.
"""
import os

import numpy as np
import yaml


class AttrDict(dict):
    """dict with attribute access; nested plain dicts are converted on assignment."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, AttrDict):
            value = AttrDict(value)
        super().__setitem__(key, value)

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    __setattr__ = __setitem__


_POSE_RESNET = {
    "NUM_LAYERS": 50, "DECONV_WITH_BIAS": False, "NUM_DECONV_LAYERS": 3, "NUM_DECONV_FILTERS": [256, 256, 256],
    "NUM_DECONV_KERNELS": [4, 4, 4], "FINAL_CONV_KERNEL": 1, "TARGET_TYPE": "gaussian", "HEATMAP_SIZE": [64, 64],
    "SIGMA": 2,
}
MODEL_EXTRAS = {"pose3d_resnet": _POSE_RESNET}


def default_config():
    """A fresh copy of the reference defaults (config.py:10-139)."""
    return AttrDict({
        "OUTPUT_DIR": "", "LOG_DIR": "", "DATA_DIR": "", "GPUS": "0", "WORKERS": 8, "PRINT_FREQ": 20,
        "EXP_NAME": "default",
        "CUDNN": {"BENCHMARK": True, "DETERMINISTIC": False, "ENABLED": True},
        "MODEL": {"NAME": "pose3d_resnet", "INIT_WEIGHTS": True, "PRETRAINED": "", "RESUME": "", "NUM_JOINTS": 17,
                  "IMAGE_SIZE": [256, 256], "DEPTH_RES": 64, "VOLUME": True,
                  "EXTRA": {k: (list(v) if isinstance(v, list) else v) for k, v in _POSE_RESNET.items()}},
        "LOSS": {"USE_TARGET_WEIGHT": True, "FN": "L1JointLocationLoss", "USE_SOFT": True, "NORM": False,
                 "DEPTH_LAMBDA": 1.0},
        "DATASET": {"ROOT": "", "DATASET": "mpii", "TRAIN_SET": "train", "TEST_SET": "valid", "DATA_FORMAT": "jpg",
                    "HYBRID_JOINTS_TYPE": "", "SELECT_DATA": False, "TRI": False, "MPII_ORDER": False,
                    "TRAIN_FRAME": 32, "VAL_FRAME": 64, "NUM_CAMS": 4, "DEPTH_RANGE": 2000,
                    "FLIP": True, "SCALE_FACTOR": 0.25, "ROT_FACTOR": 30, "OCCLUSION": False,
                    "VOC": "/media/muhammed/Other/RESEARCH/datasets/VOCdevkit/VOC2012", "BG_AUG": False,
                    "Z_WEIGHT": 1.0},
        "TRAIN": {"LR_FACTOR": 0.1, "LR_STEP": [90, 110], "LR": 0.001, "OPTIMIZER": "adam", "MOMENTUM": 0.9,
                  "WD": 0.0001, "NESTEROV": False, "GAMMA1": 0.99, "GAMMA2": 0.0, "BEGIN_EPOCH": 0, "END_EPOCH": 140,
                  "RESUME": False, "CHECKPOINT": "", "BATCH_SIZE": 32, "SHUFFLE": True},
        "TEST": {"BATCH_SIZE": 32, "FLIP_TEST": False, "POST_PROCESS": True, "SHIFT_HEATMAP": True,
                 "USE_GT_BBOX": False, "OKS_THRE": 0.5, "IN_VIS_THRE": 0.0, "COCO_BBOX_FILE": "", "BBOX_THRE": 1.0,
                 "MODEL_FILE": "", "IMAGE_THRE": 0.0, "NMS_THRE": 1.0},
        "DEBUG": {"DEBUG": False, "SAVE_BATCH_IMAGES_GT": False, "SAVE_BATCH_IMAGES_PRED": False,
                  "SAVE_HEATMAPS_GT": False, "SAVE_HEATMAPS_PRED": False, "SAVE_3D": False},
    })


config = default_config()       # the module-level mutable config, as in the reference (config.py:8)


def _as_pair(v):
    return np.array([v, v]) if isinstance(v, int) else np.array(v)


def _update_dict(cfg, k, v):
    """config.py:142-167: normalise a few entries, then overlay, rejecting unknown keys."""
    if k == 'DATASET':
        for key in ('MEAN', 'STD'):
            if key in v and v[key]:
                v[key] = np.array([eval(x) if isinstance(x, str) else x for x in v[key]])
    if k == 'MODEL':
        if 'EXTRA' in v and 'HEATMAP_SIZE' in v['EXTRA']:
            v['EXTRA']['HEATMAP_SIZE'] = _as_pair(v['EXTRA']['HEATMAP_SIZE'])
        if 'IMAGE_SIZE' in v:
            v['IMAGE_SIZE'] = _as_pair(v['IMAGE_SIZE'])
    for vk, vv in v.items():
        if vk not in cfg[k]:
            raise ValueError("{}.{} not exist in config.py".format(k, vk))
        if isinstance(vv, dict) and isinstance(cfg[k][vk], dict):      # MODEL.EXTRA: overlay, keep defaults
            for ek, ev in vv.items():
                cfg[k][vk][ek] = ev
        else:
            cfg[k][vk] = vv


def update_config(config_file, cfg=None):
    """config.py:170-184.  Overlays the YAML onto ``cfg`` (default: the global ``config``)."""
    cfg = config if cfg is None else cfg
    with open(config_file) as f:
        exp_config = yaml.safe_load(f) or {}
    for k, v in exp_config.items():
        if k not in cfg:
            raise ValueError("{} not exist in config.py".format(k))
        if isinstance(v, dict):
            _update_dict(cfg, k, v)
        elif k == 'SCALES':
            cfg[k][0] = tuple(v)
        else:
            cfg[k] = v
    return cfg


def gen_config(config_file, cfg=None):
    """config.py:187-194."""
    cfg = config if cfg is None else cfg

    def plain(d):
        return {k: (plain(v) if isinstance(v, dict) else (v.tolist() if isinstance(v, np.ndarray) else v))
                for k, v in d.items()}
    with open(config_file, 'w') as f:
        yaml.dump(plain(cfg), f, default_flow_style=False)


def update_dir(model_dir, log_dir, data_dir, cfg=None):
    """config.py:197-214."""
    cfg = config if cfg is None else cfg
    if model_dir:
        cfg.OUTPUT_DIR = model_dir
    if log_dir:
        cfg.LOG_DIR = log_dir
    if data_dir:
        cfg.DATA_DIR = data_dir
    cfg.DATASET.ROOT = os.path.join(cfg.DATA_DIR, cfg.DATASET.ROOT)
    cfg.TEST.COCO_BBOX_FILE = os.path.join(cfg.DATA_DIR, cfg.TEST.COCO_BBOX_FILE)
    cfg.MODEL.PRETRAINED = os.path.join(cfg.DATA_DIR, cfg.MODEL.PRETRAINED)


def get_model_name(cfg):
    """config.py:217-250 -> (name, full_name)."""
    extra = cfg.MODEL.EXTRA
    name = '{}_{}'.format(cfg.MODEL.NAME, extra.NUM_LAYERS)
    height, width = cfg.MODEL.IMAGE_SIZE[1], cfg.MODEL.IMAGE_SIZE[0]
    if cfg.MODEL.NAME == 'pose_resnet':
        suffix = ''.join('d{}'.format(n) for n in extra.NUM_DECONV_FILTERS)
    elif cfg.MODEL.NAME == 'pose3d_resnet':
        suffix = 'DR%s_S%s_DL%s' % (cfg.MODEL.DEPTH_RES, int(cfg.LOSS.USE_SOFT), int(cfg.LOSS.DEPTH_LAMBDA))
    else:
        raise ValueError('Unkown model: {}'.format(cfg.MODEL))
    return name, '{}x{}_{}_{}'.format(height, width, name, suffix)
