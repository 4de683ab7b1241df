"""Shared helper for the GPU tests: a config with the network geometry under test."""


def make_cfg(layers, image, joints, depth):
    from epipolarpose_amd.core.config import default_config
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = joints, depth, [image, image]
    cfg.MODEL.EXTRA.NUM_LAYERS = layers
    return cfg
