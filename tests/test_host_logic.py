"""CPU: host-side mirror of the reference interface -- config schema, model state-dict surface, label codec,
group sharding, and (when /root/reference is present) live comparison with the reference modules."""
import ast
import os

import numpy as np
import pytest
import torch

import ref_shims
from make_golden_cases import NETWORK_CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_cfg(layers, image, joints, depth):
    from epipolarpose_amd.core.config import default_config
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = joints, depth, [image, image]
    cfg.MODEL.EXTRA.NUM_LAYERS = layers
    return cfg


@pytest.mark.parametrize("case", NETWORK_CASES, ids=[c[0] for c in NETWORK_CASES])
def test_state_dict_surface_matches_reference(golden, case):
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    g = golden("network")
    name, layers, image, j, d, b = case
    sd = get_pose_net(make_cfg(layers, image, j, d), is_train=True).state_dict()
    assert list(sd.keys()) == g[name + "/keys"].tolist()
    assert [tuple(v.shape) for v in sd.values()] == [ast.literal_eval(s) for s in g[name + "/shapes"].tolist()]


def test_init_weights_requires_file():
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    cfg = make_cfg(18, 64, 3, 8)
    cfg.MODEL.INIT_WEIGHTS = True
    cfg.MODEL.PRETRAINED = "/nonexistent/imagenet.pth"
    with pytest.raises(ValueError):
        get_pose_net(cfg, is_train=True)
    get_pose_net(cfg, is_train=False)            # evaluation never touches the file (pose3d_resnet.py:302)


def test_pretrained_loading_strips_module_prefix(tmp_path):
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    cfg = make_cfg(18, 64, 3, 8)
    src = get_pose_net(cfg, is_train=False)
    sd = {"module." + k: v for k, v in src.state_dict().items()}
    sd["module.final_layer.weight"] = torch.zeros(7, 256, 1, 1)        # shape mismatch -> dropped
    path = tmp_path / "mpii_pose.pth"
    torch.save(sd, path)
    cfg.MODEL.INIT_WEIGHTS, cfg.MODEL.PRETRAINED = True, str(path)
    dst = get_pose_net(cfg, is_train=True)
    assert torch.equal(dst.state_dict()["layer3.1.conv2.weight"], src.state_dict()["layer3.1.conv2.weight"])
    assert dst.final_layer.weight.shape == (24, 256, 1, 1)


def test_config_schema_and_yaml_overlay(tmp_path):
    from epipolarpose_amd.core import config as C
    cfg = C.default_config()
    assert cfg.MODEL.EXTRA.NUM_DECONV_FILTERS == [256, 256, 256] and cfg.TRAIN.LR_STEP == [90, 110]
    assert C.get_model_name(cfg) == ("pose3d_resnet_50", "256x256_pose3d_resnet_50_DR64_S1_DL1")
    y = tmp_path / "exp.yaml"
    y.write_text("GPUS: '0,1'\nMODEL:\n  NUM_JOINTS: 16\n  EXTRA:\n    NUM_LAYERS: 18\nLOSS:\n  FN: SmoothL1JointLocationLoss\n"
                 "TRAIN:\n  LR_STEP:\n  - 90\n  - 120\n")
    C.update_config(str(y), cfg)
    assert cfg.GPUS == "0,1" and cfg.MODEL.NUM_JOINTS == 16 and cfg.MODEL.EXTRA.NUM_LAYERS == 18
    assert cfg.MODEL.EXTRA.FINAL_CONV_KERNEL == 1 and cfg.TRAIN.LR_STEP == [90, 120]
    bad = tmp_path / "bad.yaml"
    bad.write_text("MODEL:\n  NOT_A_KEY: 1\n")
    with pytest.raises(ValueError, match="MODEL.NOT_A_KEY not exist"):
        C.update_config(str(bad), cfg)
    bad.write_text("NOPE: 1\n")
    with pytest.raises(ValueError):
        C.update_config(str(bad), cfg)


@pytest.mark.skipif(not ref_shims.reference_available(), reason="/root/reference not present (GPU box)")
def test_config_defaults_equal_reference_and_yaml_files_load():
    from epipolarpose_amd.core import config as C
    ref = ref_shims.load_reference().config.config

    def flat(d, pre=""):
        out = {}
        for k, v in d.items():
            if isinstance(v, dict):
                out.update(flat(v, pre + k + "."))
            else:
                out[pre + k] = v.tolist() if isinstance(v, np.ndarray) else v
        return out
    assert flat(C.default_config()) == flat(ref)
    for rel in ("h36m/train.yaml", "h36m/train-ss.yaml", "h36m/valid.yaml", "h36m/valid-ss.yaml", "mpii/train.yaml",
                "mpii/valid.yaml"):
        cfg = C.update_config(os.path.join(ref_shims.REFERENCE_ROOT, "experiments", rel), C.default_config())
        assert cfg.MODEL.NAME == "pose3d_resnet"


def test_label_codec_mutates_like_reference(golden):
    from epipolarpose_amd.core.integral_loss import generate_joint_location_label, reverse_joint_location_label
    g = golden("integral")
    joints = g["label/joints"].copy()
    lab, vis = generate_joint_location_label(256.0, 256.0, joints, np.ones((17, 3)))
    np.testing.assert_allclose(lab, g["label/label"], atol=1e-15)
    assert np.shares_memory(lab, joints)                      # in place, as integral_loss.py:171-173
    np.testing.assert_allclose(reverse_joint_location_label(256.0, 256.0, lab.copy()), g["label/reverse"], atol=1e-12)


def test_group_sharding_never_splits_views():
    from epipolarpose_amd.distributed import shard_groups
    for world in (1, 2, 4, 8):
        spans = [shard_groups(64, r, world) for r in range(world)]
        assert sum(n for _, n in spans) == 64 and spans[0][0] == 0
        assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    assert [shard_groups(10, r, 4)[1] for r in range(4)] == [3, 3, 2, 2]


def test_synthetic_item_contract():
    from epipolarpose_amd.synthetic import SyntheticScenes
    sc = SyntheticScenes(n_group=3, n_view=4, num_joints=17, seed=0)
    assert sc.label.shape == (12, 51) and sc.label.dtype == np.float32 and sc.weight.dtype == np.float32
    assert set(sc.meta) == {"center_x", "center_y", "width", "height", "scale", "rot", "R", "T", "f", "c", "projection_matrix"}
    assert sc.meta["projection_matrix"].shape == (12, 3, 4) and sc.meta["T"].shape == (12, 3, 1)
    assert np.abs(sc.label).max() < 1.5


def test_bucket_pack_variants_write_the_same_bytes():
    """BucketedGradSync packs a completed bucket either with torch._foreach_copy_ (default) or with one torch.cat into the flat
    buffer (EPI_BUCKET_PACK=cat): identical bucket contents and parameter-strided .grad views, channels_last weights included."""
    import torch
    import torch.nn as nn
    from epipolarpose_amd import distributed as epd

    def run(mode):
        old = epd.BUCKET_PACK
        epd.BUCKET_PACK = mode
        try:
            torch.manual_seed(0)
            m = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 1), nn.ReLU(), nn.Conv2d(8, 16, 3, padding=1),
                              nn.Flatten(), nn.Linear(16 * 36, 5)).to(memory_format=torch.channels_last)
            s = epd.BucketedGradSync(m, bucket_bytes=4096)
            x = torch.randn(4, 3, 6, 6)
            for _ in range(3):
                s.zero_grad()
                m(x).square().sum().backward()
                s.finish()
            return [f.clone() for f, _, _ in s.buckets], [p.grad.clone() for p in m.parameters()], [p.grad.stride() for p in m.parameters()]
        finally:
            epd.BUCKET_PACK = old
    a, b = run("foreach"), run("cat")
    assert len(a[0]) > 1
    assert all(torch.equal(x, y) for x, y in zip(a[0], b[0])) and all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
    assert a[2] == b[2]
    t = torch.arange(24.0).reshape(2, 3, 2, 2).contiguous(memory_format=torch.channels_last)
    assert torch.equal(epd._memory_order_flat(t), t.permute(0, 2, 3, 1).reshape(-1))
    assert epd._memory_order_flat(t[:, :2]) is None


def test_step_in_backward_split_partitions_the_parameters():
    """PoseResNet.step_in_backward_split (optim.FusedAdam.enable_step_in_backward): the late modules are exactly what runs before the
    boundary's output in the forward pass -- stem + layer1 -- and hold ~1 % of the parameters; the helper is a no-op without the opt-in."""
    from epipolarpose_amd.core.config import default_config
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    from epipolarpose_amd import optim
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    model = get_pose_net(cfg, is_train=False)
    boundary, late = model.step_in_backward_split()
    assert boundary is model.layer1 and late == [model.conv1, model.bn1, model.layer1]
    n_late = sum(p.numel() for m in late for p in m.parameters())
    n_all = sum(p.numel() for p in model.parameters())
    assert 0 < n_late < 0.02 * n_all
    assert optim.enable_step_in_backward(object(), model) is False             # EPI_STEP_IN_BACKWARD unset: nothing is switched on


def test_trace_step_sequence_tool(tmp_path):
    """tools/trace_step_sequence.py on a synthetic two-queue rocprofv3 kernel trace: window between two marker launches, per-queue gaps,
    summed kernel time against the time with at least one kernel running."""
    import subprocess
    import sys
    rows = ["Kind,Agent_Id,Queue_Id,Kernel_Name,Start_Timestamp,End_Timestamp,LDS_Block_Size,Workgroup_Size_X,Workgroup_Size_Y,Workgroup_Size_Z,"
            "Grid_Size_X,Grid_Size_Y,Grid_Size_Z"]

    def k(q, name, s, e, grid=256 * 4):
        rows.append("KERNEL_DISPATCH,0,%d,%s,%d,%d,0,256,1,1,%d,1,1" % (q, name, s, e, grid))
    t = 0
    for step in range(3):
        k(1, "marker_kernel(int)", t, t + 1000)
        k(1, "void epi::a_kernel<1>(float*)", t + 1000, t + 5000)
        k(2, "void epi::b_kernel(float*)", t + 3000, t + 9000)           # overlaps a_kernel by 2 us, runs alone for 4 us
        k(1, "void epi::c_kernel(float*)", t + 10000, t + 12000)         # 5 us after a_kernel on queue 1
        t += 20000
    path = tmp_path / "trace.csv"
    path.write_text("\n".join(rows) + "\n")
    out = tmp_path / "seq.txt"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "trace_step_sequence.py"), str(path), "marker_kernel", str(out), "-1"])
    text = out.read_text().splitlines()
    assert "4 launches" in text[0] and "0.013 ms summed kernel time" in text[0] and "0.011 ms with at least one kernel running" in text[0]
    body = [l.split() for l in text[2:]]
    assert [l[-1] for l in body] == ["a_kernel<1>", "b_kernel", "c_kernel", "marker_kernel"]
    assert float(body[2][2]) == 5.0 and int(body[0][4]) == 4             # c_kernel's gap on its own queue; workgroups = grid / block


def test_h36m_frame_cache_is_bounded_lru(monkeypatch):
    """the per-path frame cache of dataset/h36m.py evicts least recently used frames under a byte budget, and a data set whose
    items are made on the device tells scripts/train.py to iterate it in-process (no forked workers behind model.cuda())."""
    import importlib
    h36m = importlib.import_module("epipolarpose_amd.dataset.h36m")     # (the package also exports a factory FUNCTION of that name)
    decoded = []

    def fake_decode(path):
        decoded.append(path)
        return np.full((10, 10, 3), len(decoded), np.uint8)             # 300 bytes per frame
    monkeypatch.setattr(h36m, "decode_bgr", fake_decode)
    cache = h36m.FrameCache(max_bytes=1000)                             # room for three frames
    cpu = torch.device("cpu")
    for p in ("a", "b", "c"):
        cache.get(p, cpu)
    assert len(cache) == 3 and cache.bytes == 900
    cache.get("a", cpu)                                                 # a is the most recent now
    cache.get("d", cpu)                                                 # evicts b
    assert len(cache) == 3 and decoded == ["a", "b", "c", "d"]
    cache.get("a", cpu)
    cache.get("c", cpu)
    assert decoded == ["a", "b", "c", "d"]                              # hits
    cache.get("b", cpu)                                                 # decoded again, evicts d
    assert decoded[-1] == "b" and len(cache) == 3 and cache.bytes == 900
    frame, hw = cache.get("b", cpu)
    assert hw == (10, 10) and frame.numel() == 300
    assert h36m.H36M_Integral.items_use_device is True
    src = open(os.path.join(ROOT, "scripts", "train.py")).read()
    assert "items_use_device" in src and "num_workers=workers_for(train_dataset)" in src and "num_workers=workers_for(valid_dataset)" in src


def test_h36m_frame_store_refuses_more_than_its_budget():
    import importlib
    h36m = importlib.import_module("epipolarpose_amd.dataset.h36m")
    per_cam = [[{"image": "x.jpg"}] * 10 for _ in range(4)]
    with pytest.raises(ValueError, match="exceed"):
        h36m.H36MFrames(per_cam, "/nowhere", device=torch.device("cpu"), max_bytes=50 * 1000 * 1000)
