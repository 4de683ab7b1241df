"""GPU: the drop-in boundary exercised end to end -- ``install_as_lib()``, ``train_integral`` in fully-supervised and TRI
(self-supervised) mode over ``SyntheticH36M`` through a ``DataLoader`` (pseudo labels checked against the oracle), the
``scripts/train.py`` counterpart (MultiStepLR, checkpoint cadence, resume), and optimizer-state interoperability with
``torch.optim.Adam`` (reference checkpoints)."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(tri, joints=5, depth=16, image=64, layers=18):
    from epipolarpose_amd.core.config import default_config
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = joints, depth, [image, image]
    cfg.MODEL.EXTRA.NUM_LAYERS = layers
    cfg.DATASET.TRI = tri
    cfg.DATASET.DATASET = "h36m"
    cfg.LOSS.FN = "SmoothL1JointLocationLoss"
    cfg.PRINT_FREQ = 1
    return cfg


class _Recorder(torch.nn.Module):
    """Wraps the criterion and keeps what the loop handed to it."""

    def __init__(self, inner):
        super().__init__()
        self.inner, self.calls = inner, []
        self.num_joints = inner.num_joints

    def forward(self, preds, label, weight):
        self.calls.append((preds.detach().float().contiguous().cpu().numpy(), label.detach().cpu().numpy(), weight.detach().cpu().numpy()))
        return self.inner(preds, label, weight)


@pytest.mark.parametrize("collate", ["stock", "view_major"])
def test_train_integral_fs_and_tri_through_dataloader(collate):
    import epipolarpose_amd
    epipolarpose_amd.install_as_lib()
    import lib.core.integral_loss as loss
    import lib.dataset as dataset
    import lib.models as models
    from lib.core.function import train_integral
    from lib.utils.utils import get_optimizer
    from oracle import geometry as o_geo
    torch.manual_seed(0)
    np.random.seed(0)
    j = 5
    # ---- fully supervised: labels come from the dataset ----
    cfg = _cfg(False)
    model = models.pose3d_resnet.get_pose_net(cfg, is_train=True).cuda()
    criterion = _Recorder(getattr(loss, cfg.LOSS.FN)(num_joints=j, norm=cfg.LOSS.NORM).cuda())
    optimizer = get_optimizer(cfg, model)
    ds = dataset.h36m(cfg=cfg, root="", image_set="train-fs", is_train=True, n_group=2, n_view=4)
    loader = list(torch.utils.data.DataLoader(ds, batch_size=4, shuffle=True))[:2]
    avg = train_integral(cfg, loader, model, criterion, optimizer, 0)
    assert np.isfinite(avg) and len(criterion.calls) == 2
    for (_, label, weight), batch in zip(criterion.calls, loader):
        np.testing.assert_array_equal(label, batch[1].numpy())
        np.testing.assert_array_equal(weight, batch[2].numpy())
    # ---- TRI: pseudo labels from the two-view triangulation inside the step ----
    cfg = _cfg(True)
    ds = dataset.h36m(cfg=cfg, root="", image_set="train-ss", is_train=True, n_group=4, n_view=4)
    kw = {} if collate == "stock" else {"collate_fn": dataset.view_major_collate}
    loader = list(torch.utils.data.DataLoader(ds, batch_size=2, shuffle=True, **kw))[:2]
    criterion.calls.clear()
    avg = train_integral(cfg, loader, model, criterion, optimizer, 1)
    assert np.isfinite(avg) and len(criterion.calls) == 2
    for (preds, label, weight), batch in zip(criterion.calls, loader):
        if isinstance(batch, dict):
            batch = dataset.tri_batch_to_view_major(batch)
        assert preds.shape[0] == 4 and label.shape == (4, 3 * j)
        meta = {k: v.numpy() for k, v in batch[3].items() if isinstance(v, torch.Tensor)}
        olab, owt, _, _ = o_geo.self_supervision(preds, meta, n_view=2, num_joints=j)       # the reference's iterative-LS pairing
        np.testing.assert_allclose(label, olab, atol=5e-6)
        np.testing.assert_array_equal(weight, owt)
        assert not np.allclose(label, batch[1].numpy(), atol=1e-3)                         # NOT the dataset's ground truth


def _write_yaml(path, tri, resume=""):
    import yaml
    doc = {"OUTPUT_DIR": str(path.parent / "out"), "WORKERS": 0, "PRINT_FREQ": 1, "EXP_NAME": "t",
           "DATASET": {"DATASET": "h36m", "ROOT": "", "TRAIN_SET": "train", "TEST_SET": "valid", "TRI": tri, "NUM_CAMS": 4},
           "MODEL": {"NAME": "pose3d_resnet", "INIT_WEIGHTS": False, "RESUME": resume, "NUM_JOINTS": 5, "IMAGE_SIZE": [64, 64],
                     "DEPTH_RES": 16, "EXTRA": {"NUM_LAYERS": 18}},
           "LOSS": {"FN": "SmoothL1JointLocationLoss", "NORM": False},
           "TRAIN": {"BATCH_SIZE": 8, "SHUFFLE": True, "BEGIN_EPOCH": 0, "END_EPOCH": 2, "OPTIMIZER": "adam", "LR": 0.001,
                     "LR_FACTOR": 0.1, "LR_STEP": [2, 3]},
           "TEST": {"BATCH_SIZE": 8}}
    with open(path, "w") as f:
        yaml.safe_dump(doc, f)


def test_train_script_epochs_checkpoint_resume(tmp_path, monkeypatch):
    """scripts/train.py:105-188: MultiStepLR before each epoch, checkpoint every epoch (reference key names), resume."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import epipolarpose_amd
    from epipolarpose_amd.core import config as C
    epipolarpose_amd.install_as_lib()
    import importlib
    train = importlib.import_module("train")

    def reset():
        fresh = C.default_config()
        C.config.clear()
        C.config.update(fresh)
    lrs = []
    import lib.core.function as F
    real = F.train_integral

    def spy(config, loader, model, criterion, optimizer, epoch, **kw):
        lrs.append((epoch, optimizer.param_groups[0]["lr"]))
        return real(config, loader, model, criterion, optimizer, epoch, **kw)
    monkeypatch.setattr(train, "train_integral", spy)
    reset()
    y1 = tmp_path / "fs.yaml"
    _write_yaml(y1, tri=False)
    out = train.main(["--cfg", str(y1), "--workers", "0"])
    ck = torch.load(os.path.join(out, "checkpoint.pth.tar"), map_location="cpu")
    assert ck["epoch"] == 2 and all(k.startswith("module.") for k in ck["state_dict"]) and "optimizer" in ck and "perf" in ck
    assert os.path.exists(os.path.join(out, "model_best.pth.tar")) and os.path.exists(os.path.join(out, "final_state.pth.tar"))
    final = torch.load(os.path.join(out, "final_state.pth.tar"), map_location="cpu")
    assert "conv1.weight" in final and set(ck["optimizer"]["state"][0]) >= {"step", "exp_avg", "exp_avg_sq"}
    # the scheduler steps BEFORE each epoch (train.py:158): milestone 2 is reached at the start of epoch index 1
    assert [e for e, _ in lrs] == [0, 1] and lrs[0][1] == pytest.approx(1e-3) and lrs[1][1] == pytest.approx(1e-4)
    # resume in TRI mode from the full checkpoint: BEGIN_EPOCH 2; the learning rate returns with the optimizer state (1e-4) and
    # the fresh scheduler counts its milestones from the resume point, exactly as the reference does
    reset()
    lrs.clear()
    y2 = tmp_path / "ss.yaml"
    _write_yaml(y2, tri=True, resume=os.path.join(out, "checkpoint.pth.tar"))
    import yaml
    doc = yaml.safe_load(open(y2))
    doc["TRAIN"]["END_EPOCH"] = 4
    yaml.safe_dump(doc, open(y2, "w"))
    out2 = train.main(["--cfg", str(y2), "--workers", "0"])
    assert [e for e, _ in lrs] == [2, 3]
    assert lrs[0][1] == pytest.approx(1e-4) and lrs[1][1] == pytest.approx(1e-5)
    ck2 = torch.load(os.path.join(out2, "checkpoint.pth.tar"), map_location="cpu")
    assert ck2["epoch"] == 4 and int(float(ck2["optimizer"]["state"][0]["step"])) > int(float(ck["optimizer"]["state"][0]["step"]))
    # bare state_dict resume branch (train.py:120-122)
    reset()
    lrs.clear()
    y3 = tmp_path / "bare.yaml"
    _write_yaml(y3, tri=False, resume=os.path.join(out, "final_state.pth.tar"))
    doc = yaml.safe_load(open(y3))
    doc["TRAIN"]["END_EPOCH"] = 1
    yaml.safe_dump(doc, open(y3, "w"))
    train.main(["--cfg", str(y3), "--workers", "0"])
    assert [e for e, _ in lrs] == [0]
    reset()


def test_fused_adam_state_interoperates_with_torch_adam():
    """A reference checkpoint holds torch.optim.Adam state: loading it must continue the moments (round-1 defect: silently
    dropped), and FusedAdam's own state_dict must load into torch.optim.Adam."""
    import torch.nn as nn
    from epipolarpose_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    base = nn.Sequential(nn.Conv2d(8, 16, 3, padding=1, bias=False), nn.ReLU(), nn.Conv2d(16, 8, 1, bias=True), nn.Flatten(),
                         nn.Linear(8 * 6 * 5, 7)).to(dev).to(memory_format=torch.channels_last)
    x, y = torch.randn(4, 8, 6, 5, device=dev), torch.randn(4, 7, device=dev)

    def steps(m, opt, n):
        for _ in range(n):
            opt.zero_grad(set_to_none=True)
            ((m(x).float() - y) ** 2).mean().backward()
            opt.step()
    ref_m = copy.deepcopy(base)
    ref_opt = torch.optim.Adam(ref_m.parameters(), lr=1e-2)
    steps(ref_m, ref_opt, 3)
    mid_model, mid_opt = copy.deepcopy(ref_m.state_dict()), copy.deepcopy(ref_opt.state_dict())
    steps(ref_m, ref_opt, 2)
    # torch Adam state -> FusedAdam (fp32 path so that the comparison is tight)
    m = copy.deepcopy(base)
    m.load_state_dict(mid_model)
    opt = FusedAdam(m, lr=1e-2, low_precision_convs=False)
    opt.load_state_dict(mid_opt)
    assert opt._step == 3
    steps(m, opt, 2)
    for a, b in zip(m.parameters(), ref_m.parameters()):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6)
    # FusedAdam state -> torch Adam
    m2 = copy.deepcopy(base)
    m2.load_state_dict(mid_model)
    opt2 = FusedAdam(m2, lr=1e-2, low_precision_convs=False)
    opt2.load_state_dict(mid_opt)
    sd = opt2.state_dict()
    m3 = copy.deepcopy(base)
    m3.load_state_dict(mid_model)
    opt3 = torch.optim.Adam(m3.parameters(), lr=1e-2)
    opt3.load_state_dict(sd)
    steps(m3, opt3, 2)
    for a, b in zip(m3.parameters(), ref_m.parameters()):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6)
    with pytest.raises(ValueError):
        bad = copy.deepcopy(mid_opt)
        del bad["state"][0]
        opt2.load_state_dict(bad)


def test_training_copy_follows_load_state_dict():
    """after FusedAdam(model) the forward reads bf16 copies; model.load_state_dict alone must not leave them
    stale (eval of model_best, the bare-state_dict resume branch)."""
    import torch.nn as nn
    from epipolarpose_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    m = nn.Sequential(nn.Conv2d(8, 8, 3, padding=1, bias=False)).to(dev).to(memory_format=torch.channels_last)
    opt = FusedAdam(m, lr=1e-2)
    x = torch.randn(2, 8, 5, 5, device=dev)
    new = {k: torch.randn_like(v) for k, v in m.state_dict().items()}
    m.load_state_dict(new)
    out = m(x)
    want = torch.nn.functional.conv2d(x.to(torch.bfloat16), new["0.weight"].to(torch.bfloat16), padding=1)
    torch.testing.assert_close(out.float(), want.float(), rtol=2e-2, atol=2e-2)
    assert torch.equal(opt.training_copies()[0].detach().float(), new["0.weight"].to(torch.bfloat16).float())
