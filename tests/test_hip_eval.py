"""GPU parity: evaluation path -- epi_evaluate_poses vs the golden output of the reference's own
H36M_Integral.evaluate, and the validate/eval loop on a synthetic dataset."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,mpii", [("h36m", False), ("mpii", True)])
def test_evaluate_vs_reference_golden(golden, tag, mpii):
    from epipolarpose_amd.dataset import h36m_eval
    from oracle import evaluation
    assert torch.cuda.is_available()
    g = golden("evaluation")
    gt = g[tag + "/gt_joints"]
    if mpii:
        gt = gt[:, h36m_eval.H36M_TO_MPII_PERM, :]
    name_value, perf, per_sample, per_joint = h36m_eval.evaluate_arrays(g[tag + "/preds"], gt, g[tag + "/pelvis"], g[tag + "/fl"],
                                                                       g[tag + "/c_p"], mpii_order=mpii)
    assert [k for k, _ in name_value] == g[tag + "/names"].tolist()
    np.testing.assert_allclose([v for _, v in name_value], g[tag + "/metrics"], rtol=1e-9, atol=1e-8)     # mm, float64
    np.testing.assert_allclose(perf, g[tag + "/perf"], rtol=1e-10)
    _, o_ps, o_pj = evaluation.evaluate(g[tag + "/preds"], gt, g[tag + "/pelvis"], g[tag + "/fl"], g[tag + "/c_p"], mpii_order=mpii)
    np.testing.assert_allclose(per_sample, o_ps, rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(per_joint, o_pj, rtol=1e-9, atol=1e-8)


def test_evaluate_degenerate_and_exact():
    """Exact predictions -> all errors 0; a similarity-transformed pose -> PA-MPJPE 0; planar pose (rank-2 covariance) is finite."""
    from epipolarpose_amd.dataset import h36m_eval
    rng = np.random.default_rng(0)
    n, j = 6, 17
    cam = rng.normal(0, 300, size=(n, j, 3)) + np.array([0, 0, 5000.0])
    fl, cp = np.tile([1145.0, 1145.0], (n, 1)), np.tile([512.0, 512.0], (n, 1))
    def to_img(x):
        uv = x[:, :, :2] / x[:, :, 2:3] * fl[:, None] + cp[:, None]
        return np.concatenate([uv, x[:, :, 2:3] - cam[:, 0:1, 2:3]], axis=2)
    gt = to_img(cam)
    nv, perf, ps, pj = h36m_eval.evaluate_arrays(gt.copy(), gt, cam[:, 0], fl, cp)
    assert np.abs(ps).max() < 1e-9
    q = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    q *= np.sign(np.linalg.det(q))
    moved = 1.2 * ((cam - cam[:, 0:1]) @ q) + cam[:, 0:1]
    nv, perf, ps, pj = h36m_eval.evaluate_arrays(to_img(moved), gt, cam[:, 0], fl, cp)
    assert ps[:, 0].min() > 50 and np.abs(ps[:, 1]).max() < 1e-6          # large MPJPE, zero after Procrustes
    flat = cam.copy()
    flat[:, :, 2] = 5000.0
    nv, perf, ps, pj = h36m_eval.evaluate_arrays(to_img(flat + rng.normal(0, 5, size=flat.shape) * [1, 1, 0]), to_img(flat), flat[:, 0], fl, cp)
    assert np.isfinite(ps).all()


def test_validate_and_eval_loop_on_synthetic_dataset():
    from torch.utils.data import DataLoader
    from epipolarpose_amd.core.config import default_config
    from epipolarpose_amd.core.function import eval_integral, validate_integral
    from epipolarpose_amd.dataset.synthetic import SyntheticH36M
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = 17, 16, [64, 64]
    cfg.MODEL.EXTRA.NUM_LAYERS = 18
    ds = SyntheticH36M(cfg, n_group=3, n_view=2, seed=5)
    loader = DataLoader(ds, batch_size=4, shuffle=False, num_workers=0)           # ragged last batch (6 = 4 + 2)
    torch.manual_seed(0)
    model = get_pose_net(cfg, is_train=False).cuda()
    preds = validate_integral(loader, model, num_joints=17)
    assert preds.shape == (6, 17, 4) and preds.dtype == np.float64 and np.all(preds[:, :, 3] == 1)
    perf = eval_integral(0, preds, loader, None)
    assert np.isfinite(perf) and perf > 0
    # feeding the ground truth back through the same path gives zero error (patch<->image round trip + evaluation)
    gt_patch = ds.scenes.patch_coords()
    assert eval_integral(0, gt_patch, loader, None) < 1e-2
