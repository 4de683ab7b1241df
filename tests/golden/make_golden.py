#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by EXECUTING THE REFERENCE (``/root/reference``).

Run in the build container only (the reference is not present on the GPU box):

    python tests/golden/make_golden.py

Everything stored here is an output of the reference's own code on seeded inputs (see ref_shims.py for
the third-party stubs).  Inputs are either stored next to the outputs or regenerated from the seeded
helpers in det_weights.py / epipolarpose_amd.synthetic (an input checksum is stored in that case).
"""
import copy
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_shims                                   # noqa: E402
from det_weights import fill_state_dict, seeded_array   # noqa: E402
from make_golden_cases import (BIG_HEAD_STD, dlogits_stride, INTEGRAL_CASES, LOGIT_STRIDE, NETWORK_BIG_CASES, NETWORK_CASES,   # noqa: E402
                               TRAJECTORY_CASES, TRAJECTORY_HEAD_STD, golden_grad_keys, grad_stride)
from epipolarpose_amd.synthetic import SyntheticScenes  # noqa: E402

REF = ref_shims.load_reference()
torch.set_num_threads(8)


def save(name, **arrays):
    # EPI_GOLDEN_OUT: write somewhere else (tests/test_golden_regeneration.py regenerates the fast sets into a temporary directory and compares)
    path = os.path.join(os.environ.get("EPI_GOLDEN_OUT", HERE), name)
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB)" % (name, os.path.getsize(path) / 1024.0))


# --------------------------------------------------------------------------------------------------
# 1. integral regression: soft-argmax, criteria, autograd gradients, decode
# --------------------------------------------------------------------------------------------------

def gen_integral():
    il = REF.integral_loss
    out = {}
    for name, b, j, d, h, w, sc in INTEGRAL_CASES:
        logits = seeded_array("logits/" + name, (b, j * d, h, w), scale=sc)
        gt = seeded_array("gt/" + name, (b, 3 * j), scale=0.25)
        wt = (np.abs(seeded_array("wt/" + name, (b, 3 * j))) > 0.15).astype(np.float32)   # ~12 % zeros
        big = logits.size > 20000     # big inputs are regenerated from the seed; only a checksum is stored
        if big:
            out[name + "/logits_crc"] = np.uint32(zlib.crc32(logits.tobytes()))
        else:
            out[name + "/logits"] = logits
        out[name + "/gt"] = gt
        out[name + "/wt"] = wt
        t = torch.from_numpy(logits)
        out[name + "/xyz"] = il.softmax_integral_tensor(t, j, True, w, h, d).numpy()
        # the SAME reference code on float64 tensors: what its arithmetic converges to.  At the configuration's shape (262 144 voxels per joint) the
        # float32 run above is itself 1.8e-5 away from it in the coordinates (fp32 sums of the marginals) -- more than the 1e-5 the kernels are held to.
        out[name + "/xyz64"] = il.softmax_integral_tensor(t.double(), j, True, w, h, d).numpy()
        assert out[name + "/xyz64"].dtype == np.float64
        for kind, fn, cls in (("l1", il.weighted_l1_loss, il.L1JointLocationLoss),
                              ("smoothl1", il.weighted_smooth_l1_loss, il.SmoothL1JointLocationLoss),
                              ("l2", il.weighted_mse_loss, None)):
            for norm in (False, True):
                for wide in (False, True):
                    if wide and not big:
                        continue           # (small cases: the float32 run is within 2e-6 of the float64 oracle, tests/test_oracle_golden.py)
                    cast = (lambda a: torch.from_numpy(a).double()) if wide else torch.from_numpy
                    tl = cast(logits).clone().requires_grad_(True)
                    if cls is not None:
                        loss = cls(num_joints=j, norm=norm)(tl, cast(gt), cast(wt))
                    else:   # L2JointLocationLoss.forward is broken in the reference (integral_loss.py:110-112);
                            # its pieces are not: compose them the way the class intends.
                        pj = il.softmax_integral_tensor(tl, j, True, w, h, d)
                        loss = fn(pj, cast(gt), cast(wt), True, norm)
                    loss.backward()
                    key = "%s/%s/norm%d" % (name, kind, int(norm))
                    g = tl.grad.numpy()
                    if wide:
                        out[key + "/loss64"] = np.float64(loss.item())
                        out[key + "/dlogits64"] = g.reshape(-1)[::dlogits_stride(g.size)]
                    else:
                        out[key + "/loss"] = np.float32(loss.item())
                        out[key + "/dlogits"] = g.reshape(-1)[::dlogits_stride(g.size)] if big else g
        if d == w:   # get_joint_location_result infers D = W (integral_loss.py:191)
            out[name + "/decode256"] = il.get_joint_location_result(256, 256, torch.from_numpy(logits))
    # label codec
    joints = np.random.default_rng(5).uniform(-100, 300, size=(17, 3))
    lab, vis = il.generate_joint_location_label(256., 256., joints.copy(), np.ones((17, 3)))
    out["label/joints"] = joints
    out["label/label"] = lab
    out["label/reverse"] = il.reverse_joint_location_label(256., 256., lab.copy())
    save("integral.npz", **out)


# --------------------------------------------------------------------------------------------------
# 2. triangulation (reference control flow; cv2 primitives stubbed)
# --------------------------------------------------------------------------------------------------
def ref_camera(cam):
    return REF.cameras.Camera((cam["R"], cam["T"], cam["f"], cam["c"], None, None, "synthetic"))


def gen_triangulation():
    tri = REF.triangulation
    out = {}
    sc = SyntheticScenes(n_group=3, n_view=4, num_joints=17, seed=11, noise_px=0.0)
    ps = np.stack([ref_camera(c).projection_matrix for c in sc.cams])          # live cameras.py:126-131
    out["P"] = ps
    out["P_ours"] = np.stack([c["projection_matrix"] for c in sc.cams])
    out["world"] = sc.world
    rng = np.random.default_rng(3)
    for noise in (0.0, 2.0):
        u = np.stack([np.stack([REF_project(sc.world[g], sc.cams[v]) for g in range(3)]) for v in range(4)])
        u = u + rng.normal(0, noise, size=u.shape) if noise > 0 else u         # [V,G,J,2]
        out["u/noise%d" % int(noise)] = u
        for (va, vb) in ((0, 1), (0, 3), (1, 2)):
            for g in range(3):
                tag = "noise%d/v%d%d/g%d" % (int(noise), va, vb, g)
                x, st = tri.iterative_LS_triangulation(u[va, g], ps[va], u[vb, g], ps[vb])
                out[tag + "/iter_x"], out[tag + "/iter_status"] = x, st
                x, st = tri.linear_LS_triangulation(u[va, g], ps[va], u[vb, g], ps[vb])
                out[tag + "/ls_x"] = x
                x, st = tri.linear_eigen_triangulation(u[va, g], ps[va], u[vb, g], ps[vb])
                out[tag + "/eigen_x"], out[tag + "/eigen_status"] = x, st
    # points behind one / both cameras -> status codes -1, -2, -3 (triangulation.py:176-179)
    cam0, cam1 = sc.cams[0], sc.cams[1]
    pts = np.array([[0.0, 0.0, 900.0],
                    cam0["T"].reshape(3) * 1.3,                     # behind camera 0
                    cam1["T"].reshape(3) * 1.3,                     # behind camera 1
                    (cam0["T"].reshape(3) + cam1["T"].reshape(3)) * 2.0 + [0, 0, 5000.0]])
    u0 = REF_project(pts, cam0)
    u1 = REF_project(pts, cam1)
    x, st = tri.iterative_LS_triangulation(u0, ps[0], u1, ps[1])
    out["behind/u0"], out["behind/u1"], out["behind/x"], out["behind/status"] = u0, u1, x, st
    # polynomial (optimal) triangulation, triangulation.py:184-220: the reference's F construction + hand-over run live
    u_n = out["u/noise2"]
    out["poly/u1"], out["poly/u2"] = u_n[0].reshape(-1, 2), u_n[3].reshape(-1, 2)
    out["poly/P1"], out["poly/P2"] = ps[0], ps[3]
    x, st = tri.polynomial_triangulation(out["poly/u1"], ps[0], out["poly/u2"], ps[3])
    out["poly/X"], out["poly/status"] = x, st
    p1f, p2f = np.eye(4), np.eye(4)
    p1f[:3], p2f[:3] = ps[0], ps[3]
    pc = p2f.dot(np.linalg.inv(p1f))
    out["poly/F"] = np.cross(pc[0:3, 3], pc[0:3, 0:3], axisb=0).T                 # the expression of triangulation.py:204
    save("triangulation.npz", **out)


def REF_project(x_world, cam):
    """Pinhole projection through the reference's CamProj (prep_h36m.py:170-175)."""
    xc = (x_world - cam["T"].reshape(3)) @ cam["R"].T
    u, v = REF.prep_h36m.CamProj(xc[:, 0], xc[:, 1], xc[:, 2], cam["f"][0], cam["f"][1], cam["c"][0], cam["c"][1])
    return np.stack([u, v], axis=1)


# --------------------------------------------------------------------------------------------------
# 3. crop affine, re-projection, full self_supervision
# --------------------------------------------------------------------------------------------------
def torch_meta(meta):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in meta.items()}


def gen_geometry():
    iu = REF.img_utils
    out = {}
    rng = np.random.default_rng(21)
    params = []
    for _ in range(12):
        params.append([rng.uniform(300, 700), rng.uniform(300, 700), rng.uniform(250, 600), rng.uniform(250, 600),
                       np.clip(rng.normal(), -1, 1) * 0.25 + 1.0,
                       np.clip(rng.normal(), -2, 2) * 30.0 if rng.random() < 0.7 else 0.0])
    params = np.asarray(params)
    out["affine/params"] = params
    out["affine/fwd"] = np.stack([iu.gen_trans_from_patch_cv(p[0], p[1], p[2], p[3], 256, 256, p[4], p[5], inv=False)
                                  for p in params])
    out["affine/inv"] = np.stack([iu.gen_trans_from_patch_cv(p[0], p[1], p[2], p[3], 256, 256, p[4], p[5], inv=True)
                                  for p in params])
    coords = np.concatenate([rng.uniform(0, 256, size=(12, 17, 2)), rng.uniform(-128, 128, size=(12, 17, 1)),
                             np.ones((12, 17, 1))], axis=2)
    out["decode/coords_patch"] = coords
    out["decode/coords_img"] = np.stack([
        iu.trans_coords_from_patch_to_org_3d(coords[i], p[0], p[1], p[2], p[3], 256, 256, 2000, 2000,
                                             scale=p[4], rot=p[5]) for i, p in enumerate(params)])

    for n_group, j, tag in ((3, 17, "h36m"), (2, 16, "mpii")):
        sc = SyntheticScenes(n_group=n_group, n_view=2, num_joints=j, seed=31 + j, noise_px=0.0)
        meta = torch_meta(sc.meta)
        # world -> image joints (prep_h36m.py:177-204)
        r = REF.prep_h36m.from_worldjt_to_imagejt(j, sc.meta["R"][0], sc.world[0], sc.meta["T"][0], sc.meta["f"][0],
                                                  sc.meta["c"][0], 2000., 2000.)
        out[tag + "/w2i/pt2d"], out[tag + "/w2i/pt3d"] = r[4], r[5]
        out[tag + "/w2i/rect"] = np.array(r[0:4])
        # labels from (exact) global coordinates == the generator's own labels
        x_world = np.concatenate([sc.world] * 2, axis=0)
        lab, wt = iu.get_batch_labels_from_global_coords(x_world, meta)
        out[tag + "/labels_from_world/label"], out[tag + "/labels_from_world/weight"] = lab, wt
        out[tag + "/scene_label"] = sc.label
        # triangulate() on noisy decoded coordinates
        cp = sc.patch_coords(noise_px=1.5, seed=7)
        kps_img = np.stack([iu.trans_coords_from_patch_to_org_3d(
            cp[n], sc.meta["center_x"][n], sc.meta["center_y"][n], sc.meta["width"][n], sc.meta["height"][n],
            256, 256, 2000, 2000, scale=sc.meta["scale"][n], rot=sc.meta["rot"][n]) for n in range(sc.batch_size)])
        out[tag + "/ss/coords_patch"] = cp
        out[tag + "/ss/kps_img"] = kps_img
        xw = iu.triangulate(kps_img, meta)
        out[tag + "/ss/x_world"] = xw
        lab, wt = iu.get_batch_labels_from_global_coords(xw, meta)
        out[tag + "/ss/label"], out[tag + "/ss/weight"] = lab, wt

    # full self_supervision(preds, meta) from logits: D = H = W = 16 (D must equal W, integral_loss.py:191)
    sc = SyntheticScenes(n_group=2, n_view=2, num_joints=5, seed=77)
    logits = sc.peaked_logits(depth=16, hm=16, gain=12.0, sigma=1.5)
    logits = logits + seeded_array("ss/noise", logits.shape, scale=0.3)
    out["ss_full/logits"] = logits.astype(np.float32)
    lab, wt = iu.self_supervision(torch.from_numpy(logits.astype(np.float32)), torch_meta(sc.meta))
    out["ss_full/label"], out["ss_full/weight"] = lab, wt
    save("geometry.npz", **out)


# --------------------------------------------------------------------------------------------------
# 4. arg-max decode
# --------------------------------------------------------------------------------------------------
def gen_maxpreds():
    hm = seeded_array("maxpreds", (3, 6, 8, 12))
    hm[0, 0] = -np.abs(hm[0, 0])                 # all negative  -> masked to 0
    hm[0, 1, 2, 3] = hm[0, 1, 5, 7] = 9.0        # tie -> first index wins
    hm[1, 2] = 0.0                               # all zero -> idx 0, masked
    preds, maxvals = REF.inference.get_max_preds(hm)
    save("maxpreds.npz", heatmaps=hm, preds=preds, maxvals=maxvals)


# --------------------------------------------------------------------------------------------------
# 5. the network (reference PoseResNet, deterministic weights)
# --------------------------------------------------------------------------------------------------

def ref_cfg(layers, image, joints, depth):
    cfg = copy.deepcopy(REF.config.config)
    cfg.MODEL.NUM_JOINTS = joints
    cfg.MODEL.DEPTH_RES = depth
    cfg.MODEL.IMAGE_SIZE = [image, image]
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.EXTRA.NUM_LAYERS = layers
    return cfg


def gen_network():
    out = {}
    for name, layers, image, j, d, b in NETWORK_CASES:
        model = REF.pose3d_resnet.get_pose_net(ref_cfg(layers, image, j, d), is_train=True)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        out[name + "/keys"] = np.array(list(shapes.keys()))
        out[name + "/shapes"] = np.array([str(s) for s in shapes.values()])
        model.load_state_dict(fill_state_dict(shapes, seed=1))
        x = torch.from_numpy(seeded_array("img/" + name, (b, 3, image, image)))
        gt = torch.from_numpy(seeded_array("gt/" + name, (b, 3 * j), scale=0.2))
        wt = torch.ones(b, 3 * j)
        model.eval()
        with torch.no_grad():
            out[name + "/logits_eval"] = model(x).numpy()
        model.train()
        logits = model(x)
        out[name + "/logits_train"] = logits.detach().numpy()
        loss = REF.integral_loss.SmoothL1JointLocationLoss(num_joints=j)(logits, gt, wt)
        loss.backward()
        out[name + "/loss"] = np.float32(loss.item())
        sd = model.state_dict()
        out[name + "/bn1.running_mean"] = sd["bn1.running_mean"].numpy()
        out[name + "/deconv_layers.7.running_var"] = sd["deconv_layers.7.running_var"].numpy()
        grads = {k: p.grad for k, p in model.named_parameters()}
        legacy = ("final_layer.weight", "final_layer.bias", "deconv_layers.6.weight", "deconv_layers.7.weight", "deconv_layers.0.weight", "conv1.weight")
        for k in golden_grad_keys(grads.keys()):
            g = grads[k].numpy()
            if k in legacy:       # (rounds 1-3 layout: the whole tensor when small)
                out[name + "/grad/" + k] = g if g.size <= 70000 else g.reshape(-1)[:: max(1, g.size // 50000)]
            else:
                out[name + "/grad/" + k] = g.reshape(-1)[:: grad_stride(k, g.size)].copy()
        out[name + "/gradnorm"] = np.array([float(g.norm()) for g in grads.values()])
    save("network.npz", **out)


def gen_network_big():
    """The reference network at BASELINE.json's full configurations (1, 2, 5), fp32 on the CPU.  Outputs are sub-sampled
    (make_golden_cases.LOGIT_STRIDE) so that the fixture stays small; the decode (soft-argmax) and the loss are stored in full."""
    out = {}
    for name, layers, image, j, d, b in NETWORK_BIG_CASES:
        model = REF.pose3d_resnet.get_pose_net(ref_cfg(layers, image, j, d), is_train=True)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        out[name + "/keys"] = np.array(list(shapes.keys()))
        out[name + "/shapes"] = np.array([str(s) for s in shapes.values()])
        model.load_state_dict(fill_state_dict(shapes, seed=1, head_std=BIG_HEAD_STD))
        x = torch.from_numpy(seeded_array("img/" + name, (b, 3, image, image)))
        gt = torch.from_numpy(seeded_array("gt/" + name, (b, 3 * j), scale=0.2))
        wt = torch.ones(b, 3 * j)
        hm = image // 4
        model.eval()
        with torch.no_grad():
            le = model(x)
            out[name + "/logits_eval"] = le.reshape(-1)[::LOGIT_STRIDE].numpy().copy()
            out[name + "/logits_eval_absmax"] = np.float32(le.abs().max().item())
            out[name + "/xyz_eval"] = REF.integral_loss.softmax_integral_tensor(le, j, True, hm, hm, d).numpy()
        model.train()
        logits = model(x)
        out[name + "/logits_train"] = logits.detach().reshape(-1)[::LOGIT_STRIDE].numpy().copy()
        out[name + "/logits_train_absmax"] = np.float32(logits.detach().abs().max().item())
        out[name + "/xyz_train"] = REF.integral_loss.softmax_integral_tensor(logits.detach(), j, True, hm, hm, d).numpy()
        loss = REF.integral_loss.SmoothL1JointLocationLoss(num_joints=j)(logits, gt, wt)
        loss.backward()
        out[name + "/loss"] = np.float32(loss.item())
        sd = model.state_dict()
        out[name + "/bn1.running_mean"] = sd["bn1.running_mean"].numpy()
        out[name + "/deconv_layers.7.running_var"] = sd["deconv_layers.7.running_var"].numpy()
        grads = {k: p.grad for k, p in model.named_parameters()}
        for k in golden_grad_keys(grads.keys()):
            g = grads[k].numpy()
            out[name + "/grad/" + k] = g.reshape(-1)[:: grad_stride(k, g.size)].copy()
        print(name, "loss", loss.item(), flush=True)
    save("network_big.npz", **out)


# --------------------------------------------------------------------------------------------------
# 6. evaluation: H36M_Integral.evaluate executed live on a synthetic ground-truth db
# --------------------------------------------------------------------------------------------------
def eval_scene(n_group, j, seed, noise_mm):
    """gt db records in the reference's format + noisy predictions in image coordinates."""
    sc = SyntheticScenes(n_group=n_group, n_view=4, num_joints=j, seed=seed, augment=False)
    rng = np.random.default_rng(seed + 1)
    gts, preds = [], []
    for i in range(sc.batch_size):
        v, g = divmod(i, n_group)
        cam = sc.cams[v]
        xc = (sc.world[g] - cam["T"].reshape(3)) @ cam["R"].T
        root = 6 if j == 16 else 0
        uv = xc[:, :2] / xc[:, 2:3] * cam["f"] + cam["c"]
        joints = np.concatenate([uv, xc[:, 2:3] - xc[root, 2]], axis=1)
        xp = xc + rng.normal(0, noise_mm, size=xc.shape)
        uvp = xp[:, :2] / xp[:, 2:3] * cam["f"] + cam["c"]
        preds.append(np.concatenate([uvp, xp[:, 2:3] - xc[root, 2], np.ones((j, 1))], axis=1))
        gts.append({"fl": cam["f"].copy(), "c_p": cam["c"].copy(), "pelvis": xc[root].copy(), "joints_3d": joints,
                    "joints_3d_vis": np.ones((j, 3))})
    return gts, np.asarray(preds)


def gen_evaluation():
    import importlib
    h36m = importlib.import_module("lib.dataset.h36m")
    out = {}
    for tag, j, mpii in (("h36m", 17, False), ("mpii", 16, True)):
        gts, preds = eval_scene(5, j, 200 + j, noise_mm=25.0)
        if mpii:   # the reference permutes the 17-joint gt into MPII order itself (h36m.py:219-220): give it 17-joint records
            gts17, _ = eval_scene(5, 17, 200 + j, noise_mm=25.0)
            perm = h36m.H36M_TO_MPII_PERM
            preds = np.stack([p for p in eval_scene(5, 17, 200 + j, noise_mm=25.0)[1]])[:, perm, :]
            gts = gts17
        obj = object.__new__(h36m.H36M_Integral)
        obj.db = gts
        obj.cfg = REF.EasyDict({"DATASET": {"MPII_ORDER": mpii}, "DEBUG": {"DEBUG": False}})
        obj.root = ""
        name_value, perf = h36m.H36M_Integral.evaluate(obj, preds.copy())
        out[tag + "/preds"] = preds
        out[tag + "/gt_joints"] = np.stack([g["joints_3d"] for g in gts])
        out[tag + "/pelvis"] = np.stack([g["pelvis"] for g in gts])
        out[tag + "/fl"] = np.stack([g["fl"] for g in gts])
        out[tag + "/c_p"] = np.stack([g["c_p"] for g in gts])
        out[tag + "/metrics"] = np.array([v for _, v in name_value])
        out[tag + "/names"] = np.array([k for k, _ in name_value])
        out[tag + "/perf"] = np.float64(perf)
    rng = np.random.default_rng(9)
    x = rng.normal(0, 300, size=(6, 17, 3))
    y = np.stack([1.3 * (xi @ np.linalg.qr(rng.normal(size=(3, 3)))[0]) + rng.normal(0, 20, size=xi.shape) + 50 for xi in x])
    out["procrustes/x"], out["procrustes/y"] = x, y
    res = [REF.prep_h36m.compute_similarity_transform(x[i], y[i], compute_optimal_scale=True) for i in range(6)]
    out["procrustes/T"] = np.stack([r[2] for r in res])
    out["procrustes/b"] = np.array([r[3] for r in res])
    out["procrustes/c"] = np.stack([r[4] for r in res])
    save("evaluation.npz", **out)



# --------------------------------------------------------------------------------------------------
# 7. refinement MLP: the reference's refiner/model.py executed live (fp32, CPU)
# --------------------------------------------------------------------------------------------------
REFINER_GRADS = ("w1.weight", "w1.bias", "w4.weight", "w2.bias", "batch_norm1.weight", "linear_stages.0.w2.weight",
                 "linear_stages.1.w3.weight", "linear_stages.1.batch_norm3.bias", "linear_stages.0.batch_norm1.weight")


def gen_refiner():
    import importlib
    rmodel = importlib.import_module("refiner.model")
    out = {}
    b, n = 64, 45
    x = torch.from_numpy(seeded_array("refiner/x", (b, n)))
    t = torch.from_numpy(seeded_array("refiner/t", (b, n)))
    model = rmodel.LinearModelPG()                                   # defaults of refiner/main.py:102: 1024 wide, 2 stages, dropout 0.5
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    out["keys"] = np.array(list(shapes.keys()))
    out["shapes"] = np.array([str(s) for s in shapes.values()])
    model.load_state_dict(fill_state_dict(shapes, seed=3))
    model.eval()
    with torch.no_grad():
        p1, p2 = model(x)
    out["eval/p1"], out["eval/p2"] = p1.numpy(), p2.numpy()
    m0 = rmodel.LinearModelPG(p_dropout=0.0)                         # training-mode arithmetic without the random mask
    m0.load_state_dict(fill_state_dict(shapes, seed=3))
    m0.train()
    opt = torch.optim.Adam(m0.parameters(), lr=1e-3)
    crit = torch.nn.MSELoss(reduction='mean')
    p1, p2 = m0(x)
    loss = crit(p1, t) + crit(p2, t)                                 # refiner/main.py:49
    opt.zero_grad()
    loss.backward()
    out["train/p1"], out["train/p2"], out["train/loss"] = p1.detach().numpy(), p2.detach().numpy(), np.float32(loss.item())
    grads = {k: p.grad.clone() for k, p in m0.named_parameters()}
    sub = lambda a: a if a.size <= 20000 else a.reshape(-1)[::97].copy()      # noqa: E731  (large matrices: every 97th element)
    for k in REFINER_GRADS:
        out["train/grad/" + k] = sub(grads[k].numpy())
    norm = torch.nn.utils.clip_grad_norm_(m0.parameters(), max_norm=1.)      # refiner/main.py:53
    out["train/grad_norm"] = np.float32(float(norm))
    before = {k: v.detach().clone() for k, v in m0.named_parameters()}
    opt.step()
    for k in ("w4.weight", "w1.bias", "linear_stages.0.w1.weight"):
        out["train/delta/" + k] = sub((dict(m0.named_parameters())[k].detach() - before[k]).numpy())
    out["train/running_mean"] = m0.state_dict()["batch_norm1.running_mean"].numpy()
    save("refiner.npz", **out)


# --------------------------------------------------------------------------------------------------
# 8. training trajectory (VERDICT round 2, item 2 iii): the reference's model, criterion and torch.optim.Adam stepped 20 times
#    (scripts/train.py:105 -> lib/core/function.py:23-52, lib/utils/utils.py:55-59) on one fixed synthetic batch, fp32 on the CPU
# --------------------------------------------------------------------------------------------------
def gen_trajectory():
    out = {}
    for name, layers, image, j, d, b, steps, lr in TRAJECTORY_CASES:
        model = REF.pose3d_resnet.get_pose_net(ref_cfg(layers, image, j, d), is_train=True)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(fill_state_dict(shapes, seed=2, head_std=TRAJECTORY_HEAD_STD))
        model.train()
        x = torch.from_numpy(seeded_array("img/traj/" + name, (b, 3, image, image)))
        gt = torch.from_numpy(seeded_array("gt/traj/" + name, (b, 3 * j), scale=0.2))
        wt = torch.ones(b, 3 * j)
        crit = REF.integral_loss.SmoothL1JointLocationLoss(num_joints=j)
        opt = torch.optim.Adam(model.parameters(), lr=lr)              # utils.py:55-59: Adam(lr), no weight decay
        losses = []
        for _ in range(steps):
            opt.zero_grad()
            loss = crit(model(x), gt, wt)
            loss.backward()
            opt.step()
            losses.append(float(loss.item()))
        with torch.no_grad():
            final = crit(model(x), gt, wt)                              # (train mode: one more statistics update, as a 21st forward would)
        losses.append(float(final.item()))
        out[name + "/losses"] = np.array(losses, dtype=np.float64)
        sd = model.state_dict()
        for k in ("conv1.weight", "layer2.0.conv1.weight", "final_layer.weight", "final_layer.bias", "bn1.running_mean", "deconv_layers.7.running_var",
                  "layer4.1.bn2.weight"):
            v = sd[k].numpy().reshape(-1)
            out[name + "/end/" + k] = v[:: max(1, v.size // 20000)].copy()
        print(name, "losses", " ".join("%.5f" % v for v in losses), flush=True)
    save("trajectory.npz", **out)


# --------------------------------------------------------------------------------------------------
# 9. input pipeline: the reference's do_augmentation, paste_over, occlude_with_objects and
#    get_single_patch_sample executed live.  Third-party stand-ins (no cv2 here, see ref_shims.py): cv2.imread -> a synthetic frame,
#    cv2.warpAffine and cv2.resize(INTER_AREA) -> the restatements of oracle/imgproc.py (the unpinned OpenCV layer); everything else --
#    the order of the random draws, the affine, the alpha blend with its uint8 truncation, colour scaling, normalisation, the label
#    arithmetic -- is the reference's own code.
# --------------------------------------------------------------------------------------------------
def gen_pipeline():
    import importlib
    import random
    from oracle import imgproc as o_img
    from epipolarpose_amd.dataset.synthetic_frames import render_frame
    from epipolarpose_amd.utils.augmentation import load_occluders
    aug = importlib.import_module("lib.utils.augmentation")
    iu = REF.img_utils
    cv2 = sys.modules["cv2"]
    out = {}
    random.seed(11)
    np.random.seed(11)
    draws = [iu.do_augmentation() for _ in range(64)]
    out["aug/seed"] = np.int64(11)
    out["aug/draws"] = np.array([[d[0], d[1], float(d[2])] + list(d[3]) for d in draws])
    rng = np.random.default_rng(3)
    for t in range(8):
        src = rng.integers(0, 256, (int(rng.integers(5, 40)), int(rng.integers(5, 40)), 4)).astype(np.uint8)
        dst = rng.integers(0, 256, (48, 64, 3)).astype(np.uint8)
        center = rng.uniform(-10, 70, 2)
        res = dst.copy()
        aug.paste_over(src, res, center)
        out["paste/%d/src" % t], out["paste/%d/dst" % t], out["paste/%d/center" % t], out["paste/%d/out" % t] = src, dst, center, res
    occluders = load_occluders(seed=5, count=6)
    aug.resize_by_factor = o_img.resize_by_factor                    # cv2.resize stand-in
    im = rng.integers(0, 256, (256, 256, 3)).astype(np.uint8)
    random.seed(21)
    np.random.seed(21)
    out["occlude/im"], out["occlude/seed"] = im, np.int64(21)
    out["occlude/out"] = aug.occlude_with_objects(im, occluders)
    # configs[4]'s patch size: im_scale_factor = 1.5, so part of the draws GROW the occluder (resize_by_factor's INTER_LINEAR branch, :122)
    im384 = rng.integers(0, 256, (384, 384, 3)).astype(np.uint8)
    random.seed(22)
    np.random.seed(22)
    out["occlude384/im"], out["occlude384/seed"] = im384, np.int64(22)
    out["occlude384/out"] = aug.occlude_with_objects(im384, occluders)
    # get_single_patch_sample on two synthetic frames, with and without occluders
    sc = SyntheticScenes(n_group=2, n_view=2, num_joints=17, seed=31, augment=False)
    frames = {}
    cv2.imread = lambda path, flags=None: frames[path]
    cv2.IMREAD_COLOR, cv2.IMREAD_IGNORE_ORIENTATION = 1, 128
    cv2.INTER_LINEAR = 1
    cv2.warpAffine = lambda img, trans, dsize, flags=None: o_img.warp_affine_linear(np.ascontiguousarray(img), trans, dsize)
    mean, std = np.array([123.675, 116.280, 103.530]), np.array([58.395, 57.120, 57.375])
    label_func = REF.integral_loss.get_label_func()
    from epipolarpose_amd.synthetic import project
    for i in range(sc.batch_size):
        v, g = divmod(i, 2)
        uv, xc = project(sc.world[g], sc.cams[v])
        joints = np.concatenate([uv, xc[:, 2:3] - xc[0, 2]], axis=1)
        frames["f%d" % i] = render_frame(uv, 1000, seed=500 + i)
        for occ in (False, True):
            random.seed(40 + i)
            np.random.seed(40 + i)
            img, label, weight, scale, rot = iu.get_single_patch_sample(
                "f%d" % i, sc.meta["center_x"][i], sc.meta["center_y"][i], sc.meta["width"][i], sc.meta["height"][i], joints.copy(),
                np.ones((17, 3)), [], None, 256, 256, 2000., 2000., mean, std, True, label_func, occluder=occluders if occ else None)
            tag = "sample/%d/occ%d" % (i, int(occ))
            out[tag + "/img_crc"] = np.uint32(zlib.crc32(np.ascontiguousarray(img, np.float32).tobytes()))
            out[tag + "/img_sub"] = np.ascontiguousarray(img, np.float32)[:, ::8, ::8].copy()
            out[tag + "/label"], out[tag + "/weight"] = label.astype(np.float32), weight.astype(np.float32)
            out[tag + "/scale_rot"] = np.array([scale, rot], np.float64)
    out["sample/seed_base"] = np.int64(40)
    save("pipeline.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["integral", "triangulation", "geometry", "maxpreds", "network", "evaluation", "network_big", "refiner", "trajectory", "pipeline"]
    for w in which:
        globals()["gen_" + w]()
