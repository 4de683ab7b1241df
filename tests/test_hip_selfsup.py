"""GPU parity: self-supervision geometry kernels (through the C ABI) vs. golden vectors / the oracle.
Tolerances: triangulated 3-D <= 1e-2 mm abs for the fp32 path; the float64 path (the
drop-in default, same precision as the reference) is held to 1e-6 mm."""
import numpy as np
import pytest
import torch

from oracle import geometry as o_geo
from oracle import triangulation as o_tri

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run through gpurun)"
    from epipolarpose_amd import hip
    hip.load()
    return torch.device("cuda:0")


def scene(n_group, j, seed, n_view=2, **kw):
    from epipolarpose_amd.synthetic import SyntheticScenes
    return SyntheticScenes(n_group=n_group, n_view=n_view, num_joints=j, seed=seed, **kw)


def patch_to_xyz(cp):
    return np.stack([cp[:, :, 0] / 256 - 0.5, cp[:, :, 1] / 256 - 0.5, cp[:, :, 2] / 256], axis=2).reshape(cp.shape[0], -1)


def test_decode_to_image_golden(golden, dev):
    from epipolarpose_amd import hip
    g = golden("geometry")
    p = g["affine/params"]
    meta = {"center_x": p[:, 0], "center_y": p[:, 1], "width": p[:, 2], "height": p[:, 3], "scale": p[:, 4], "rot": p[:, 5]}
    xyz = patch_to_xyz(g["decode/coords_patch"]).astype(np.float32)
    out = hip.decode_to_image(torch.from_numpy(xyz).to(dev), hip.DeviceMeta(meta, dev)).cpu().numpy()
    ref = g["decode/coords_img"][:, :, :3]
    np.testing.assert_allclose(out, ref, atol=2e-4)          # input rounded to f32: 256 px * 2^-24 * |affine|
    # exact check against the oracle fed the same float32 input
    cp = np.ones(g["decode/coords_patch"].shape)
    x64 = xyz.astype(np.float64).reshape(12, 17, 3)
    cp[:, :, 0], cp[:, :, 1], cp[:, :, 2] = (x64[:, :, 0] + 0.5) * 256, (x64[:, :, 1] + 0.5) * 256, x64[:, :, 2] * 256
    np.testing.assert_allclose(out, o_geo.decode_to_image(cp, meta)[:, :, :3], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("dtype,atol", [(torch.float64, 1e-6), (torch.float32, 1e-2)], ids=["f64", "f32"])
@pytest.mark.parametrize("noise", [0, 2])
def test_two_view_triangulators_golden(golden, dev, dtype, atol, noise):
    from epipolarpose_amd import hip
    g = golden("triangulation")
    u, ps = g["u/noise%d" % noise], g["P"]
    for va, vb in ((0, 1), (0, 3), (1, 2)):
        us = torch.from_numpy(np.concatenate([u[va], u[vb]])).to(dev, dtype)              # [2*G, J, 2] view-major
        pp = torch.from_numpy(np.concatenate([np.repeat(ps[va][None], 3, 0), np.repeat(ps[vb][None], 3, 0)])).to(dev, dtype)
        for method, key in (("iterative", "iter_x"), ("ls", "ls_x"), ("dlt", "eigen_x")):
            x, st = hip.triangulate(us, pp, 2, method)
            for grp in range(3):
                tag = "noise%d/v%d%d/g%d" % (noise, va, vb, grp)
                np.testing.assert_allclose(x[grp].double().cpu().numpy(), g[tag + "/" + key], atol=atol)
                if method == "iterative":
                    np.testing.assert_array_equal(st[grp].cpu().numpy(), g[tag + "/iter_status"])
                if method == "dlt":
                    np.testing.assert_array_equal(st[grp].cpu().numpy().astype(bool), g[tag + "/eigen_status"])


def test_reference_signature_and_status_codes(golden, dev):
    from epipolarpose_amd.utils import triangulation as tri
    g = golden("triangulation")
    x, st = tri.iterative_LS_triangulation(g["behind/u0"], g["P"][0], g["behind/u1"], g["P"][1])
    np.testing.assert_array_equal(st, g["behind/status"])
    np.testing.assert_allclose(x, g["behind/x"], rtol=1e-7, atol=1e-5)
    assert x.dtype == np.float64
    x, st = tri.linear_LS_triangulation(g["u/noise2"][0, 0], g["P"][0], g["u/noise2"][1, 0], g["P"][1])
    np.testing.assert_allclose(x, g["noise2/v01/g0/ls_x"], atol=1e-6)
    assert st.dtype == bool and st.all()
    x, st = tri.linear_eigen_triangulation(g["u/noise2"][0, 0], g["P"][0], g["u/noise2"][1, 0], g["P"][1])
    np.testing.assert_allclose(x, g["noise2/v01/g0/eigen_x"], atol=1e-6)


@pytest.mark.parametrize("n_view", [2, 3, 4, 6])
def test_multiview_vs_oracle(dev, n_view):
    from epipolarpose_amd import hip
    sc = scene(5, 17, 100 + n_view, n_view=n_view, noise_px=1.5)
    kps = torch.from_numpy(sc.kps_img).to(dev)
    pm = torch.from_numpy(sc.meta["projection_matrix"]).to(dev)
    for method, fn in (("iterative", o_tri.iterative_ls_triangulation), ("ls", o_tri.linear_ls_triangulation),
                       ("dlt", o_tri.dlt_triangulation)):
        x, st = hip.triangulate(kps, pm, n_view, method)
        x32, _ = hip.triangulate(kps.float(), pm.float(), n_view, method)
        for grp in range(5):
            idx = [v * 5 + grp for v in range(n_view)]
            ref, rst = fn(sc.kps_img[idx], sc.meta["projection_matrix"][idx])
            np.testing.assert_allclose(x[grp].cpu().numpy(), ref, atol=1e-6)
            np.testing.assert_allclose(x32[grp].double().cpu().numpy(), ref, atol=1e-2)
            if method == "iterative":
                np.testing.assert_array_equal(st[grp].cpu().numpy(), rst)


@pytest.mark.parametrize("tag,n_group,j", [("h36m", 3, 17), ("mpii", 2, 16)])
def test_reprojection_and_fused_ss_golden(golden, dev, tag, n_group, j):
    from epipolarpose_amd import hip
    g = golden("geometry")
    sc = scene(n_group, j, 31 + j)
    meta = hip.DeviceMeta(sc.meta, dev)
    lab, wt = hip.reproject_labels(torch.from_numpy(sc.world).to(dev), meta, 2)
    np.testing.assert_allclose(lab.cpu().numpy(), g[tag + "/labels_from_world/label"], atol=1e-7)
    np.testing.assert_array_equal(wt.cpu().numpy(), g[tag + "/labels_from_world/weight"])
    # staged: decode -> triangulate -> re-project, against the reference's own intermediate results
    xyz = torch.from_numpy(patch_to_xyz(g[tag + "/ss/coords_patch"]).astype(np.float32)).to(dev)
    kps = hip.decode_to_image(xyz, meta)
    np.testing.assert_allclose(kps.cpu().numpy(), g[tag + "/ss/kps_img"][:, :, :3], atol=2e-4)
    x, st = hip.triangulate(kps, meta.tensors["projection_matrix"], 2, "iterative")
    np.testing.assert_allclose(x.cpu().numpy(), g[tag + "/ss/x_world"][:n_group], atol=5e-3)   # f32-rounded input
    lab2, _ = hip.reproject_labels(x, meta, 2)
    np.testing.assert_allclose(lab2.cpu().numpy(), g[tag + "/ss/label"], atol=2e-6)
    # fused single launch == staged
    lab3, wt3, xw = hip.self_supervision(xyz, meta, 2, "iterative", want_world=True)
    np.testing.assert_allclose(xw.cpu().numpy(), x.cpu().numpy(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(lab3.cpu().numpy(), lab2.cpu().numpy(), rtol=0, atol=1e-7)
    assert torch.equal(wt3, torch.ones_like(wt3))
    # the same fused launch with the polynomial (optimal) two-view solver == staged poly solve == oracle pipeline
    lab4, _, xw4 = hip.self_supervision(xyz, meta, 2, "poly", want_world=True)
    x4, _ = hip.triangulate(kps, meta.tensors["projection_matrix"], 2, "poly")
    np.testing.assert_allclose(xw4.cpu().numpy(), x4.cpu().numpy(), rtol=0, atol=1e-7)
    ref_x = o_tri.triangulate_pairs(kps.cpu().numpy(), sc.meta["projection_matrix"], n_view=2, method="poly")
    np.testing.assert_allclose(xw4.cpu().numpy(), ref_x[:n_group], atol=1e-5)
    np.testing.assert_allclose(lab4.cpu().numpy(), o_geo.labels_from_global_coords(ref_x, sc.meta)[0], atol=1e-6)


def test_self_supervision_from_logits_golden(golden, dev):
    from epipolarpose_amd.utils import img_utils
    g = golden("geometry")
    sc = scene(2, 5, 77)
    logits = torch.from_numpy(g["ss_full/logits"]).to(dev)
    lab, wt = img_utils.self_supervision(logits, sc.meta)
    assert lab.dtype == np.float32 and lab.shape == (4, 15)
    np.testing.assert_allclose(lab, g["ss_full/label"], atol=5e-6)
    np.testing.assert_array_equal(wt, g["ss_full/weight"])
    lab_cl, _ = img_utils.self_supervision(logits.contiguous(memory_format=torch.channels_last), sc.meta)
    np.testing.assert_allclose(lab_cl, g["ss_full/label"], atol=5e-6)


def test_four_view_ss_vs_oracle_and_ground_truth(dev):
    from epipolarpose_amd import hip
    sc = scene(8, 17, 123, n_view=4)
    xyz32 = sc.label.copy()
    xyz32[:, 2::3] = xyz32[:, 2::3]          # z label has no -0.5 offset: decode uses z*pw, label z = pz/pw -> same
    meta = hip.DeviceMeta(sc.meta, dev)
    for method in ("iterative", "ls", "dlt"):
        lab, wt, xw = hip.self_supervision(torch.from_numpy(xyz32).to(dev), meta, 4, method, want_world=True)
        # exact 2-D in -> exact 3-D out (labels are f32: 256 px * 2^-24 -> ~1e-4 mm) -> labels reproduce themselves
        np.testing.assert_allclose(xw.cpu().numpy(), sc.world, atol=5e-3)
        np.testing.assert_allclose(lab.cpu().numpy(), sc.label, atol=2e-6)
        cp = np.ones((32, 17, 4))
        x64 = xyz32.astype(np.float64).reshape(32, 17, 3)
        cp[:, :, 0], cp[:, :, 1], cp[:, :, 2] = (x64[:, :, 0] + 0.5) * 256, (x64[:, :, 1] + 0.5) * 256, x64[:, :, 2] * 256
        olab, _, oxw, _ = o_geo.self_supervision(None, sc.meta, n_view=4, method=method, coords_patch=cp)
        np.testing.assert_allclose(xw.cpu().numpy(), oxw[:8], atol=1e-6)
        np.testing.assert_allclose(lab.cpu().numpy(), olab, atol=1e-7)


def test_bulk_properties(dev):
    """2^16 groups x 4 views: view-permutation invariance (DLT / LS), row-scaling invariance of P (DLT),
    noise-free recovery in fp32 within 1e-2 mm."""
    from epipolarpose_amd import hip
    from epipolarpose_amd.synthetic import make_cameras, project
    g_n, j, v_n = 1 << 16, 17, 4
    gen = torch.Generator().manual_seed(3)
    world = torch.randn((g_n, j, 3), generator=gen, dtype=torch.float64) * 300 + torch.tensor([0.0, 0.0, 900.0], dtype=torch.float64)
    cams = make_cameras(v_n)
    kps, pm = [], []
    for c in cams:
        uv, _ = project(world.reshape(-1, 3).numpy(), c)
        kps.append(torch.from_numpy(uv).reshape(g_n, j, 2))
        pm.append(torch.from_numpy(c["projection_matrix"]).expand(g_n, 3, 4))
    kps = torch.cat(kps).to(dev)
    pm = torch.cat(pm).contiguous().to(dev)
    world = world.to(dev)
    for method in ("iterative", "ls", "dlt"):
        x32, st = hip.triangulate(kps.float(), pm.float(), v_n, method)
        assert (x32.double() - world).abs().max().item() <= 1e-2
        assert (st == 1).all()
    noisy = kps + torch.randn(kps.shape, generator=gen, dtype=torch.float64).to(dev) * 2.0
    perm = [2, 0, 3, 1]
    kp_p = noisy.reshape(v_n, g_n, j, 2)[perm].reshape(-1, j, 2).contiguous()
    pm_p = pm.reshape(v_n, g_n, 3, 4)[perm].reshape(-1, 3, 4).contiguous()
    for method in ("ls", "dlt"):
        a, _ = hip.triangulate(noisy, pm, v_n, method)
        b, _ = hip.triangulate(kp_p, pm_p, v_n, method)
        assert (a - b).abs().max().item() <= 1e-6
    a, _ = hip.triangulate(noisy, pm, v_n, "dlt")
    b, _ = hip.triangulate(noisy, pm * 3.7, v_n, "dlt")
    assert (a - b).abs().max().item() <= 1e-6


# ------------------------------------------------------------------ polynomial (optimal) two-view triangulation
def test_polynomial_triangulation_golden_and_reference_signature(golden, dev):
    """triangulation.py:184-220 -- golden vectors from the live reference (tests/golden/make_golden.py)."""
    from epipolarpose_amd import hip
    from epipolarpose_amd.utils import triangulation as tri
    g = golden("triangulation")
    x, st = tri.polynomial_triangulation(g["poly/u1"], g["poly/P1"], g["poly/u2"], g["poly/P2"])
    np.testing.assert_allclose(x, g["poly/X"], rtol=1e-9, atol=1e-6)
    np.testing.assert_array_equal(st, g["poly/status"].astype(bool))
    # cv2.correctMatches mirror: [1, N, 2] in and out; result is exactly epipolar and equals the oracle's
    c1, c2 = tri.correct_matches(g["poly/F"], g["poly/u1"][None], g["poly/u2"][None])
    assert c1.shape == (1,) + g["poly/u1"].shape
    o1, o2 = o_tri.correct_matches(g["poly/F"], g["poly/u1"], g["poly/u2"])
    np.testing.assert_allclose(c1[0], o1, atol=1e-9)
    np.testing.assert_allclose(c2[0], o2, atol=1e-9)
    # f32 storage (arithmetic stays f64): inputs rounded to f32 move the answer by ~1e-4 px * depth/f
    us = torch.from_numpy(np.stack([g["poly/u1"], g["poly/u2"]])).to(dev)
    pp = torch.from_numpy(np.stack([g["poly/P1"], g["poly/P2"]])).to(dev)
    x32, _ = hip.triangulate(us.float(), pp.float(), 2, "poly")
    np.testing.assert_allclose(x32[0].double().cpu().numpy(), g["poly/X"], atol=0.5)
    with pytest.raises(RuntimeError):
        hip.triangulate(torch.cat([us, us[:1]]), torch.cat([pp, pp[:1]]), 3, "poly")        # two views only


@pytest.mark.parametrize("noise", [0.0, 1.0, 20.0])
def test_polynomial_triangulation_vs_oracle_batched(dev, noise):
    from epipolarpose_amd import hip
    sc = scene(6, 17, 321, n_view=2, noise_px=noise)
    kps = torch.from_numpy(sc.kps_img).to(dev)
    pm = torch.from_numpy(sc.meta["projection_matrix"]).to(dev)
    x, st = hip.triangulate(kps, pm, 2, "poly")
    for grp in range(6):
        idx = [grp, 6 + grp]
        ref, rst = o_tri.polynomial_triangulation(sc.kps_img[idx][:, :, :2], sc.meta["projection_matrix"][idx])
        np.testing.assert_allclose(x[grp].cpu().numpy(), ref, atol=1e-4 if noise == 0.0 else 1e-6)
        np.testing.assert_array_equal(st[grp].cpu().numpy().astype(bool), rst)


def test_polynomial_bulk_properties(dev):
    """2^18 pairs: corrected matches are exactly epipolar, at the Sampson distance to first order, invariant to the
    scale of F; poly == dlt on noise-free input; swapping the two views gives the same 3-D point."""
    from epipolarpose_amd import hip
    from epipolarpose_amd.synthetic import make_cameras, project
    g_n, j = 1 << 14, 16
    gen = torch.Generator().manual_seed(9)
    world = torch.randn((g_n, j, 3), generator=gen, dtype=torch.float64) * 300 + torch.tensor([0.0, 0.0, 900.0], dtype=torch.float64)
    cams = make_cameras(4)
    ca, cb = cams[0], cams[2]
    ua = torch.from_numpy(project(world.reshape(-1, 3).numpy(), ca)[0]).reshape(g_n, j, 2)
    ub = torch.from_numpy(project(world.reshape(-1, 3).numpy(), cb)[0]).reshape(g_n, j, 2)
    pa, pb = torch.from_numpy(ca["projection_matrix"]), torch.from_numpy(cb["projection_matrix"])
    f = torch.from_numpy(o_tri.fundamental_from_projections(pa.numpy(), pb.numpy()))
    kps = torch.cat([ua, ub]).to(dev)
    pm = torch.cat([pa.expand(g_n, 3, 4), pb.expand(g_n, 3, 4)]).contiguous().to(dev)
    xp, st = hip.triangulate(kps, pm, 2, "poly")
    xd, _ = hip.triangulate(kps, pm, 2, "dlt")
    assert (st == 1).all()
    assert (xp - world.to(dev)).abs().max().item() <= 1e-3 and (xp - xd).abs().max().item() <= 1e-3
    noisy = kps + torch.randn(kps.shape, generator=gen, dtype=torch.float64).to(dev) * 3.0
    n1, n2 = noisy[:g_n].contiguous(), noisy[g_n:].contiguous()
    fm = f.to(dev).expand(g_n, 3, 3).contiguous()
    c1, c2 = hip.correct_matches(fm, n1, n2)

    def hom(u):
        return torch.cat([u, torch.ones_like(u[..., :1])], dim=-1)
    fd = f.to(dev)
    resid = torch.einsum("gji,ik,gjk->gj", hom(c2), fd, hom(c1)).abs() / fd.abs().max()
    assert resid.max().item() < 1e-9 * (1 + noisy.abs().max().item()) ** 2
    moved = ((c1 - n1) ** 2).sum(-1) + ((c2 - n2) ** 2).sum(-1)
    e = torch.einsum("gji,ik,gjk->gj", hom(n2), fd, hom(n1))
    fx1, ftx2 = hom(n1) @ fd.T, hom(n2) @ fd
    sampson = e ** 2 / (fx1[..., 0] ** 2 + fx1[..., 1] ** 2 + ftx2[..., 0] ** 2 + ftx2[..., 1] ** 2)
    rel = (moved - sampson).abs() / (sampson + 1e-12)           # Sampson = first-order approximation of the same distance
    assert rel.median().item() <= 1e-3 and (rel <= 5e-2).double().mean().item() >= 0.98
    s1, s2 = hip.correct_matches((fm * -41.0).contiguous(), n1, n2)
    assert (s1 - c1).abs().max().item() <= 1e-8 and (s2 - c2).abs().max().item() <= 1e-8
    a, _ = hip.triangulate(noisy, pm, 2, "poly")
    b, _ = hip.triangulate(torch.cat([n2, n1]), torch.cat([pm[g_n:], pm[:g_n]]).contiguous(), 2, "poly")
    assert (a - b).abs().max().item() <= 1e-3 and (a - b).abs().median().item() <= 1e-9    # worst of 2^18: near-multiple root
    # the optimal method is never worse than the linear one in reprojection error (it minimises it over epipolar pairs)
    def reproj(x):
        ra = torch.from_numpy(project(x.reshape(-1, 3).cpu().numpy(), ca)[0]).reshape(g_n, j, 2).to(dev)
        rb = torch.from_numpy(project(x.reshape(-1, 3).cpu().numpy(), cb)[0]).reshape(g_n, j, 2).to(dev)
        return ((ra - n1) ** 2).sum(-1) + ((rb - n2) ** 2).sum(-1)
    d, _ = hip.triangulate(noisy, pm, 2, "dlt")
    assert (reproj(a) <= reproj(d) + 1e-6).all()


def test_fundamental_8point_vs_oracle(dev):
    """cv2.findFundamentalMat(FM_8POINT) mirror: batched kernel == oracle; exact on noise-free matches; degenerate status."""
    from epipolarpose_amd import hip
    from epipolarpose_amd.synthetic import make_cameras, project
    from epipolarpose_amd.utils import triangulation as tri
    cams = make_cameras(4)
    rng = np.random.default_rng(5)
    g_n, j = 64, 17
    world = rng.normal(0, 300, size=(g_n, j, 3)) + [0, 0, 900]
    u1 = np.stack([project(world[g], cams[g % 4])[0] for g in range(g_n)]) + rng.normal(0, 1.0, (g_n, j, 2))
    u2 = np.stack([project(world[g], cams[(g + 1) % 4])[0] for g in range(g_n)]) + rng.normal(0, 1.0, (g_n, j, 2))
    u1[5] = u1[5, :1]                                    # coincident points in view 1 -> degenerate
    f, st = hip.fundamental_8point(torch.from_numpy(u1).to(dev), torch.from_numpy(u2).to(dev))
    f, st = f.cpu().numpy(), st.cpu().numpy()
    for g in range(g_n):
        ref, ok = o_tri.fundamental_8point(u1[g], u2[g])
        assert bool(st[g]) == ok
        np.testing.assert_allclose(f[g], ref, rtol=0, atol=1e-10 * max(np.abs(ref).max(), 1e-300))
    assert st[5] == 0 and st.sum() == g_n - 1
    # reference-style wrappers
    x = rng.normal(0, 300, size=(12, 3)) + [0, 0, 900]
    a, b = project(x, cams[0])[0], project(x, cams[2])[0]
    fm, ok = tri.find_fundamental_mat_8point(a, b)
    ft = o_tri.fundamental_from_projections(cams[0]["projection_matrix"], cams[2]["projection_matrix"])
    assert ok
    np.testing.assert_allclose(fm, ft / ft[2, 2], atol=1e-8 * np.abs(ft / ft[2, 2]).max())
    k = np.array([[1100.0, 0, 500.0], [0, 1100.0, 510.0], [0, 0, 1.0]])
    np.testing.assert_allclose(tri.essential_matrix(fm, k), k.T @ fm @ k)


@pytest.mark.parametrize("n,outliers", [(60, 20), (17, 0), (1001, 400), (7, 0)])
def test_fundamental_lmeds_vs_oracle(dev, n, outliers):
    """cv2.findFundamentalMat(FM_LMEDS) mirror (cameras.py:136-143): host sampling + 7-point, GPU medians / errors == the oracle's plain restatement
    (same samples from OpenCV's fixed-seed generator, so the same matrix and the same mask), through `Camera.get_fundamental_matrix` as well."""
    from epipolarpose_amd import hip
    from epipolarpose_amd.synthetic import make_cameras, project
    from epipolarpose_amd.utils import triangulation as tri
    from epipolarpose_amd.utils.cameras import Camera
    cams = make_cameras(4)
    rng = np.random.default_rng(n + outliers)
    x = rng.normal(0, 300, size=(n, 3)) + [0, 0, 900]
    u1 = project(x, cams[0])[0] + rng.normal(0, 0.5, (n, 2))
    u2 = project(x, cams[2])[0] + rng.normal(0, 0.5, (n, 2))
    bad = rng.choice(n, outliers, replace=False)
    u2[bad] += rng.uniform(40, 120, (outliers, 2)) * rng.choice([-1, 1], (outliers, 2))
    i1, i2 = np.int32(u1), np.int32(u2)
    ref_f, ref_mask = o_tri.fundamental_lmeds(i1, i2)
    f, mask = tri.find_fundamental_mat_lmeds(i1, i2)
    assert ref_f is not None and f is not None and mask.shape == (n, 1)
    np.testing.assert_allclose(f, ref_f, rtol=1e-9, atol=1e-12 * np.abs(ref_f).max())
    assert np.array_equal(mask.ravel(), ref_mask)
    if outliers:
        good = np.setdiff1d(np.arange(n), bad)
        assert mask.ravel()[good].mean() >= 0.8 and mask.ravel()[bad].mean() <= 0.15
    # the scoring kernels against the oracle on arbitrary candidates (odd and even N take different median paths)
    cand = np.stack([ref_f, ref_f + 1e-7 * rng.normal(size=(3, 3)), rng.normal(size=(3, 3))])
    d1, d2 = torch.from_numpy(i1.astype(np.float64)).to(dev), torch.from_numpy(i2.astype(np.float64)).to(dev)
    med = hip.fundamental_lmeds_medians(torch.from_numpy(cand).to(dev), d1, d2).cpu().numpy()
    for h in range(3):
        e_ref = o_tri.fm_compute_error(cand[h], i1.astype(np.float32), i2.astype(np.float32))
        e = hip.fundamental_errors(torch.from_numpy(cand[h]).to(dev), d1, d2).cpu().numpy()
        np.testing.assert_allclose(e, e_ref, rtol=2e-6, atol=1e-10)        # (a sample point fits its own candidate: d = a cancellation at 1e-12)
        assert abs(med[h] - o_tri.fm_median(e)) <= 1e-12 * max(1.0, abs(med[h]))          # the median of the kernel's own errors, exactly
    for k in (n - 1, n - 2) if n > 8 else ():              # the other parity of N
        m2 = hip.fundamental_lmeds_medians(torch.from_numpy(cand[:1]).to(dev), d1[:k], d2[:k]).cpu().numpy()
        e = hip.fundamental_errors(torch.from_numpy(cand[0]).to(dev), d1[:k], d2[:k]).cpu().numpy()
        assert abs(m2[0] - o_tri.fm_median(e)) <= 1e-12 * max(1.0, abs(m2[0]))
    # the reference's method
    cam = Camera((np.eye(3), np.zeros((3, 1)), (1100.0, 1100.0), (500.0, 510.0), None, None, "cam"))       # cameras.py:8
    fm, (k1, k2) = cam.get_fundamental_matrix(u1, u2)
    np.testing.assert_allclose(fm, ref_f, rtol=1e-9, atol=1e-12 * np.abs(ref_f).max())
    assert k1.dtype == np.int32 and len(k1) == len(k2) == int(ref_mask.sum()) and np.array_equal(k1, i1[ref_mask == 1])


# ------------------------------------------------------------------ round 5: the staged bulk kernel against the per-item kernel
@pytest.mark.parametrize("n_view,g_n,j", [(2, 40, 17), (4, 64, 17), (4, 31, 16), (3, 300, 1), (6, 23, 13), (4, 1000, 3)])
def test_staged_bulk_kernel_matches_per_item_kernel(dev, n_view, g_n, j):
    """triangulation.py:8-27,34-97,104-181 through csrc/selfsup.hip:triangulate_staged_kernel (>= 256 items: LDS-staged projection matrices, vector
    key-point loads, 16-byte result stores) and through triangulate_kernel on the SAME inputs: the arithmetic is shared, so results agree to the last
    bits and status codes exactly -- ragged last workgroup, groups straddling workgroups (J = 17 / 13 / 3), one item per group (J = 1), float64 and float32 storage,
    and an unaligned key-point view that must fall back to the per-item kernel by itself."""
    from epipolarpose_amd import hip
    lib = hip.load()
    sc = scene(g_n, j, 500 + n_view + j, n_view=n_view, noise_px=2.0)
    kps = torch.from_numpy(sc.kps_img).to(dev)
    pm = torch.from_numpy(sc.meta["projection_matrix"]).to(dev)
    assert g_n * j >= 256
    methods = ["iterative", "ls", "dlt"] + (["poly"] if n_view == 2 else [])
    try:
        for dt in (torch.float64, torch.float32):
            k, p = kps[..., :2].to(dt).contiguous(), pm.to(dt).contiguous()        # two entries per key point: the vector-load path
            for method in methods:
                lib.epi_triangulate_staged(0)
                xi, si = hip.triangulate(k, p, n_view, method)
                for mode in (2, 1):                     # the staged kernel for every method / the default selection
                    lib.epi_triangulate_staged(mode)
                    xs, ss = hip.triangulate(k, p, n_view, method)
                    # (the same source arithmetic; the compiler contracts multiply-adds differently per instantiation -> last-bit differences)
                    tol = 1e-9 if dt == torch.float64 else 2e-4
                    assert (xs - xi).abs().max().item() <= tol, (method, dt, mode, (xs - xi).abs().max().item())
                    assert torch.equal(ss, si), (method, dt, mode)
                lib.epi_triangulate_staged(1)
            # three entries per key point (u, v, score): the scalar key-point path of the staged kernel
            k3 = torch.cat([k, torch.ones_like(k[..., :1])], dim=-1).contiguous()
            xs3, _ = hip.triangulate(k3, p, n_view, "ls")
            x2, _ = hip.triangulate(k, p, n_view, "ls")
            assert torch.equal(xs3, x2)
        # oracle on a few groups of the float64 run (the staged kernel is the default path)
        x, st = hip.triangulate(kps, pm, n_view, "iterative")
        for grp in (0, g_n // 2, g_n - 1):
            idx = [v * g_n + grp for v in range(n_view)]
            ref, rst = o_tri.iterative_ls_triangulation(sc.kps_img[idx], sc.meta["projection_matrix"][idx])
            np.testing.assert_allclose(x[grp].cpu().numpy(), ref, atol=1e-6)
            np.testing.assert_array_equal(st[grp].cpu().numpy(), rst)
    finally:
        lib.epi_triangulate_staged(1)
