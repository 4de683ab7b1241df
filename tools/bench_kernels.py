#!/usr/bin/env python
"""Micro-benchmarks of the hand-written kernels (run on the GPU box):  python tools/bench_kernels.py [what ...]
what: softargmax gemm deconv tri.  Prints one line per case: time, achieved GB/s or TFLOP/s, fraction of peak."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolarpose_amd import hip  # noqa: E402

DEV = torch.device("cuda:0")
HBM, MFMA = 8000.0, 2500.0


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def bench_softargmax():
    for dt, name in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        for cl in (False, True):
            x = torch.randn(32, 17 * 64, 64, 64, device=DEV).to(dt)
            if cl:
                x = x.contiguous(memory_format=torch.channels_last)
            nbytes = x.numel() * x.element_size()
            xyz, rmax, rsum = hip.softargmax3d_fwd(x, 17)
            g = torch.randn(32, 51, device=DEV)
            tf = timeit(lambda: hip.softargmax3d_fwd(x, 17))
            tb = timeit(lambda: hip.softargmax3d_bwd(x, 17, rmax, rsum, xyz, g))
            print("softargmax %-4s %-4s fwd %.4f ms %7.1f GB/s (%.3f)   bwd %.4f ms %7.1f GB/s (%.3f)" % (
                name, "nhwc" if cl else "nchw", tf, nbytes / tf / 1e6, nbytes / tf / 1e6 / HBM,
                tb, 2 * nbytes / tb / 1e6, 2 * nbytes / tb / 1e6 / HBM), flush=True)


def bench_gemm():
    for m, n, k in ((131072, 1088, 256), (131072, 256, 1088), (32768, 256, 1024), (8192, 8192, 8192), (4096, 4096, 4096)):
        a = torch.randn(m, k, device=DEV).to(torch.bfloat16)
        bt = torch.randn(n, k, device=DEV).to(torch.bfloat16)
        out = torch.empty(m, n, device=DEV, dtype=torch.bfloat16)
        t = timeit(lambda: hip.gemm_bf16(a, bt, out=out))
        tt = timeit(lambda: torch.matmul(a, bt.t(), out=out))
        fl = 2.0 * m * n * k
        print("gemm %6dx%5dx%5d  ours %.4f ms %7.1f TF (%.3f)   torch(hipBLASLt) %.4f ms %7.1f TF" % (
            m, n, k, t, fl / t / 1e9, fl / t / 1e9 / MFMA, tt, fl / tt / 1e9), flush=True)


def bench_conv1x1():
    """ResNet-50 bottleneck 1x1 convolutions at B=32: our GEMM entry points vs MIOpen through torch (fwd, bwd-data, bwd-weight)."""
    import torch.nn.functional as F
    for hw, cin, cout in ((64, 64, 256), (64, 256, 64), (32, 512, 128), (32, 128, 512), (16, 1024, 256), (16, 256, 1024),
                          (8, 2048, 512), (8, 512, 2048)):
        b = 32
        m = b * hw * hw
        x = torch.randn(b, cin, hw, hw, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 1, 1, device=DEV) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(b, cout, hw, hw, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        x2 = x.permute(0, 2, 3, 1).reshape(m, cin)
        dy2 = dy.permute(0, 2, 3, 1).reshape(m, cout)
        w2 = w.reshape(cout, cin)
        w2t = w2.t().contiguous()
        fl = 2.0 * m * cin * cout
        t_f = timeit(lambda: hip.gemm_bf16(x2, w2))
        t_b = timeit(lambda: hip.gemm_bf16(dy2, w2t))
        t_w = timeit(lambda: hip.gemm_tn_bf16(dy2, x2))
        xr = x.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        s_f = timeit(lambda: F.conv2d(xr, wr))
        y = F.conv2d(xr, wr)
        s_b = timeit(lambda: torch.autograd.grad(y, xr, dy, retain_graph=True))
        s_w = timeit(lambda: torch.autograd.grad(y, wr, dy, retain_graph=True))
        print("conv1x1 %3dx%-3d %4d->%-4d fwd ours %.4f (%5.0f TF) MIOpen %.4f | bwd-data ours %.4f MIOpen %.4f | bwd-w ours %.4f MIOpen %.4f ms" % (
            hw, hw, cin, cout, t_f, fl / t_f / 1e9, s_f, t_b, s_b, t_w, s_w), flush=True)


def bench_deconv():
    import torch.nn.functional as F
    for b, h, cin, cout in ((32, 8, 2048, 256), (32, 16, 256, 256), (32, 32, 256, 256)):
        x = torch.randn(b, cin, h, h, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cin, cout, 4, 4, device=DEV) * 0.02)
        wp, wb = hip.deconv_pack_weight(w)
        wb16 = w.to(torch.bfloat16)
        fl = 2.0 * b * h * h * 16 * cin * cout
        t = timeit(lambda: hip.deconv4x4s2_fwd(x, wp))
        tt = timeit(lambda: F.conv_transpose2d(x, wb16, None, stride=2, padding=1))
        dy = torch.randn(b, cout, 2 * h, 2 * h, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        t2 = timeit(lambda: hip.deconv4x4s2_bwd_data(dy, wb))
        print("deconv B%d %dx%d %d->%d  fwd ours %.4f ms %6.1f TF  MIOpen %.4f ms %6.1f TF | bwd-data ours %.4f ms %6.1f TF" % (
            b, h, h, cin, cout, t, fl / t / 1e9, tt, fl / tt / 1e9, t2, fl / t2 / 1e9), flush=True)


def bench_tri():
    g_n, j, v_n = 1 << 20, 17, 4
    kps = torch.rand(v_n * g_n, j, 2, device=DEV) * 1000
    from epipolarpose_amd.synthetic import make_cameras
    pm = torch.cat([torch.from_numpy(c["projection_matrix"]).float().expand(g_n, 3, 4) for c in make_cameras(v_n)]).contiguous().to(DEV)
    nbytes = (v_n * j * 2 + v_n * 12 + j * 3) * 4.0 * g_n
    lib = hip.load()
    for staged in (2, 0):       # round 5: the staged bulk kernel / the per-item kernel with vector loads / the per-item kernel of rounds 1-4
        lib.epi_triangulate_staged(staged)
        for method in ("ls", "iterative", "dlt"):
            t = timeit(lambda: hip.triangulate(kps, pm, v_n, method), iters=8, warm=2)
            print("triangulate %-9s f32 storage 2^20 groups x 4 views, %s kernel: %.3f ms  %7.1f GB/s (%.3f of HBM peak)" % (
                method, ("per-item", "default ", "staged  ")[staged], t, nbytes / t / 1e6, nbytes / t / 1e6 / HBM), flush=True)
    lib.epi_triangulate_staged(1)
    # polynomial (optimal) two-view solver: compute bound (root isolation, ~20k f64 flops per point), realistic 2 px noise
    from epipolarpose_amd.synthetic import project
    g2 = 1 << 16
    cams = make_cameras(2)
    world = torch.randn(g2 * j, 3, dtype=torch.float64) * 300 + torch.tensor([0.0, 0.0, 900.0], dtype=torch.float64)
    kp2 = torch.cat([torch.from_numpy(project(world.numpy(), c)[0]).reshape(g2, j, 2) for c in cams]) + torch.randn(2 * g2, j, 2) * 2.0
    kp2 = kp2.to(DEV)
    pm2 = torch.cat([torch.from_numpy(c["projection_matrix"]).expand(g2, 3, 4) for c in cams]).contiguous().to(DEV)
    for method in ("dlt", "poly"):
        t = timeit(lambda: hip.triangulate(kp2, pm2, 2, method), iters=5, warm=1)
        print("triangulate %-9s f64 2^16 groups x 2 views x 17 joints: %.3f ms  %6.1f Mpoints/s" % (method, t, g2 * j / t / 1e3), flush=True)


def bench_bn():
    """epi_bn_act_fwd / epi_bn_act_bwd called through the C ABI in the steady fwd -> bwd order (each direction clears the
    other's accumulator); stock nn.BatchNorm2d (+ add + relu) timed beside it."""
    lib = hip.load()
    shapes = (((32, 64, 128, 128), False), ((32, 64, 64, 64), False), ((32, 256, 64, 64), True), ((32, 256, 64, 64), False),
              ((32, 128, 32, 32), False), ((32, 512, 32, 32), True), ((32, 256, 16, 16), False), ((32, 1024, 16, 16), True),
              ((32, 512, 8, 8), False), ((32, 2048, 8, 8), True))
    for shape, res in shapes:
        b, c, h, w = shape
        mk = lambda: torch.randn(shape, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
        x, dy, y, dx = mk(), mk(), mk(), mk()
        r = mk() if res else None
        dres = mk() if res else None
        f32 = lambda n, v=0.0: torch.full((n,), v, dtype=torch.float32, device=DEV)  # noqa: E731
        gamma, beta, rm, rv = f32(c, 1.0), f32(c), f32(c), f32(c, 1.0)
        nbt = torch.zeros((), dtype=torch.long, device=DEV)
        stats, sums, bsums = f32(4 * c), f32(2 * c * hip.bn_sum_copies(c)), f32(2 * c)
        sp, st = stats.data_ptr(), torch.cuda.current_stream().cuda_stream
        R = b * h * w

        def fwd():
            rc = lib.epi_bn_act_fwd(x.data_ptr(), r.data_ptr() if res else None, R, c, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, 1, 1,
                                    rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), sp, sp + 4 * c, sp + 8 * c, sums.data_ptr(),
                                    bsums.data_ptr(), y.data_ptr(), st)
            assert rc == 0

        def bwd():
            rc = lib.epi_bn_act_bwd(dy.data_ptr(), x.data_ptr(), y.data_ptr() if res else None, R, c, gamma.data_ptr(), sp, sp + 4 * c,
                                    sp + 8 * c, 1, bsums.data_ptr(), dx.data_ptr(), dres.data_ptr() if res else None, sums.data_ptr(), None, st)
            assert rc == 0
        tf = tb = 0.0
        n = 20
        for it in range(n + 3):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record(); fwd(); e[1].record(); bwd(); e[2].record()
            torch.cuda.synchronize()
            if it >= 3:
                tf += e[0].elapsed_time(e[1]) / n
                tb += e[1].elapsed_time(e[2]) / n
        ref = torch.nn.BatchNorm2d(c).to(DEV)
        xg = x.clone().requires_grad_(True)
        rg = r.clone().requires_grad_(True) if res else None

        def stock_fwd():
            t = ref(xg)
            if res:
                t = t + rg
            return torch.relu(t)
        ts = timeit(stock_fwd)
        t = stock_fwd()
        tsb = timeit(lambda: torch.autograd.grad(t, [xg] + ([rg] if res else []) + [ref.weight, ref.bias], dy, retain_graph=True))
        nbytes = x.numel() * 2
        rf, rb = (3 if res else 2) + 1, (5 if res else 4) + (2 if res else 1)      # passes over the tensor: fwd (stats+apply), bwd
        print("bn %-18s res=%d  fwd ours %.4f ms (%5.0f GB/s) stock %.4f ms | bwd ours %.4f ms (%5.0f GB/s) stock %.4f ms" % (
            shape, res, tf, rf * nbytes / tf / 1e6, ts, tb, rb * nbytes / tb / 1e6, tsb), flush=True)


def bench_wrw1x1():
    """Weight gradient of the backbone's 1x1 convolutions: our TN GEMM (fp32 out, split over rows) vs MIOpen/CK through
    aten.convolution_backward (weight only)."""
    torch.backends.cudnn.benchmark = True
    shapes = [(131072, 64, 64, 1), (131072, 256, 64, 3), (131072, 64, 256, 2), (131072, 128, 256, 1), (32768, 512, 128, 4),
              (32768, 128, 512, 3), (32768, 256, 512, 1), (8192, 1024, 256, 6), (8192, 256, 1024, 5), (8192, 512, 1024, 1),
              (2048, 2048, 512, 3), (2048, 512, 2048, 2)]
    tot_o = tot_s = 0.0
    for r, cout, cin, count in shapes:
        hw = int((r // 32) ** 0.5)
        x = torch.randn(32, cin, hw, hw, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(32, cout, hw, hw, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, 1, 1, device=DEV).to(torch.bfloat16)
        a2, b2 = dy.permute(0, 2, 3, 1).reshape(r, cout), x.permute(0, 2, 3, 1).reshape(r, cin)
        to = timeit(lambda: hip.gemm_tn_bf16(a2, b2))
        ts = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
        fl = 2.0 * r * cout * cin
        tot_o += to * count
        tot_s += ts * count
        print("wrw1x1 R=%6d %4d<-%4d x%d  ours %.4f ms %6.1f TF   stock %.4f ms %6.1f TF" % (r, cout, cin, count, to, fl / to / 1e9, ts, fl / ts / 1e9),
              flush=True)
    print("wrw1x1 total per step: ours %.3f ms, stock %.3f ms (stock includes its fill / cast launches)" % (tot_o, tot_s))


def bench_wrw3x3():
    """Weight gradient of the backbone's 3x3 convolutions: epi_conv2d_bwd_weight (TN GEMM with a 9-tap gather) vs MIOpen."""
    torch.backends.cudnn.benchmark = True
    shapes = [(64, 64, 64, 1, 3), (128, 128, 64, 2, 1), (128, 128, 32, 1, 3), (256, 256, 32, 2, 1), (256, 256, 16, 1, 5),
              (512, 512, 16, 2, 1), (512, 512, 8, 1, 2)]
    tot_o = tot_s = 0.0
    for cin, cout, hw, stride, count in shapes:
        x = torch.randn(32, cin, hw, hw, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ho = (hw + 2 - 3) // stride + 1
        dy = torch.randn(32, cout, ho, ho, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, 3, 3, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        to = timeit(lambda: hip.conv2d_bwd_weight(x, dy, 3, stride, 1, dtype=torch.bfloat16))
        ts = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [stride, stride], [1, 1], [1, 1], False, [0, 0], 1,
                                                                [False, True, False]))
        fl = 2.0 * 32 * ho * ho * cout * cin * 9
        tot_o += to * count
        tot_s += ts * count
        print("wrw3x3 %4d->%4d @%3d s%d x%d  ours %.4f ms %6.1f TF   stock %.4f ms %6.1f TF" % (cin, cout, hw, stride, count, to, fl / to / 1e9,
                                                                                              ts, fl / ts / 1e9), flush=True)
    print("wrw3x3 total per step: ours %.3f ms, stock %.3f ms (stock includes its fill / cast launches)" % (tot_o, tot_s))


if __name__ == "__main__":
    hip.load()
    what = sys.argv[1:] or ["softargmax", "gemm", "deconv", "tri", "bn"]
    for w in what:
        globals()["bench_" + w]()
