"""GPU probe: what the memory system delivers for the access MIXES of the step's kernels, with plain torch kernels (no GEMM): write-only, read-only,
copy 1:1, read 1 : write 4 (a 64 -> 256 channel 1x1 convolution's traffic), read 4 : write 1 (256 -> 64), on buffers rotating through > 1.5 GB so that
nothing is L2 / MALL resident between launches.  Prints us per launch and TB/s of the bytes moved."""
import torch

dev = torch.device("cuda:0")
M = 131072                                   # rows of layer 1 at batch 32 (32 x 64 x 64)


def timeit(fn, n_rot, iters=24):
    for i in range(6):
        fn(i % n_rot)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i % n_rot)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def rot(shape, n):
    return [torch.randn(shape, device=dev).to(torch.bfloat16) for _ in range(n)]


for rows, tag in ((M, "M=131072"), (M // 4, "M=32768"), (M // 16, "M=8192")):
    n = 12 if rows == M else 24
    x64, x256 = rot((rows, 64), n), rot((rows, 256), n)
    y64, y256 = rot((rows, 64), n), rot((rows, 256), n)
    mb64, mb256 = rows * 64 * 2 / 1e6, rows * 256 * 2 / 1e6
    t = timeit(lambda i: y256[i].fill_(1.0), n)
    print("%-9s write-only  %6.1f MB            %7.2f us  %5.2f TB/s" % (tag, mb256, t, mb256 / t))
    t = timeit(lambda i: x256[i].float().sum() if False else torch.sum(x256[i], dtype=torch.float32), n)
    print("%-9s read-only   %6.1f MB            %7.2f us  %5.2f TB/s" % (tag, mb256, t, mb256 / t))
    t = timeit(lambda i: y256[i].copy_(x256[i]), n)
    print("%-9s copy 1:1    %6.1f + %6.1f MB   %7.2f us  %5.2f TB/s" % (tag, mb256, mb256, t, 2 * mb256 / t))
    t = timeit(lambda i: y256[i].view(rows, 4, 64).copy_(x64[i].view(rows, 1, 64).expand(rows, 4, 64)), n)
    print("%-9s read 1 : write 4  %6.1f + %6.1f MB   %7.2f us  %5.2f TB/s" % (tag, mb64, mb256, t, (mb64 + mb256) / t))
    t = timeit(lambda i: torch.sum(x256[i].view(rows, 4, 64), dim=1, out=y64[i]), n)
    print("%-9s read 4 : write 1  %6.1f + %6.1f MB   %7.2f us  %5.2f TB/s" % (tag, mb256, mb64, t, (mb64 + mb256) / t))
    t = timeit(lambda i: torch.add(x256[i], y256[i], out=y256[(i + 1) % n]), n)
    print("%-9s read 2 : write 1 (add) %6.1f x 3 MB   %7.2f us  %5.2f TB/s" % (tag, mb256, t, 3 * mb256 / t))
    del x64, x256, y64, y256
