"""Phase stamps of the experimental ring-staged 3x3 weight-gradient kernel (tools/experiments/wgrad3x3_ring.hip).  Needs that kernel in the library WITH four
s_memrealtime stamps of wave 0 (entry, after the prologue, after the K loop, after the drained result stores; K tiles in slot 4) in a __device__ array
epi_w3_trace[4096 * 8] and an extern "C" epi_w3_trace_read(out) -- the patch is a dozen lines and was not kept."""
import ctypes, sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from epipolarpose_amd import hip
lib = hip.load()
dev = torch.device('cuda:0')
for (b, h, cin, cout) in ((32, 64, 64, 64), (32, 16, 256, 256)):
    x = torch.randn(b, cin, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(b, cout, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        hip.conv2d_bwd_weight(x, dy, 3, 1, 1, dtype=torch.bfloat16)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (4096 * 8))()
    lib.epi_w3_trace_read.restype = ctypes.c_int
    assert lib.epi_w3_trace_read(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
    used = a[a[:, 3] > 0]
    t = used[:, :4] * 0.01          # s_memrealtime: 100 MHz -> 10 ns ticks -> us
    nk = used[:, 4]
    print("shape B%d H%d %d->%d: %d workgroups, K tiles per workgroup %d..%d" % (b, h, cin, cout, len(used), nk.min(), nk.max()))
    for name, lo, hi in (("prologue (ring fill, first tiles issued)", 0, 1), ("K loop", 1, 2), ("epilogue (stores drained)", 2, 3), ("whole", 0, 3)):
        d = t[:, hi] - t[:, lo]
        print("   %-42s p10 %6.2f  p50 %6.2f  p90 %6.2f us" % (name, np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90)))
    print("   first start -> last end %.2f us; K loop per tile p50 %.2f us" % (t[:, 3].max() - t[:, 0].min(), np.median((t[:, 2] - t[:, 1]) / nk)))
