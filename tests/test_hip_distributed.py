"""GPU: the N > 1 training choreography on ONE device -- two ranks (gloo) sharing cuda:0 run the real bucketed gradient path on the
hand-written kernels: per-dtype flat buckets, learned bucket hooks that flush the deferred slab sums and the grouped weight gradients,
the second stream, asynchronous all-reduce handles, FusedAdam on the bucket views (reference: nn.DataParallel, scripts/train.py:93-94,
loss normalisation lib/core/integral_loss.py:29,45).  The synchronised gradients of the first step are compared with the mean over the two
shards of single-process gradients; the parameters must be bit-identical on both ranks after three steps."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("layers", [18, 50])
def test_two_ranks_on_one_device_bucketed_path(tmp_path, layers):
    out = str(tmp_path / "report.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_same_device_worker.py"), "--out", out, "--layers", str(layers),
           "--image", "64" if layers == 18 else "128"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    r0, r1 = (json.load(open(out + ".rank%d" % r)) for r in range(2))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "two_ranks_one_device_r%d.json" % layers), "w") as f:
        json.dump([r0, r1], f, indent=1)
    assert r0["params_identical_across_ranks"] and r1["params_identical_across_ranks"]
    assert r0["learning_done"] and r0["hooks_after"] == r0["buckets"] <= 6          # one learned hook per bucket after the first step
    assert set(r0["bucket_dtypes"]) == {"torch.float32", "torch.bfloat16"}
    # bf16 activations + fp32 atomics make two runs of the same gradient differ by a few per cent in the early layers (see
    # tests/test_hip_conv.py); a bucket that left before its gradient was final, a missing 1/N or a dropped shard is O(1)
    # measured on MI355X: ResNet-18 min 0.9998 / median 0.99999; ResNet-50 (bf16 noise floor of 50 layers at batch 8) min 0.77 / median 0.92
    lo, med = (0.99, 0.999) if layers == 18 else (0.6, 0.85)
    assert r0["min_cos"] >= lo and r0["median_cos"] >= med, (r0["median_cos"], r0["worst"])
    assert 0.8 <= r0["norm_ratio_range"][0] and r0["norm_ratio_range"][1] <= 1.25, r0["norm_ratio_range"]
    assert all(l == l and l < 10 for l in r0["losses"] + r1["losses"])
