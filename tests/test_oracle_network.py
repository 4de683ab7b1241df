"""CPU: the fp32 torch restatement of the pose network (oracle.network) against the reference's outputs."""
import ast

import numpy as np
import pytest
import torch

from det_weights import fill_state_dict, seeded_array
from make_golden_cases import NETWORK_CASES
from oracle import network


def golden_state(g, name, seed=1):
    shapes = {k: ast.literal_eval(s) for k, s in zip(g[name + "/keys"].tolist(), g[name + "/shapes"].tolist())}
    return fill_state_dict(shapes, seed=seed)


@pytest.mark.parametrize("case", NETWORK_CASES, ids=[c[0] for c in NETWORK_CASES])
def test_network_forward_backward(golden, case):
    g = golden("network")
    name, layers, image, j, d, b = case
    torch.set_num_threads(8)
    sd = golden_state(g, name)
    x = torch.from_numpy(seeded_array("img/" + name, (b, 3, image, image)))
    with torch.no_grad():
        out = network.forward(sd, x, layers, training=False)
    ref = g[name + "/logits_eval"]
    np.testing.assert_allclose(out.numpy(), ref, atol=2e-4 * np.abs(ref).max())
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v)
              for k, v in sd.items()}
    stats = {}
    logits = network.forward(params, x, layers, training=True, new_stats=stats)
    ref = g[name + "/logits_train"]
    np.testing.assert_allclose(logits.detach().numpy(), ref, atol=2e-4 * np.abs(ref).max())
    gt = torch.from_numpy(seeded_array("gt/" + name, (b, 3 * j), scale=0.2))
    loss = network.joint_location_loss(logits, gt, torch.ones(b, 3 * j), j, "smoothl1")
    np.testing.assert_allclose(loss.item(), g[name + "/loss"], rtol=1e-4)
    loss.backward()
    np.testing.assert_allclose(stats["bn1"][0].numpy(), g[name + "/bn1.running_mean"], atol=1e-5)
    np.testing.assert_allclose(stats["deconv_layers.7"][1].numpy(), g[name + "/deconv_layers.7.running_var"],
                               rtol=1e-3)
    for k in ("final_layer.weight", "final_layer.bias", "deconv_layers.6.weight", "deconv_layers.0.weight"):
        got = params[k].grad.numpy()
        ref = g[name + "/grad/" + k]
        if ref.ndim == 1 and got.ndim > 1:
            got = got.reshape(-1)[:: max(1, got.size // 50000)]
        np.testing.assert_allclose(got, ref, atol=2e-3 * np.abs(ref).max() + 1e-9)
