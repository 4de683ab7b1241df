"""GPU probe: the residual-unit node against the per-stage nodes in deterministic mode (how many bf16 units apart, over repeats), and both against
a plain fp32 torch restatement of the unit on the same weights.  Prints one line per kind; used to set the bars of
tests/test_hip_conv.py::test_residual_unit_node_matches_per_stage_nodes."""
import sys

import torch

sys.path.insert(0, ".")
from epipolarpose_amd import hip                                      # noqa: E402
from epipolarpose_amd.models import pose3d_resnet as P                # noqa: E402

from tests.test_hip_conv import _UNIT_KINDS, _make_unit_pair, _run_unit        # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for det in (True, False):
    hip.set_deterministic(det)
    for kind in _UNIT_KINDS:
        prev = hip.glue().bn_dual_mode(0)
        unit, staged, ref, x = _make_unit_pair(kind)
        r = _run_unit(ref, x.float().cpu(), fp32=True)
        worst_self, worst_ref, worst_key, nonident, per_key, diff_keys = 0.0, 0.0, "", 0, {}, set()
        for _ in range(reps):
            a = _run_unit(unit, x)
            b = _run_unit(staged, x)
            for k in a:
                d = (a[k].float() - b[k].float()).abs().max().item() / max(a[k].float().abs().max().item(), 1e-9)
                worst_self = max(worst_self, d)
                nonident += int(not torch.equal(a[k], b[k]))
                if not torch.equal(a[k], b[k]):
                    diff_keys.add("%s(%s)" % (k, str(a[k].dtype).replace("torch.", "")))
                for o in (a, b):
                    e = (o[k].float().cpu() - r[k]).norm().item() / max(r[k].norm().item(), 1e-9)
                    per_key[k] = max(per_key.get(k, 0.0), e)
                    if e > worst_ref:
                        worst_ref, worst_key = e, k
        hip.glue().bn_dual_mode(prev)
        print("det=%d %-20s unit-vs-staged worst max|d|/max|a| %.3e (%d tensors not bit-identical over %d reps); worst rel-L2 vs fp32 torch %.3e (%s)"
              % (det, kind, worst_self, nonident, reps, worst_ref, worst_key), flush=True)
        if det:
            print("      differing: %s" % sorted(diff_keys))
            print("      rel-L2 vs fp32 torch: " + "  ".join("%s %.3f" % (k, v) for k, v in sorted(per_key.items(), key=lambda kv: -kv[1])))
hip.set_deterministic(False)
