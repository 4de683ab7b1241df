#!/bin/bash
# the projection BatchNorm folded into the unit's last BatchNorm pass (EPI_BN_DUAL=1, default) against two passes (0): tests, then A/B
timeout 900 python -m pytest tests/test_hip_conv.py tests/test_hip_network.py tests/test_hip_head.py tests/test_hip_step_in_backward.py -q -x 2>&1 | tail -4
mkdir -p gpurun_out/r03p
for i in 1 2 3; do
  EPI_BN_DUAL=1 python bench.py --no-cpu-baseline --no-ss-leg --no-loader-leg --steps 30 --warmup 8 > gpurun_out/r03p/dual_$i.log 2>&1
  EPI_BN_DUAL=0 python bench.py --no-cpu-baseline --no-ss-leg --no-loader-leg --steps 30 --warmup 8 > gpurun_out/r03p/two_$i.log 2>&1
done
python - <<'PY'
import json, glob
for arm in ("dual", "two"):
    for f in sorted(glob.glob("gpurun_out/r03p/%s_[0-9].log" % arm)):
        for line in open(f):
            if line.startswith("{"):
                d = json.loads(line)
                fam = d["roofline"]["families"]
                print(arm, d["value"], d["ms_per_step"], "bn", d["roofline"]["batchnorm"]["ms_per_step"], d["roofline"]["batchnorm"]["launches_per_step"],
                      {k: round(v["ms_per_step"], 3) for k, v in fam.items() if k.startswith("bn_fwd")})
PY
