#!/bin/bash
# the projection's input gradient at half resolution (EPI_HALF_SHORTCUT=1, default) against the zero-expanded four-phase launch (0): tests, then A/B
timeout 900 python -m pytest tests/test_hip_conv.py tests/test_hip_network.py -q -x 2>&1 | tail -4
mkdir -p gpurun_out/r03q
for i in 1 2 3; do
  EPI_HALF_SHORTCUT=1 python bench.py --no-cpu-baseline --no-ss-leg --no-loader-leg --steps 30 --warmup 8 > gpurun_out/r03q/half_$i.log 2>&1
  EPI_HALF_SHORTCUT=0 python bench.py --no-cpu-baseline --no-ss-leg --no-loader-leg --steps 30 --warmup 8 > gpurun_out/r03q/full_$i.log 2>&1
done
python - <<'PY'
import json, glob
for arm in ("half", "full"):
    for f in sorted(glob.glob("gpurun_out/r03q/%s_[0-9].log" % arm)):
        for line in open(f):
            if line.startswith("{"):
                d = json.loads(line)
                fam = d["roofline"]["families"]
                print(arm, d["value"], d["ms_per_step"], "dgrad", round(fam["backbone_conv_bwd_data"]["ms_per_step"], 3), fam["backbone_conv_bwd_data"]["launches_per_step"])
PY
