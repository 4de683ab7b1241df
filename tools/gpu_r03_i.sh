#!/bin/bash
# the stem's max-pool inside the stem node (EPI_STEM_POOL=1, default) against the separate pool node (0): tests, then A/B
timeout 900 python -m pytest tests/test_hip_conv.py tests/test_hip_network.py tests/test_hip_precise.py -q -x 2>&1 | tail -4
mkdir -p gpurun_out/r03w
for i in 1 2 3; do
  EPI_STEM_POOL=1 python bench.py --no-cpu-baseline --no-ss-leg --no-loader-leg --steps 30 --warmup 8 > gpurun_out/r03w/pool_$i.log 2>&1
  EPI_STEM_POOL=0 python bench.py --no-cpu-baseline --no-ss-leg --no-loader-leg --steps 30 --warmup 8 > gpurun_out/r03w/sep_$i.log 2>&1
done
python - <<'PY'
import json, glob
for arm in ("pool", "sep"):
    for f in sorted(glob.glob("gpurun_out/r03w/%s_[0-9].log" % arm)):
        for line in open(f):
            if line.startswith("{"):
                d = json.loads(line)
                fam = d["roofline"]["families"]
                print(arm, d["value"], d["ms_per_step"], "bn", d["roofline"]["batchnorm"]["ms_per_step"], d["roofline"]["batchnorm"]["launches_per_step"], "pool fwd", round(fam["maxpool_fwd"]["ms_per_step"], 3))
PY
