#!/bin/bash
# round 3, call A: GPU test suite + A/B of the grouped weight-gradient launches
mkdir -p gpurun_out/r03a
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r03a/pytest.log
tail -4 gpurun_out/r03a/pytest.log
bash tools/ab_bench.sh r03a "-" "EPI_WGRAD_GROUP=0" "EPI_WGRAD_GROUP=1" "EPI_WGRAD_GROUP=2 EPI_EVENT_FENCE=0" "EPI_WGRAD_GROUP=0 EPI_EVENT_FENCE=0" 2>&1 | tee gpurun_out/r03a/ab.txt
python - <<'PY'
import json
for i in (1, 2):
    try:
        d = json.loads(open("gpurun_out/r03a/arm%d.log" % i).read().strip().splitlines()[-1])
        r = d["roofline"]
        print("arm", i, "gemm family", r["ms_per_step"], "ms", r["achieved"], "TF;", {k: (v["ms_per_step"], v.get("achieved_tflops"), v["launches_per_step"]) for k, v in r["families"].items() if "conv" in k or "head" in k})
        print("   bn", r.get("batchnorm", {}).get("ms_per_step"), "host", {k: v for k, v in d["config"].items() if "host" in k})
    except Exception as e:
        print("arm", i, "failed", e)
PY
