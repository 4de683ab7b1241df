#!/bin/bash
# round 3, call: full GPU suite, bench A/B of the forced bucket path, BASELINE config 5 per-GPU workload
O=gpurun_out/r03c; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/pytest.log; tail -3 $O/pytest.log
for i in 1 2; do
  (timeout 300 python bench.py --no-cpu-baseline --no-ss-leg 2>&1 | tail -1) > $O/plain_$i.json
  (timeout 300 python bench.py --no-cpu-baseline --no-ss-leg --force-grad-sync 2>&1 | tail -1) > $O/forced_$i.json
done
(timeout 600 python bench.py --no-cpu-baseline --no-ss-leg --layers 152 --image 384 --steps 10 --warmup 3 2>&1 | tail -1) > $O/cfg5_r152_384.json
python - <<'PY'
import json
for n in ("plain_1", "forced_1", "plain_2", "forced_2", "cfg5_r152_384"):
    try:
        d = json.loads(open("gpurun_out/r03c/%s.json" % n).read().strip().splitlines()[-1])
        r = d["roofline"]
        print("%-14s %8.1f img/s %7.3f ms  gemm %.3f ms %.1f TF frac %.3f  bn %.3f ms  host burst %.2f" % (n, d["value"], d["ms_per_step"], r["ms_per_step"], r["achieved"], r["frac"],
              r.get("batchnorm", {}).get("ms_per_step", 0), d["config"]["host_enqueue_ms_per_step_burst3"]))
    except Exception as e:
        print(n, "FAILED", e)
PY
