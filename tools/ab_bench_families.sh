#!/bin/bash
# Run on the GPU box (through gpurun): A/B a list of environment settings on bench.py with the per-family figures of the instrumented steps beside the step time.
# Usage: tools/ab_bench_families.sh <tag> "<ENV=..> <ENV=..>" "-" ...    (one quoted environment per arm; "-" = defaults)
TAG=$1; shift
mkdir -p gpurun_out/$TAG
i=0
for arm in "$@"; do
  i=$((i+1))
  envs=$arm; [ "$arm" = "-" ] && envs=""
  (env $envs timeout 200 python bench.py --no-cpu-baseline --no-ss-leg --no-loader-leg 2>&1 | tail -1) > gpurun_out/$TAG/arm$i.log
  python - "$arm" gpurun_out/$TAG/arm$i.log <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = d["roofline"]; f = r["families"]
    ms = lambda k: f.get(k, {}).get("ms_per_step", 0.0)
    n = lambda k: int(f.get(k, {}).get("launches_per_step", 0))
    print("%-28s %8.1f img/s  %6.3f ms | gemm family %.3f ms (%d launches) conv fwd %.3f  bwd-data %.3f  bwd-weight %.3f | BatchNorm %.3f (%d): fwd apply %.3f  bwd apply %.3f (%d)  bwd reduce+apply %.3f (%d) | loss %.4f"
          % (sys.argv[1], d["value"], d["ms_per_step"], r["ms_per_step"], int(r["launches_per_step"]), ms("backbone_conv_fwd"), ms("backbone_conv_bwd_data"),
             ms("backbone_conv_bwd_weight"), r["batchnorm"]["ms_per_step"], int(r["batchnorm"]["launches_per_step"]), ms("bn_fwd_apply"), ms("bn_bwd_apply"),
             n("bn_bwd_apply"), ms("bn_bwd_reduce+apply"), n("bn_bwd_reduce+apply"), d["config"]["final_loss"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]).read()[-300:])
PY
done
