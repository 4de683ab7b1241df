#!/bin/bash
# Run on the GPU box (through gpurun): A/B a list of environment settings on bench.py.
# Usage: tools/ab_bench.sh <tag> "<ENV=..> <ENV=..>" "<ENV=..>" ...    (one quoted environment per arm; "-" = defaults)
TAG=$1; shift
mkdir -p gpurun_out/$TAG
i=0
for arm in "$@"; do
  i=$((i+1))
  envs=$arm; [ "$arm" = "-" ] && envs=""
  (env $envs timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/$TAG/arm$i.log
  python - "$arm" gpurun_out/$TAG/arm$i.log <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-44s %8.1f img/s  %6.3f ms  host %5.2f  ss %6.3f ms  loss %.4f" % (sys.argv[1], d["value"], d["ms_per_step"],
          d["config"]["host_enqueue_ms_one_step_empty_queue"], d["workload_ss"]["ms_per_step"], d["config"]["final_loss"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]).read()[-300:])
PY
done
