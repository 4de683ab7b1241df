#!/bin/bash
# round 6, call g: the projection branch on its own stream -- parity (bit-identity in deterministic mode), same-box A/B of the bench step, 3 pairs
OUT=gpurun_out/r06g
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_hip_conv.py tests/test_hip_deterministic.py tests/test_hip_network.py -x -q -m gpu -k "unit or branch or deterministic or golden" 2>&1 | tail -15) > $OUT/pytest_a.log; tail -3 $OUT/pytest_a.log
for i in 1 2 3; do
  for arm in 1 0; do
    (timeout 300 python bench.py --no-cpu-baseline --no-ss-leg --no-loader-leg --branch-stream $arm 2>$OUT/bench_${arm}_$i.err | tail -1) > $OUT/bench_${arm}_$i.json
    python - $OUT/bench_${arm}_$i.json $arm <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("branch-stream %s: %8.1f img/s  %6.3f ms  streams %s  loss %.4f" % (sys.argv[2], d["value"], d["ms_per_step"], d["config"]["streams"], d["config"]["final_loss"]))
except Exception as e:
    print("FAILED", e)
PY
  done
done
