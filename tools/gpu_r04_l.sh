#!/bin/bash
mkdir -p gpurun_out/r04l
for i in 1 2; do
(timeout 1500 python -m pytest tests/test_hip_network.py tests/test_hip_precise.py -q -m gpu -k "golden or trained" 2>&1 | tail -5) > gpurun_out/r04l/tests$i.txt
tail -3 gpurun_out/r04l/tests$i.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/network_trained_state.json"))
for k, v in d["trained_state"].items():
    print(k, [round(v[kk], 4) for kk in ("min_cos", "p05_cos", "median_cos", "head_min_cos", "min_cos_weights", "p05_cos_weights", "median_cos_weights")], v["worst_weights"][:2])
d = json.load(open("gpurun_out/precise_parity.json"))
for k, v in d.items():
    if k.startswith("bf16_vs_precise_trained"):
        print(k, [round(v[kk], 4) for kk in ("min_cos", "p05_cos", "median_cos", "head_min_cos", "min_cos_weights", "p05_cos_weights", "median_cos_weights")], v["worst_weights"][:2])
PY
done
