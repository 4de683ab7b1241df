#!/bin/bash
# round 4, call i: the whole GPU suite after the parity work (deterministic mode, trained-state checks, fp32-result planning fix)
mkdir -p gpurun_out/r04i
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -60) > gpurun_out/r04i/tests.txt
tail -25 gpurun_out/r04i/tests.txt
cp gpurun_out/network_trained_state.json gpurun_out/precise_parity.json gpurun_out/two_ranks_one_device_r*.json gpurun_out/r04i/ 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/network_trained_state.json"))
print(json.dumps(d["loss_rel_deterministic"]))
for k, v in d["trained_state"].items():
    print(k, {kk: round(v[kk], 5) for kk in ("min_cos", "p05_cos", "median_cos", "head_min_cos", "loss_precise", "loss_bf16")}, v["worst"][:2])
d = json.load(open("gpurun_out/precise_parity.json"))
for k, v in d.items():
    if k.startswith("bf16_vs_precise_trained"):
        print(k, {kk: round(v[kk], 5) for kk in ("min_cos", "p05_cos", "median_cos", "head_min_cos", "loss_precise", "loss_bf16")}, v["worst"][:3])
PY
bash tools/ab_bench.sh r04i/ab "EPI_LIB_DIR=/root/repo/epipolarpose_amd/_lib_base" "-"
