#!/bin/bash
# round 4, call f: the new parity tests (deterministic mode, trained-state gradient checks at every golden configuration and at the bench shape,
# the network used twice in one graph, two ranks in deterministic mode) -- values for the floors
mkdir -p gpurun_out/r04f
(timeout 2400 python -m pytest tests/test_hip_deterministic.py tests/test_hip_network.py tests/test_hip_precise.py tests/test_hip_distributed.py -q -m gpu 2>&1 | tail -40) > gpurun_out/r04f/tests.txt
tail -40 gpurun_out/r04f/tests.txt
cp gpurun_out/network_trained_state.json gpurun_out/precise_parity.json gpurun_out/two_ranks_one_device_r*.json gpurun_out/r04f/ 2>/dev/null
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/network_trained_state.json"))
    print(json.dumps(d["loss_rel_deterministic"]))
    for k, v in d["trained_state"].items():
        print(k, {kk: v[kk] for kk in ("min_cos", "p05_cos", "median_cos", "head_min_cos", "loss_precise", "loss_bf16")}, v["worst"][:3])
except Exception as e:
    print("no network_trained_state", e)
try:
    d = json.load(open("gpurun_out/precise_parity.json"))
    for k, v in d.items():
        if k.startswith("bf16_vs_precise_trained"):
            print(k, {kk: v[kk] for kk in ("min_cos", "p05_cos", "median_cos", "head_min_cos", "loss_precise", "loss_bf16")}, v["worst"][:3])
except Exception as e:
    print("no precise_parity", e)
for l in (18, 50):
    try:
        r = json.load(open("gpurun_out/two_ranks_one_device_r%d.json" % l))[0]
        print("two ranks r%d" % l, r["min_cos"], r["median_cos"], r["worst"][:3], r["norm_ratio_range"])
    except Exception as e:
        print("no two_ranks", l, e)
PY
