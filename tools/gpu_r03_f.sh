#!/bin/bash
# A/B of the fused BatchNorm-backward reduction on the bench step (interleaved arms), then the GPU suite
mkdir -p gpurun_out/r03m
for i in 1 2 3; do
  EPI_BN_BWD_FUSE=1 python bench.py --no-cpu-baseline --no-ss-leg --no-loader-leg --steps 30 --warmup 8 > gpurun_out/r03m/fuse_$i.log 2>&1
  EPI_BN_BWD_FUSE=0 python bench.py --no-cpu-baseline --no-ss-leg --no-loader-leg --steps 30 --warmup 8 > gpurun_out/r03m/plain_$i.log 2>&1
done
python - <<'PY'
import json, glob
for arm in ("fuse", "plain"):
    for f in sorted(glob.glob("gpurun_out/r03m/%s_[0-9].log" % arm)):
        for line in open(f):
            if line.startswith("{"):
                d = json.loads(line)
                fam = d["roofline"]["families"] if "families" in d["roofline"] else {}
                print(arm, d["value"], d["ms_per_step"], "bn", d["roofline"]["batchnorm"]["ms_per_step"],
                      "conv", d["roofline"]["ms_per_step"], {k: round(v["ms_per_step"], 3) for k, v in fam.items() if k.startswith("bn_bwd") or "bwd_data" in k or k == "head_gemm_bf16"})
PY
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
