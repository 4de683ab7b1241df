#!/bin/bash
# forced bucket path A/B (N = 1, no collective): what the N > 1 gradient path costs besides the all-reduce itself
O=gpurun_out/r03d; mkdir -p $O
(timeout 600 python -m pytest tests/test_hip_precise.py tests/test_hip_distributed.py tests/test_hip_conv.py -m gpu -q 2>&1 | tail -4) > $O/pytest.log; tail -2 $O/pytest.log
for i in 1 2 3; do
  (timeout 300 python bench.py --no-cpu-baseline --no-ss-leg 2>&1 | tail -1) > $O/plain_$i.json
  (timeout 300 python bench.py --no-cpu-baseline --no-ss-leg --force-grad-sync 2>&1 | tail -1) > $O/forced_$i.json
done
python - <<'PY' | tee gpurun_out/r03d/bucket_path_overhead.txt
import json
print("# bench.py [--force-grad-sync] --no-ss-leg, alternating runs on one MI355X; forced = the N > 1 gradient path at N = 1 (flat per-dtype buckets,")
print("# learned bucket hooks that flush the grouped weight gradients and the deferred slab sums, no collective)")
res = {}
for n in ("plain_1", "forced_1", "plain_2", "forced_2", "plain_3", "forced_3"):
    try:
        d = json.loads(open("gpurun_out/r03d/%s.json" % n).read().strip().splitlines()[-1])
        r = d["roofline"]
        res.setdefault(n.split("_")[0], []).append(d["ms_per_step"])
        print("%-10s %8.1f img/s %7.3f ms/step   conv wgrad family %.3f ms (%s launches)  gemm family %.3f ms" % (n, d["value"], d["ms_per_step"],
              r["families"]["backbone_conv_bwd_weight"]["ms_per_step"], r["families"]["backbone_conv_bwd_weight"]["launches_per_step"], r["ms_per_step"]))
    except Exception as e:
        print(n, "FAILED", e)
if res.get("plain") and res.get("forced"):
    p, f = sum(res["plain"]) / len(res["plain"]), sum(res["forced"]) / len(res["forced"])
    print("# mean: plain %.3f ms, forced %.3f ms -> overhead %.2f %%" % (p, f, 100.0 * (f / p - 1.0)))
PY
