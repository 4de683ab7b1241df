#!/bin/bash
O=gpurun_out/r03g; mkdir -p $O
(timeout 600 python -m pytest tests/test_hip_distributed.py tests/test_hip_step_in_backward.py tests/test_hip_conv.py -m gpu -q 2>&1 | tail -4) > $O/pytest.log; tail -2 $O/pytest.log
for i in 1 2 3; do
  (timeout 300 python bench.py --no-cpu-baseline --no-ss-leg 2>&1 | tail -1) > $O/plain_$i.json
  (timeout 300 python bench.py --no-cpu-baseline --no-ss-leg --force-grad-sync 2>&1 | tail -1) > $O/forced_$i.json
  (EPI_BUCKET_PIPELINE=0 timeout 300 python bench.py --no-cpu-baseline --no-ss-leg --force-grad-sync 2>&1 | tail -1) > $O/forcedjoin_$i.json
done
python - <<'PY' | tee gpurun_out/r03g/bucket_path_overhead.txt
import json
print("# bench.py [--force-grad-sync] --no-ss-leg, alternating runs on one MI355X; forced = the N > 1 gradient path at N = 1 (flat per-dtype buckets,")
print("# learned bucket hooks, no collective); forcedjoin = EPI_BUCKET_PIPELINE=0: every hook joins the second stream before it packs its bucket")
res = {}
for i in (1, 2, 3):
    for n in ("plain", "forced", "forcedjoin"):
        try:
            d = json.loads(open("gpurun_out/r03g/%s_%d.json" % (n, i)).read().strip().splitlines()[-1])
            r = d["roofline"]
            res.setdefault(n, []).append(d["ms_per_step"])
            print("%-12s %8.1f img/s %7.3f ms/step   conv wgrad family %.3f ms (%s launches)" % (n + "_%d" % i, d["value"], d["ms_per_step"],
                  r["families"]["backbone_conv_bwd_weight"]["ms_per_step"], r["families"]["backbone_conv_bwd_weight"]["launches_per_step"]))
        except Exception as e:
            print(n, i, "FAILED", e)
m = {k: sum(v) / len(v) for k, v in res.items()}
if len(m) == 3:
    print("# mean: plain %.3f ms, forced %.3f ms (+%.2f %%), forced with a join per bucket %.3f ms (+%.2f %%)" % (m["plain"], m["forced"], 100 * (m["forced"] / m["plain"] - 1),
          m["forcedjoin"], 100 * (m["forcedjoin"] / m["plain"] - 1)))
PY
