#!/bin/bash
# Run on the GPU box (through gpurun): why does the N>1 gradient-bucket path cost ~0.85 ms/step on one GPU?
# Kernel-trace profiles of the plain step and of the same step with the bucket path forced on, windowed to the steady
# state, plus the phase timings; diff the two steady_state_kernels.csv files (copy kernels? extra elementwise launches?).
#   gpurun --timeout 900 -- 'bash tools/gpu_bucket_path_profile.sh'
set -u
cd "$GRAFT_REPO_ROOT"
bash tools/gpu_profile.sh plain 6 > /dev/null 2>&1
bash tools/gpu_profile.sh buckets 6 --force-grad-sync > /dev/null 2>&1
python - <<'PY'
import csv, os
root = os.environ["GRAFT_REPO_ROOT"]
def load(tag):
    rows = list(csv.reader(open(os.path.join(root, "gpurun_out", "prof_" + tag, "steady_state_kernels.csv"))))
    print(tag, rows[0])
    return {r[0]: (float(r[1]), float(r[2])) for r in rows[2:]}
a, b = load("plain"), load("buckets")
diff = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0.0, 0.0))
    cb, tb = b.get(k, (0.0, 0.0))
    if abs(tb - ta) > 5.0 or abs(cb - ca) > 0.5:
        diff.append((tb - ta, cb - ca, k[:110]))
for d in sorted(diff, reverse=True)[:25]:
    print("%+9.1f us/step  %+6.1f launches/step  %s" % d)
PY
python tools/phase_times.py 2>&1 | tail -2
for mode in foreach cat; do
  echo "EPI_BUCKET_PACK=$mode"; EPI_BUCKET_PACK=$mode python bench.py --no-cpu-baseline --force-grad-sync 2>/dev/null | tail -1 | cut -c1-250
done
