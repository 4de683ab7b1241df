#!/bin/bash
# Run on the GPU box (through gpurun): HBM traffic of the BENCH STEP per kernel family, from the L2 fabric counters.
# FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC: 3 + 2 of 4 slots) -> two separate rocprofv3 --pmc runs of the same command
# (kernel trace only: no other trace domain is combined with the counters).  The process runs warm-up 2 + 1 + 3 diagnostic + 3 timed + 3 event-timed
# = 12 steps of the same launches.  Usage: tools/gpu_pmc_step.sh <tag>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_step_$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcs_$C -o run -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-ss-leg --no-loader-leg --steps 3 --warmup 2 > "$OUT/run_$C.log" 2>&1
  find /tmp/pmcs_$C -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} /tmp/counters_$C.csv
done
cd "$GRAFT_REPO_ROOT"
python tools/pmc_step_summary.py /tmp/counters_FETCH_SIZE.csv /tmp/counters_WRITE_SIZE.csv 12 "$OUT/pmc_step_summary.csv" "$OUT/pmc_step_families.json"
head -30 "$OUT/pmc_step_summary.csv" | cut -c1-200
cat "$OUT/pmc_step_families.json"
