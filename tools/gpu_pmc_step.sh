#!/bin/bash
# Run on the GPU box (through gpurun): the BENCH STEP per kernel class -- HBM traffic (FETCH_SIZE, WRITE_SIZE: two passes, TCC slots) and the matrix pipe's
# occupancy (one SQ + GRBM pass).  Kernel trace only: no other trace domain is combined with the counters.  The process runs warm-up 2 + 1 + 3 diagnostic +
# 3 timed + 3 event-timed = 12 steps of the same launches.  Usage: tools/gpu_pmc_step.sh <tag>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_step_$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcs_$i -o run -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-ss-leg --no-loader-leg --steps 3 --warmup 2 > "$OUT/run_$i.log" 2>&1
  find /tmp/pmcs_$i -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} /tmp/counters_$i.csv
  tail -1 "$OUT/run_$i.log" | cut -c1-200
done
cd "$GRAFT_REPO_ROOT"
python tools/pmc_step_summary_r05.py /tmp/counters_1.csv /tmp/counters_2.csv /tmp/counters_3.csv 12 "$OUT/pmc_step_classes.csv" "$OUT/pmc_step_families.json"
head -50 "$OUT/pmc_step_classes.csv" | cut -c1-260
