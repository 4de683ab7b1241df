#!/bin/bash
# Run on the GPU box (through gpurun): HBM traffic of the hand-written kernels from the L2 fabric counters.
# FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC: 3 + 2 of 4 slots) -> two separate rocprofv3 --pmc runs.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o run -- python "$GRAFT_REPO_ROOT/tools/pmc_kernels.py" > "$OUT/run_$C.log" 2>&1
  find /tmp/pmc_$C -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} "$OUT/counters_$C.csv"
done
cd "$GRAFT_REPO_ROOT"
python tools/pmc_summary.py "$OUT/counters_FETCH_SIZE.csv" "$OUT/counters_WRITE_SIZE.csv" "$OUT/pmc_summary.csv"
head -40 "$OUT/pmc_summary.csv" | cut -c1-200
ls -la "$OUT"
