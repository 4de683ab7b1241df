#!/bin/bash
# Run on the GPU box (through gpurun): L2 hit / miss / fabric-read counters of the convolution GEMMs (tools/pmc_conv.py).
# Usage: tools/gpu_pmc_conv_l2.sh <tag>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/pmc_l2 -o run -- python "$GRAFT_REPO_ROOT/tools/pmc_conv.py" > "$OUT/run_l2.log" 2>&1
find /tmp/pmc_l2 -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} "$OUT/counters_l2.csv"
cd "$GRAFT_REPO_ROOT"
python tools/pmc_conv_summary.py "$OUT/counters_l2.csv" "$OUT/conv_l2_summary.txt" | cut -c1-330
