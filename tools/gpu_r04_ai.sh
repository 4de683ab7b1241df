#!/bin/bash
# round 4, call ai: three-stage weight-gradient kernels (EPI_TN_PIPE=1) -- parity with the switch on, per-layer table, A/B in the step
mkdir -p gpurun_out/r04ai
(EPI_TN_PIPE=1 timeout 900 python -m pytest tests/test_hip_head.py tests/test_hip_conv.py -q -m gpu -k "weight or tn or grouped or deferred" 2>&1 | tail -3) > gpurun_out/r04ai/tests.txt; cat gpurun_out/r04ai/tests.txt
tools/gemm_lab_bin layers 2>&1 | grep "total ours" ; EPI_TN_PIPE=1 tools/gemm_lab_bin layers 2>&1 | tee gpurun_out/r04ai/layers_pipe.txt | grep "total ours"
bash tools/ab_bench_families.sh r04ai/ab "-" "EPI_TN_PIPE=1" "-" "EPI_TN_PIPE=1" "-" "EPI_TN_PIPE=1" > gpurun_out/r04ai/ab.txt 2>&1; cut -c1-200 gpurun_out/r04ai/ab.txt
