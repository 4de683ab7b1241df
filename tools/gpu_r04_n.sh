#!/bin/bash
# round 4, call n: under-filled launches with fused column sums on 64 x 128 tiles (EPI_GEMM_UNDERFILL) -- operator parity with the rule on, A/B in the step
mkdir -p gpurun_out/r04n
(EPI_GEMM_UNDERFILL=2 timeout 900 python -m pytest tests/test_hip_conv.py tests/test_hip_deterministic.py -x -q -m gpu 2>&1 | tail -8) > gpurun_out/r04n/tests_rule2.txt
tail -8 gpurun_out/r04n/tests_rule2.txt
bash tools/ab_bench_families.sh r04n/ab "-" "EPI_GEMM_UNDERFILL=1" "EPI_GEMM_UNDERFILL=2" "-" "EPI_GEMM_UNDERFILL=1" "EPI_GEMM_UNDERFILL=2" | tee gpurun_out/r04n/ab.txt
