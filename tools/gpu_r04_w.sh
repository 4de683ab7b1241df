#!/bin/bash
# round 4, call w: one weight-gradient workgroup per CU (96 KB LDS request), so that the chain's GEMMs always find an LDS slot
mkdir -p gpurun_out/r04w
bash tools/ab_bench_families.sh r04w/ab "-" "EPI_TN_ONE_PER_CU=1" "-" "EPI_TN_ONE_PER_CU=1" "-" "EPI_TN_ONE_PER_CU=1" > gpurun_out/r04w/ab.txt 2>&1; cat gpurun_out/r04w/ab.txt
bash tools/gpu_step_sequence.sh r04w EPI_TN_ONE_PER_CU=1 2>&1 | tail -2
