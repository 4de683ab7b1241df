#!/bin/bash
# round 4, call p: grouped weight-gradient plan by makespan (no second round of workgroups) against the previous build
bash tools/ab_bench_families.sh r04p/ab "EPI_LIB_DIR=/root/repo/epipolarpose_amd/_lib_prev" "-" "EPI_TN_GROUP_MODEL=0" "EPI_LIB_DIR=/root/repo/epipolarpose_amd/_lib_prev" "-" "EPI_TN_GROUP_MODEL=0" "EPI_LIB_DIR=/root/repo/epipolarpose_amd/_lib_prev" "-" > gpurun_out/r04p_ab.txt 2>&1
cat gpurun_out/r04p_ab.txt
