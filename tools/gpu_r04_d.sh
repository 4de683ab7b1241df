#!/bin/bash
# round 4, call d: same-box A/B of the round-3 library (_lib_base) against the current one
bash tools/ab_bench.sh r04d/ab "EPI_LIB_DIR=/root/repo/epipolarpose_amd/_lib_base" "-" "EPI_LIB_DIR=/root/repo/epipolarpose_amd/_lib_base" "-" "EPI_GEMM_PIPE=1"
