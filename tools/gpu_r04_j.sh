#!/bin/bash
# round 4, call j: selective pipelining (EPI_GEMM_PIPE=3 default, fused-reduction launches included) against off, and against the round-3 library
B=EPI_LIB_DIR=/root/repo/epipolarpose_amd/_lib_base
bash tools/ab_bench.sh r04j/ab "$B" "-" "EPI_GEMM_PIPE=0" "-" "EPI_GEMM_PIPE=0" "$B"
(timeout 900 python -m pytest tests/test_hip_conv.py tests/test_hip_head.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/r04j/tests.txt
tail -2 gpurun_out/r04j/tests.txt
