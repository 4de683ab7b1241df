#!/bin/bash
# round 4, call e: same-box A/B of the round-3 library (_lib_base) against the branch-light epilogue build (no spills now)
B=EPI_LIB_DIR=/root/repo/epipolarpose_amd/_lib_base
bash tools/ab_bench.sh r04e/ab "$B" "-" "$B" "-" "EPI_GEMM_STORES=nt" "EPI_GEMM_PIPE=1" "-"
(timeout 900 python -m pytest tests/test_hip_conv.py tests/test_hip_head.py -x -q -m gpu 2>&1 | tail -5) > gpurun_out/r04e/tests.txt
tail -3 gpurun_out/r04e/tests.txt
timeout 300 tools/gemm_lab_bin gemm quick > gpurun_out/r04e/gemm.txt 2>&1
