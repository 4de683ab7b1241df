#!/bin/bash
# round 4, call c: the branch-light epilogue -- phase trace again, kernel table, GPU tests of the GEMM paths, the bench step
mkdir -p gpurun_out/r04c
timeout 300 tools/gemm_trace_bin > gpurun_out/r04c/trace.txt 2>&1
timeout 300 tools/gemm_lab_bin gemm quick > gpurun_out/r04c/gemm.txt 2>&1
timeout 300 tools/gemm_lab_bin conv > gpurun_out/r04c/conv.txt 2>&1
(timeout 900 python -m pytest tests/test_hip_conv.py tests/test_hip_head.py tests/test_hip_network.py -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r04c/tests.txt
tail -3 gpurun_out/r04c/tests.txt
bash tools/ab_bench.sh r04c/ab "-" "-"
grep -A8 "l3.c1  1024->256 M=8192  \[default\]\|l2.c3  128->512  M=32768  \[small\|l1.c3  64->256   M=131072  \[small" gpurun_out/r04c/trace.txt
