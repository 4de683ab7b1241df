#!/bin/bash
# round 4, call a: where does the time of the small GEMM launches go?  (tools/gemm_lab.hip) + the K-rotated kernels through the GPU tests
mkdir -p gpurun_out/r04a
timeout 300 tools/gemm_lab_bin probe > gpurun_out/r04a/probe.txt 2>&1
timeout 300 tools/gemm_lab_bin gemm > gpurun_out/r04a/gemm.txt 2>&1
timeout 300 tools/gemm_lab_bin conv > gpurun_out/r04a/conv.txt 2>&1
EPI_GEMM_KROT=1 timeout 900 python -m pytest tests/test_hip_conv.py tests/test_hip_head.py -x -q -m gpu > gpurun_out/r04a/tests_krot1.txt 2>&1
tail -3 gpurun_out/r04a/tests_krot1.txt
head -60 gpurun_out/r04a/probe.txt
