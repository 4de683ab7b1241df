#!/bin/bash
# round 4, call q: 64 x 64 weight-pack tiles with 16-byte accesses -- parity of everything that consumes packed weights, A/B, launch-by-launch trace
mkdir -p gpurun_out/r04q
(timeout 900 python -m pytest tests/test_hip_conv.py tests/test_hip_head.py tests/test_hip_step_in_backward.py -x -q -m gpu 2>&1 | tail -5) > gpurun_out/r04q/tests.txt
tail -5 gpurun_out/r04q/tests.txt
bash tools/ab_bench_families.sh r04q/ab "EPI_LIB_DIR=/root/repo/epipolarpose_amd/_lib_prev" "-" "EPI_LIB_DIR=/root/repo/epipolarpose_amd/_lib_prev" "-" "EPI_LIB_DIR=/root/repo/epipolarpose_amd/_lib_prev" "-" > gpurun_out/r04q/ab.txt 2>&1
cat gpurun_out/r04q/ab.txt
bash tools/gpu_step_sequence.sh r04q 2>&1 | tail -3
