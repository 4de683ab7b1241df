#!/bin/bash
# round 4, call k: BatchNorm + ReLU of the bottleneck interiors inside conv3's launch -- unit test, GPU suite of the affected files, A/B in the step
mkdir -p gpurun_out/r04k
(timeout 900 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "interior_batchnorm" 2>&1 | tail -25) > gpurun_out/r04k/unit.txt
tail -25 gpurun_out/r04k/unit.txt
(timeout 1800 python -m pytest tests/test_hip_conv.py tests/test_hip_network.py tests/test_hip_train_loop.py tests/test_hip_step_in_backward.py tests/test_hip_distributed.py tests/test_hip_deterministic.py -q -m gpu 2>&1 | tail -8) > gpurun_out/r04k/tests.txt
tail -8 gpurun_out/r04k/tests.txt
bash tools/ab_bench.sh r04k/ab "EPI_BN_IN_FUSE=0" "-" "EPI_BN_IN_FUSE=0" "-"
