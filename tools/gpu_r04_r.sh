#!/bin/bash
# round 4, call r: bias gradient of the final layer from the criterion's backward kernel -- tests, A/B
mkdir -p gpurun_out/r04r
(timeout 900 python -m pytest tests/test_hip_integral.py tests/test_hip_network.py tests/test_hip_step_in_backward.py tests/test_hip_train_loop.py -x -q -m gpu -k "not trained and not golden_network" 2>&1 | tail -8) > gpurun_out/r04r/tests.txt
tail -8 gpurun_out/r04r/tests.txt
bash tools/ab_bench_families.sh r04r/ab "EPI_BIAS_GRAD_FUSE=0" "-" "EPI_BIAS_GRAD_FUSE=0" "-" "EPI_BIAS_GRAD_FUSE=0" "-" > gpurun_out/r04r/ab.txt 2>&1
cat gpurun_out/r04r/ab.txt
