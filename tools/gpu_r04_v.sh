#!/bin/bash
# round 4, call v: split launches finished by their own last workgroups -- operator parity (every split shape of the network), determinism, A/B
mkdir -p gpurun_out/r04v
(timeout 1200 python -m pytest tests/test_hip_conv.py tests/test_hip_head.py tests/test_hip_deterministic.py tests/test_hip_network.py tests/test_hip_step_in_backward.py -x -q -m gpu -k "not trained" 2>&1 | tail -6) > gpurun_out/r04v/tests.txt
tail -6 gpurun_out/r04v/tests.txt
bash tools/gpu_ab_prev.sh r04v 3
bash tools/ab_bench_families.sh r04v/ab2 "EPI_SPLITK_FINISH=0" "-" "EPI_SPLITK_FINISH=0" "-" > gpurun_out/r04v/ab2.txt 2>&1; cat gpurun_out/r04v/ab2.txt
