#!/bin/bash
# round-3 end artefacts: tools/gpu_round_end.sh + the configs[4] per-GPU line with the refiner leg
bash tools/gpu_round_end.sh r03
(timeout 400 python bench.py --layers 152 --image 384 --no-cpu-baseline --no-ss-leg --no-loader-leg --refiner-leg --steps 10 --warmup 3 2>&1 | tail -1) > gpurun_out/end_r03/bench_cfg5_r152_384.json
cut -c1-700 gpurun_out/end_r03/bench_cfg5_r152_384.json
