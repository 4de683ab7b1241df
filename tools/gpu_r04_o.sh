#!/bin/bash
# round 4, call o: runtime knobs that touch the kernel boundary
bash tools/ab_bench_families.sh r04o/ab "-" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "GPU_MAX_HW_QUEUES=2" "-" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "GPU_MAX_HW_QUEUES=2" | tee gpurun_out/r04o/ab.txt
