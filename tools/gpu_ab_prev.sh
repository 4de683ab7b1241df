#!/bin/bash
# Run on the GPU box (through gpurun): the build in epipolarpose_amd/_lib against the one kept in epipolarpose_amd/_lib_prev, interleaved arms.
# Usage: tools/gpu_ab_prev.sh <tag> [pairs]
TAG=$1; N=${2:-3}
mkdir -p gpurun_out/$TAG
ARMS=()
for i in $(seq $N); do ARMS+=("EPI_LIB_DIR=/root/repo/epipolarpose_amd/_lib_prev" "-"); done
bash tools/ab_bench_families.sh $TAG/ab "${ARMS[@]}" > gpurun_out/$TAG/ab.txt 2>&1
cat gpurun_out/$TAG/ab.txt
