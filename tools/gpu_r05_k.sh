#!/bin/bash
# one-wave-per-SIMD 256 x 256 loop laboratory (tools/w4_lab.hip): variants, ablations, reference check and race screen; operand-data experiment
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
: > gpurun_out/w4_lab_data.txt
for mode in 0 1 2 3; do
    if [ $mode = 0 ]; then timeout 600 tools/w4_lab_bin all > gpurun_out/w4_lab.txt 2>&1; echo "exit $?" >> gpurun_out/w4_lab.txt
    else timeout 300 tools/w4_lab_bin quick $mode >> gpurun_out/w4_lab_data.txt 2>&1; fi
done
grep -E "^#|F_CLOCK" gpurun_out/w4_lab.txt | head -16
cat gpurun_out/w4_lab_data.txt
