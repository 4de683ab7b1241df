#!/bin/bash
# the matrix pipes alone: MFMA shape x operand fill
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
: > gpurun_out/w4_lab_power.txt
for mode in 0 3 1 2; do echo "# operand fill mode $mode" >> gpurun_out/w4_lab_power.txt; timeout 120 tools/w4_lab_bin power $mode >> gpurun_out/w4_lab_power.txt 2>&1; done
cat gpurun_out/w4_lab_power.txt
