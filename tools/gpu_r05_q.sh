#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 300 tools/w4_lab_bin swz > gpurun_out/w4_lab_swz.txt 2>&1
cat gpurun_out/w4_lab_swz.txt
