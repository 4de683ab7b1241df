#!/bin/bash
# row-pitch experiment of the one-wave-per-SIMD laboratory kernels
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 300 tools/w4_lab_bin quick 0 pitch > gpurun_out/w4_lab_pitch.txt 2>&1
cat gpurun_out/w4_lab_pitch.txt
