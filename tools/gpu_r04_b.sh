#!/bin/bash
# round 4, call b: phase trace of the NT kernel (where the per-launch fixed cost sits) + store policies
mkdir -p gpurun_out/r04b
timeout 300 tools/gemm_trace_bin > gpurun_out/r04b/trace.txt 2>&1
head -150 gpurun_out/r04b/trace.txt
