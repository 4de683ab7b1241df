#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 600 python tools/bench_gemm_vs_blaslt.py 20 > gpurun_out/gemm_vs_blaslt.txt 2>&1
cat gpurun_out/gemm_vs_blaslt.txt | grep -v amdgpu.ids
