#!/bin/bash
mkdir -p gpurun_out/r04g
timeout 900 python tools/diag_bench_shape.py > gpurun_out/r04g/diag_b32.txt 2>&1
DIAG_BATCH=8 timeout 900 python tools/diag_bench_shape.py > gpurun_out/r04g/diag_b8.txt 2>&1
(timeout 900 python -m pytest tests/test_hip_deterministic.py -q -m gpu 2>&1 | tail -60) > gpurun_out/r04g/det.txt
cat gpurun_out/r04g/diag_b32.txt | tail -60
grep -n "assert\|Error\|^E " gpurun_out/r04g/det.txt | head -20
