#!/bin/bash
# round 6, call e: persistent / pipelined staged triangulation kernel: parity tests, micro-benchmark
OUT=gpurun_out/r06e
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_hip_selfsup.py tests/test_polynomial_triangulation.py -x -q -m gpu 2>&1 | tail -15) > $OUT/pytest_selfsup.log; tail -3 $OUT/pytest_selfsup.log
timeout 600 python tools/bench_kernels.py tri > $OUT/tri.txt 2>&1; grep -v amdgpu.ids $OUT/tri.txt
