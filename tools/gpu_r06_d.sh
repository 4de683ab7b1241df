#!/bin/bash
# round 6, call d: triangulation micro-benchmark after the index / float32-row changes, self-supervision parity tests, the whole suite the driver's way
OUT=gpurun_out/r06d
mkdir -p $OUT
timeout 600 python tools/bench_kernels.py tri > $OUT/tri.txt 2>&1; grep -v amdgpu.ids $OUT/tri.txt
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -30) > $OUT/pytest.log
tail -3 $OUT/pytest.log
