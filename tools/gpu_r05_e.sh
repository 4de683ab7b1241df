#!/bin/bash
# round 5, call e: staged bulk triangulation -- parity (staged vs per-item vs oracle / goldens) and the bulk microbenchmark
O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_selfsup.py tests/test_polynomial_triangulation.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.txt
timeout 600 python tools/bench_kernels.py tri > $O/tri.txt 2>&1; echo "tri rc $?"; cat $O/tri.txt
