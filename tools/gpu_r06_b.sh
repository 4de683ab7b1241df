#!/bin/bash
# round 6, call b: the GPU suite the driver's way (-x), the unit-node probe, baseline bench line and per-layer table of the round-5 build
OUT=gpurun_out/r06b
mkdir -p $OUT
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -30) > $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 600 python tools/probe_unit_node.py 5 > $OUT/unit_probe.txt 2>&1
grep -v amdgpu.ids $OUT/unit_probe.txt | cut -c1-600
(timeout 400 python bench.py 2>$OUT/bench.err | tail -1) > $OUT/bench.json; cut -c1-400 $OUT/bench.json
timeout 300 tools/gemm_lab_bin layers 32 > $OUT/layers.txt 2>&1; tail -30 $OUT/layers.txt
