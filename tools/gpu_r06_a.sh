#!/bin/bash
# round 6, call a: the whole GPU suite (no -x, slowest tests listed), the unit-node probe
OUT=gpurun_out/r06a
mkdir -p $OUT
(timeout 2400 python -m pytest tests -m gpu -q --durations=30 2>&1 | tail -80) > $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python tools/probe_unit_node.py 10 > $OUT/unit_probe.txt 2>&1
cat $OUT/unit_probe.txt | tail -12
cp gpurun_out/step_in_backward.json $OUT/ 2>/dev/null
