#!/bin/bash
# round 6, call c: the GPU suite the driver's way (-x), three times
OUT=gpurun_out/r06c
mkdir -p $OUT
for i in 1 2 3; do
  (timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -30) > $OUT/pytest_$i.log
  tail -1 $OUT/pytest_$i.log
done
