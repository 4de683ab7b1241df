#!/bin/bash
# Run on the GPU box (through gpurun): the artefacts a round ends with -- GPU test suite, the bench line, a steady-state rocprofv3
# kernel table, the launch-by-launch step sequence and the per-layer convolution table.  Usage: tools/gpu_round_end.sh <tag>
TAG=$1
OUT=gpurun_out/end_$TAG
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $OUT/pytest.log
(timeout 300 python bench.py 2>&1 | tail -1) > $OUT/bench.json
timeout 300 bash tools/gpu_profile.sh $TAG 6 --no-ss-leg > $OUT/profile.log 2>&1
cp gpurun_out/prof_$TAG/steady_state_kernels.csv $OUT/steady_state_kernels.csv
timeout 200 bash tools/gpu_step_sequence.sh $TAG > $OUT/sequence.log 2>&1
(timeout 400 python tools/bench_conv.py > $OUT/conv_layers.txt) 2> $OUT/conv_layers.err
tail -3 $OUT/pytest.log; cut -c1-200 $OUT/bench.json; head -2 $OUT/steady_state_kernels.csv; tail -3 $OUT/conv_layers.txt
