#!/bin/bash
# launch-by-launch trace of one step on the forced bucket path
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/seq_forced -o run -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-ss-leg --force-grad-sync --steps 4 --warmup 3 > /tmp/seq_forced.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/trace_step_sequence.py /tmp/seq_forced/run_kernel_trace.csv softargmax_bwd_kernel gpurun_out/step_sequence_r03_forced.txt -5 || tail -5 /tmp/seq_forced.log
head -2 gpurun_out/step_sequence_r03_forced.txt
timeout 300 python -m pytest tests/test_hip_conv.py -m gpu -q -k "residual_unit_node" --count 1 2>&1 | tail -3
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_hip_conv.py -m gpu -q -k "bottleneck_stride2" 2>&1 | tail -1; done
