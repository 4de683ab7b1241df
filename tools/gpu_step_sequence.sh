#!/bin/bash
# (bench.py: 3 warm-up + 1 diagnostic + 4 timed + 4 event-instrumented steps = 12 marker launches; -5 = the last timed step)
# Run on the GPU box (through gpurun): launch-by-launch trace of one steady-state step.  Usage: tools/gpu_step_sequence.sh <tag> [ENV=.. ...]
TAG=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/seq_$TAG -o run -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-ss-leg --no-loader-leg --workload fs --steps 4 --warmup 3 > /tmp/seq_$TAG.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/trace_step_sequence.py /tmp/seq_$TAG/run_kernel_trace.csv softargmax_bwd_kernel gpurun_out/step_sequence_$TAG.txt -5 || tail -5 /tmp/seq_$TAG.log
head -3 gpurun_out/step_sequence_$TAG.txt
