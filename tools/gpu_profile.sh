#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace profile of the bench step.
# Usage: tools/gpu_profile.sh <tag> <steps> [extra bench args...]   (warm-up fixed at 3)
set -u
TAG=$1; STEPS=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o run -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --steps $STEPS --warmup 3 "$@" > "$OUT/bench.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/trace_window.py /tmp/prof_$TAG/run_kernel_trace.csv softargmax_bwd_kernel $((STEPS-1)) "$OUT/steady_state_kernels.csv"
grep -E "softargmax|joint_loss|self_supervision|head_gemm|bn_" /tmp/prof_$TAG/run_kernel_stats.csv > "$OUT/our_kernels_stats.csv"
head -1 /tmp/prof_$TAG/run_kernel_stats.csv > "$OUT/kernel_stats_head.csv"
tail -1 "$OUT/bench.log" | cut -c1-300
head -60 "$OUT/steady_state_kernels.csv" | cut -c1-200
