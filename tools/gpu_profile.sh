#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace profile of the bench step.  Usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o run -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
ls -la /tmp/prof_$TAG/* | head -20
tail -2 "$OUT/bench.log" | cut -c1-400
head -45 "$OUT/kernel_stats.csv" | cut -c1-220
