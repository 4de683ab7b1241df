#!/bin/bash
# round 5, call h: same-box A/B of the current library against epipolarpose_amd/_lib_base (the previous build), interleaved arms, per-family figures;
# then the BatchNorm tests.  Usage: tools/gpu_r05_h.sh <tag>
T=${1:-r05h}
B="EPI_LIB_DIR=$GRAFT_REPO_ROOT/epipolarpose_amd/_lib_base"
bash tools/ab_bench_families.sh $T/ab "$B" "-" "$B" "-" "$B" "-" > gpurun_out/$T.ab.txt 2>&1; cat gpurun_out/$T.ab.txt | cut -c1-330
timeout 900 python -m pytest tests/test_hip_head.py tests/test_hip_conv.py tests/test_hip_deterministic.py -m gpu -q -x 2>&1 | tail -3
