#!/bin/bash
# round 5, call j: per-layer table and the GEMM battery of the current build
O=gpurun_out/r05j; mkdir -p $O
timeout 300 tools/gemm_lab_bin layers 32 > $O/layers.txt 2>&1; echo "layers rc $?"; cat $O/layers.txt
timeout 300 tools/gemm_lab_bin stag > $O/stag.txt 2>&1; echo "stag rc $?"; grep "fin.fwd\|131072x64x256\|8192^3\|4096^3" $O/stag.txt
