#!/bin/bash
# EPI_GEMM_TRACE build of the library for tools/gemm_trace.hip: the same objects as epipolarpose_amd/_lib with head_gemm.hip recompiled with the
# phase stamps in (never shipped: tools/_trace/ is git-ignored, the product library has no stamps)
set -e
cd "$(dirname "$0")/.."
python -c "from epipolarpose_amd import build as b; b.build(verbose=False)"
mkdir -p tools/_trace
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DEPI_GEMM_TRACE -c epipolarpose_amd/csrc/head_gemm.hip -o tools/_trace/head_gemm.o
objs=$(ls epipolarpose_amd/_lib/*.o | grep -v head_gemm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_trace/libepipolar_hip.so tools/_trace/head_gemm.o $objs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/gemm_trace_bin tools/gemm_trace.hip -Ltools/_trace -lepipolar_hip -Wl,-rpath,'$ORIGIN/_trace'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/gemm_lab_bin tools/gemm_lab.hip -Lepipolarpose_amd/_lib -lepipolar_hip -Wl,-rpath,'$ORIGIN/../epipolarpose_amd/_lib'
