#!/bin/bash
# grouped tile order of the 256 x 256 configuration: GEMM tests, then the stag battery with this build and with the build in _lib_base (strip order), interleaved
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_hip_head.py tests/test_hip_conv.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2 3; do
    LD_LIBRARY_PATH=$PWD/epipolarpose_amd/_lib_base timeout 300 tools/gemm_lab_bin stag > $O/stag_base_$rep.txt 2>&1
    timeout 300 tools/gemm_lab_bin stag > $O/stag_new_$rep.txt 2>&1
    echo "== rep $rep base"; grep -E "8192\^3|4096\^3|2048\^3|dc1.dX" $O/stag_base_$rep.txt | grep -E "tile 2|tile 0"
    echo "== rep $rep new";  grep -E "8192\^3|4096\^3|2048\^3|dc1.dX" $O/stag_new_$rep.txt | grep -E "tile 2|tile 0"
done
grep -c "bad 0" $O/stag_new_1.txt; grep -v "bad 0" $O/stag_new_1.txt | head
