#!/bin/bash
# round 6, call f: triangulation -- parity tests of the bulk kernels, micro-benchmark of the final arrangement, SQ counters of the same launches
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06f
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_hip_selfsup.py tests/test_polynomial_triangulation.py -x -q -m gpu 2>&1 | tail -15) > $OUT/pytest_selfsup.log; tail -2 $OUT/pytest_selfsup.log
timeout 600 python tools/bench_kernels.py tri > $OUT/tri.txt 2>&1; grep -v amdgpu.ids $OUT/tri.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_tri -o run -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py tri > $OUT/pmc_run.log 2>&1
find /tmp/pmc_tri -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/tri_counters.csv
cd $GRAFT_REPO_ROOT
python tools/pmc_tri_summary.py $OUT/tri_counters.csv > $OUT/pmc_tri.txt 2>&1; cat $OUT/pmc_tri.txt
head -3 $OUT/tri_counters.csv | cut -c1-400
