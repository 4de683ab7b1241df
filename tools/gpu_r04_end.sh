#!/bin/bash
# round 4: the artefacts the round ends with -- GPU test suite, the bench line, steady-state rocprofv3 kernel table, launch-by-launch step
# sequence, PMC traffic of the step, the per-layer convolution table and the phase trace from the C++ tools.  Usage: tools/gpu_r04_end.sh <tag>
TAG=${1:-r04}
OUT=gpurun_out/end_$TAG
mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $OUT/pytest.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2) > $OUT/smoke.log
(timeout 400 python bench.py 2>&1 | tail -1) > $OUT/bench.json
timeout 400 bash tools/gpu_profile.sh $TAG 6 --no-ss-leg --no-loader-leg > $OUT/profile.log 2>&1
cp gpurun_out/prof_$TAG/steady_state_kernels.csv $OUT/steady_state_kernels.csv
cp gpurun_out/prof_$TAG/our_kernels_stats.csv $OUT/our_kernels_stats.csv
cp gpurun_out/prof_$TAG/kernel_stats_head.csv $OUT/kernel_stats_head.csv
timeout 300 bash tools/gpu_step_sequence.sh $TAG > $OUT/sequence.log 2>&1
cp gpurun_out/step_sequence_$TAG.txt $OUT/step_sequence.txt
timeout 900 bash tools/gpu_pmc_step.sh $TAG > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_step_$TAG/pmc_step_families.json gpurun_out/pmc_step_$TAG/pmc_step_summary.csv $OUT/ 2>/dev/null
timeout 300 tools/gemm_lab_bin layers > $OUT/conv_layers.txt 2>&1
timeout 300 tools/gemm_trace_bin > $OUT/phase_trace.txt 2>&1
cp gpurun_out/precise_parity.json gpurun_out/network_trained_state.json gpurun_out/two_ranks_one_device_r*.json $OUT/ 2>/dev/null
tail -3 $OUT/pytest.log; cut -c1-300 $OUT/bench.json; head -3 $OUT/steady_state_kernels.csv | cut -c1-200; head -2 $OUT/step_sequence.txt; tail -4 $OUT/conv_layers.txt; cat $OUT/pmc_step_families.json | cut -c1-600
