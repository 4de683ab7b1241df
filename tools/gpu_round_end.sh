#!/bin/bash
# Run on the GPU box (through gpurun): the artefacts a round ends with -- GPU test suite the driver's way (-x), smoke, the bench line, the configs[4] share,
# a steady-state rocprofv3 kernel table (+ rocprofv3's own --stats rows), the launch-by-launch step sequence, the per-layer convolution table, the GEMM
# battery of the head shapes, the kernel micro-benchmarks, the PMC passes of the bench step.  Usage: tools/gpu_round_end.sh <tag>
TAG=${1:-end}
OUT=gpurun_out/end_$TAG
mkdir -p $OUT
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6) > $OUT/pytest.log; tail -2 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
(timeout 400 python bench.py 2>$OUT/bench.err | tail -1) > $OUT/bench.json; cut -c1-260 $OUT/bench.json
(timeout 400 python bench.py --layers 152 --image 384 --refiner-leg --steps 10 --warmup 3 --no-cpu-baseline 2>$OUT/bench_cfg5.err | tail -1) > $OUT/bench_cfg5.json; cut -c1-200 $OUT/bench_cfg5.json
timeout 400 bash tools/gpu_profile.sh $TAG 6 --no-ss-leg --no-loader-leg > $OUT/profile.log 2>&1
cp gpurun_out/prof_$TAG/steady_state_kernels.csv gpurun_out/prof_$TAG/our_kernels_stats.csv gpurun_out/prof_$TAG/kernel_stats_head.csv $OUT/ 2>/dev/null
timeout 300 bash tools/gpu_step_sequence.sh $TAG > $OUT/sequence.log 2>&1
cp gpurun_out/step_sequence_$TAG.txt $OUT/ 2>/dev/null
timeout 300 tools/gemm_lab_bin layers 32 > $OUT/layers.txt 2>&1
timeout 300 python tools/bench_kernels.py softargmax bn > $OUT/microbench.txt 2>&1
timeout 900 bash tools/gpu_pmc_step.sh $TAG > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_step_$TAG/pmc_step_classes.csv gpurun_out/pmc_step_$TAG/pmc_step_families.json $OUT/ 2>/dev/null
head -3 $OUT/steady_state_kernels.csv | cut -c1-200; tail -3 $OUT/layers.txt; grep -v amdgpu $OUT/microbench.txt | head -12; ls $OUT
