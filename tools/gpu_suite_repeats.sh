#!/bin/bash
# Run on the GPU box (through gpurun): the driver's GPU command -- python -m pytest tests -x -q -m gpu -- N times back to back on the same build;
# the last line of every run goes to gpurun_out/suite_repeats.txt.  Usage: tools/gpu_suite_repeats.sh [N]
N=${1:-20}
OUT=gpurun_out/suite_repeats.txt
mkdir -p gpurun_out
echo "# $(date -u +%Y-%m-%dT%H:%M:%SZ)  python -m pytest tests -x -q -m gpu  x $N, one gpurun call, build $(python -c 'from epipolarpose_amd import hip; print(hip.load().epi_version().decode())' 2>/dev/null)" > $OUT
for i in $(seq 1 $N); do
  timeout 1500 python -m pytest tests -x -q -m gpu > /tmp/suite_$i.log 2>&1
  rc=$?
  echo "run $i: rc $rc: $(tail -1 /tmp/suite_$i.log)" >> $OUT
  if [ $rc -ne 0 ]; then tail -60 /tmp/suite_$i.log > gpurun_out/suite_repeats_failure_$i.log; fi
done
cat $OUT
