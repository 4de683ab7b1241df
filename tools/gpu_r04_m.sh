#!/bin/bash
mkdir -p gpurun_out/r04m
for i in 1 2; do
(timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -5) > gpurun_out/r04m/tests$i.txt
tail -3 gpurun_out/r04m/tests$i.txt
done
cp gpurun_out/precise_parity.json gpurun_out/network_trained_state.json gpurun_out/two_ranks_one_device_r*.json gpurun_out/r04m/
