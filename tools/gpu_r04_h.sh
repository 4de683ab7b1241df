#!/bin/bash
mkdir -p gpurun_out/r04h
timeout 900 python tools/diag_precise_ops.py > gpurun_out/r04h/ops.txt 2>&1
cat gpurun_out/r04h/ops.txt | tail -40
