#!/bin/bash
# round 5, call b: stag_lab -- ablations and slot timing of the staggered loop
O=gpurun_out/r05b; mkdir -p $O
timeout 600 tools/stag_lab_bin > $O/stag_lab.txt 2>&1; echo "rc $?"; cat $O/stag_lab.txt
