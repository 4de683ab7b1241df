#!/bin/bash
# round 4, call s: which launches should carry the fused BatchNorm-backward reduction (by output rows)
mkdir -p gpurun_out/r04s
bash tools/ab_bench_families.sh r04s/ab "-" "EPI_BN_BWD_FUSE_MAX_ROWS=65536" "EPI_BN_BWD_FUSE_MAX_ROWS=16384" "EPI_BN_BWD_FUSE_MAX_ROWS=4096" "EPI_BN_BWD_FUSE_MIN_ROWS=16384" "-" "EPI_BN_BWD_FUSE_MAX_ROWS=65536" "EPI_BN_BWD_FUSE_MAX_ROWS=16384" "EPI_BN_BWD_FUSE_MAX_ROWS=4096" "EPI_BN_BWD_FUSE_MIN_ROWS=16384" > gpurun_out/r04s/ab.txt 2>&1
cat gpurun_out/r04s/ab.txt
