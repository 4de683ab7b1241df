#!/bin/bash
# which hipBLASLt kernel serves torch.matmul at 8192^3 / 4096^3 bf16 (its name encodes macro tile, MFMA shape, staging scheme)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && R=$PWD
cat > /tmp/mm.py <<'PY'
import torch
for n in (8192, 4096):
    a = torch.randn(n, n, device="cuda").to(torch.bfloat16); b = torch.randn(n, n, device="cuda").to(torch.bfloat16)
    for _ in range(5): torch.matmul(a, b.t())
    torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mmprof -o mm -- python /tmp/mm.py > /tmp/mm.log 2>&1
f=$(find /tmp/mmprof -name "*kernel_stats.csv" | head -1)
head -8 "$f" > $R/gpurun_out/blaslt_kernel_names.txt
cut -c1-500 $R/gpurun_out/blaslt_kernel_names.txt; tail -5 /tmp/mm.log
