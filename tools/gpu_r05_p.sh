#!/bin/bash
# L2 / TA counters of the 8192^3 GEMM: hipBLASLt's kernel (torch.matmul) and the laboratory kernels, separate rocprofv3 --pmc passes
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/pmc_gemm8k && R=$PWD && O=$R/gpurun_out/pmc_gemm8k
cat > /tmp/mm.py <<'PY'
import torch
n = 8192
a = torch.rand(n, n, device="cuda").sub_(0.5).to(torch.bfloat16); b = torch.rand(n, n, device="cuda").sub_(0.5).to(torch.bfloat16)
for _ in range(3): torch.matmul(a, b.t())
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
i=0
for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pm_a$i -o run -- python /tmp/mm.py > $O/mm_$i.log 2>&1
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pm_b$i -o run -- $R/tools/w4_lab_bin pmc > $O/lab_$i.log 2>&1
  find /tmp/pm_a$i -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/mm_counters_$i.csv
  find /tmp/pm_b$i -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/lab_counters_$i.csv
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_gemm8k/*_counters_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "Cijk" in k or "w4" in k:
            acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_gemm8k/summary.txt", "w") as out:
    for k, d in acc.items():
        line = k + "\n   " + "  ".join("%s=%.4g" % (c, sorted(v)[len(v) // 2]) for c, v in sorted(d.items()))
        print(line); out.write(line + "\n")
PY
