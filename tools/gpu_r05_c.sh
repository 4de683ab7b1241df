#!/bin/bash
# round 5, call c: SQ / TA counters of the staggered-loop variants (what is busy while the matrix pipe waits)
O=$GRAFT_REPO_ROOT/gpurun_out/r05c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_ANY"
P3="TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmc_$i -o run -- $GRAFT_REPO_ROOT/tools/stag_lab_bin pmc > $O/run_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1); cp "$f" $O/counters_$i.csv 2>/dev/null
  tail -3 $O/run_$i.log
done
python3 - <<'PY'
import csv, collections, os
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r05c'
agg = collections.OrderedDict()
for i in (1, 2, 3):
    try: rows = list(csv.DictReader(open('%s/counters_%d.csv' % (O, i))))
    except Exception as e: print('no csv', i, e); continue
    for r in rows:
        k = r['Kernel_Name'][:60]
        if 'stag' not in k: continue
        d = agg.setdefault(k, collections.OrderedDict())
        c = r['Counter_Name']; v = float(r['Counter_Value'])
        d.setdefault(c, []).append(v)
for k, d in agg.items():
    print(k)
    print('   ' + '  '.join('%s=%.3g' % (c, sum(v) / len(v)) for c, v in d.items()))
PY
