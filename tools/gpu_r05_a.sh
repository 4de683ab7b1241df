#!/bin/bash
# round 5, call a: staggered 256x256 loop -- reference check + race screen + A/B against the lock-step loop, per-layer table, GEMM-facing GPU tests, step
O=gpurun_out/${1:-r05a}; mkdir -p $O
timeout 300 tools/gemm_lab_bin stag > $O/stag.txt 2>&1; echo "stag rc $?"; cat $O/stag.txt
timeout 300 tools/gemm_lab_bin layers 32 > $O/layers.txt 2>&1; echo "layers rc $?"; tail -4 $O/layers.txt
timeout 600 python -m pytest tests/test_hip_head.py tests/test_hip_conv.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; python - ${1:-r05a} <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/'+__import__("sys").argv[1]+'/bench.json').read().strip().splitlines()[-1])
    print('img/s', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'gemm ms', d['roofline']['ms_per_step'])
    for k, v in d['roofline']['families'].items(): print(' ', k, v.get('ms_per_step'), v.get('frac'))
except Exception as e: print('bench parse failed', e)
PY
