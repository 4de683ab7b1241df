#!/bin/bash
# round 5, call g: after the hygiene pass (no env switches, no library back-ends) -- whole GPU suite, smoke, bench
O=gpurun_out/${1:-r05g}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"; tail -3 $O/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; python - ${1:-r05g} <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/' + sys.argv[1] + '/bench.json').read().strip().splitlines()[-1])
    print('img/s', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'gemm ms', d['roofline']['ms_per_step'], 'bn ms', d['roofline']['batchnorm']['ms_per_step'])
    print('cpu_baseline', {k: (v if not isinstance(v, dict) else '{...}') for k, v in d['cpu_baseline'].items()})
    print('ss', d['workload_ss']['value'], 'loader', d['workload_loader']['value'])
except Exception as e: print('bench parse failed', e)
PY
