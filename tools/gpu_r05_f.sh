#!/bin/bash
# round 5, call f: the whole GPU suite (new: stock-bf16 yardstick per tensor class, staged triangulation, per-device deterministic scratch)
O=gpurun_out/r05f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -25 $O/pytest.txt
cp gpurun_out/network_trained_state.json $O/ 2>/dev/null
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/network_trained_state.json'))
    for name, rep in d['trained_state'].items():
        print(name, 'loss precise %.6f bf16 %.6f stock %.6f' % (rep['loss_precise'], rep['loss_bf16'], rep.get('loss_stock_bf16', float('nan'))))
        for c, v in sorted(rep.get('by_class', {}).items()):
            print('   %-7s n %3d  cos ours %.4f stock %.4f | norm ours %.3f stock %.3f | median ours %.4f stock %.4f | min ours %.3f stock %.3f' % (
                c, v['n'], v['cos_ours'], v['cos_stock'], v['norm_ratio_ours'], v['norm_ratio_stock'], v['median_cos_ours'], v['median_cos_stock'], v['min_cos_ours'], v['min_cos_stock']))
except Exception as e: print('no report', e)
PY
