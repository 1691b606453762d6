#!/bin/bash
mkdir -p gpurun_out
(
echo "=== tests"; timeout 2400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_adam.py tests/test_gpu_nodes.py -x -q 2>&1 | tail -15
timeout 2400 python -m pytest tests/test_gpu_dp.py -x -q -k "configs3" 2>&1 | tail -15
timeout 2400 python -m pytest tests/test_gpu_mtgs_contract.py -x -q -k "regularizers" 2>&1 | tail -15
python - <<'PY'
import json
rep=json.load(open('gpurun_out/parity_report.json'))
for r in rep if isinstance(rep,list) else rep.get('report',[]):
    if r.get('kind')=='gradient' and 'outliers' in r:
        print(r['case'][:40], '|', r['name'], '| outl', r['outliers'], 'cancel', r['outliers_cancelling'], 'flip', r['outliers_flipped'], 'self', r['outliers_self_critical'], 'beyond', r.get('outliers_beyond_magnitude_bound'), 'unexpl', r['outliers_unexplained'], 'excess', r.get('worst_excess_over_bound_rel'))
PY
) > gpurun_out/exp_train.log 2>&1
