#!/bin/bash
mkdir -p gpurun_out
(
echo "=== tests"; timeout 3000 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_nodes.py tests/test_gpu_dp.py -x -q 2>&1 | tail -4
echo "=== bench"; timeout 900 python bench.py --cpu-steps 0 --no-also 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print([(e['entry_point'], e['avg_us_live']) for e in d['roofline']['entry_points']])"
echo "=== graph bench"; timeout 600 python scripts/mtgs_like_train.py --shipped --graph --reps 96 --visfirst --optimizer fused --row-lazy --geometry-rows 2>&1 | tail -1 | cut -c1-120
) > gpurun_out/exp_train.log 2>&1
