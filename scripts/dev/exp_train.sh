#!/bin/bash
mkdir -p gpurun_out
(
echo "=== tests"; timeout 3000 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q 2>&1 | tail -8
echo "=== bench"; timeout 900 python bench.py --cpu-steps 0 --no-also 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print([(e['entry_point'], e['avg_us_live']) for e in d['roofline']['entry_points']])"
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 bench.py --steps 4 --warmup 2 --cpu-steps 0 --no-also > $OUT/log.txt 2>&1
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_sq/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'blend' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k, {c: round(sum(x)/len(x)/1e6,2) for c,x in v.items()}, 'launches', len(next(iter(v.values()))))
PY
) > gpurun_out/exp_train.log 2>&1
