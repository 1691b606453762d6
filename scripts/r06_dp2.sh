#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_dp2
rm -rf $OUT && mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_dp.py -x -q -m gpu > $OUT/pytest_dp.txt 2>&1
tail -6 $OUT/pytest_dp.txt
python scripts/dp_cost.py > $OUT/dp_cost.txt 2>&1; grep "touched\|render leg\|single-GPU\|data-parallel" $OUT/dp_cost.txt
for f in "touched-chunked" "touched-chunked --dp-no-prezero" "touched"; do
  MTGS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --dp-finish $f --steps 10 --warmup 3 > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/b.json") if l.startswith("{")][-1])
    print("$f", d["n_gpus"], d["ms_per_step"], d.get("dp_phases_ms"), d.get("dp_chunk_caps_rows"), d.get("dp_overflow"))
    open("$OUT/bench_gloo2_" + "$f".replace(" ", "_") + ".json", "w").write(json.dumps(d))
except Exception as e:
    print("$f failed", e); print(open("$OUT/b.err").read()[-1500:])
PY
done
