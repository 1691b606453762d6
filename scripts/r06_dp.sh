#!/bin/bash
# Round 6: the N > 1 bench line (ranks sharing the box's one GPU over gloo) in the exchange forms, and their tests
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_dp
rm -rf $OUT && mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_dp.py -x -q -m gpu > $OUT/pytest_dp.txt 2>&1
tail -6 $OUT/pytest_dp.txt
for f in touched-chunked touched static; do
  MTGS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --dp-finish $f --steps 10 --warmup 3 > $OUT/bench_gloo2_$f.json 2> $OUT/bench_gloo2_$f.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/bench_gloo2_$f.json") if l.startswith("{")][-1])
    print("$f", d["n_gpus"], d["ms_per_step"], d["config"]["parallelism"][:160], d.get("dp_exchange_GBs_per_link"), d.get("dp_chunk_caps_rows"), d.get("dp_overflow"), d.get("dp_phases_ms"))
except Exception as e:
    print("$f failed", e); print(open("$OUT/bench_gloo2_$f.err").read()[-1500:])
PY
done
python scripts/dp_cost.py --no-render-leg > $OUT/dp_cost.txt 2>&1; tail -12 $OUT/dp_cost.txt
