#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_shr
rm -rf $OUT && mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_sh_raster.py tests/test_gpu_sh_lazy.py tests/test_gpu_sh_prefill.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -25 $OUT/pytest.txt
python bench.py --steps 20 --warmup 5 --cpu-steps 0 > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
print({k: d.get(k) for k in ("value","ms_per_step","ms_per_step_eager","ms_per_step_graph","ms_per_step_tight_lists","ms_per_step_torch_activation","launch")})
print(d.get("entry_points_us") or d.get("also"))
PY
tail -5 $OUT/bench.err
