#!/bin/bash
cd $GRAFT_REPO_ROOT
show='import json,sys
d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["ms_per_step_eager"], d["ms_per_step_graph"], d["config"]["launch"][:20])'
for i in 1 2 3; do timeout 600 python bench.py --cpu-steps 0 --no-also 2>/dev/null | tail -1 | python -c "$show"; done
