#!/bin/bash
cd $GRAFT_REPO_ROOT
show='import json,sys
d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step_eager"], [ (e["entry_point"][5:], e["avg_us_live"]) for e in d["roofline"]["entry_points"] if "blend" in e["entry_point"]])'
for k in 0 128 256 512 1024; do echo "K=$k"; MTGS_SPLIT_K=$k timeout 600 python bench.py --cpu-steps 0 --no-also --launch eager 2>/dev/null | tail -1 | python -c "$show"; done
