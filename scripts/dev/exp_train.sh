#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/pytest_full.log 2>&1
echo rc=$?
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_full.log | head -20
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_r04_final.json; cut -c1-260 gpurun_out/bench_r04_final.json
