#!/bin/bash
# Round 6: the whole -m gpu suite (writes gpurun_out/parity_report.json) + the smoke entry
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_tests
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/r06_tests/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_tests/pytest_gpu.log
tail -4 gpurun_out/r06_tests/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
