#!/bin/bash
# round 4: the data the docs quote (run through gpurun; summaries are copied to profiles/ by scripts/collect_profiles.py r04)
mkdir -p gpurun_out
(
echo "=== full GPU suite"; timeout 3400 python -m pytest tests -q -m gpu 2>&1 | tail -6
) > gpurun_out/final_tests.log 2>&1
cp gpurun_out/parity_report.json gpurun_out/parity_report_full.json 2>/dev/null
