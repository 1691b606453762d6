#!/bin/bash
# Round 6: the data the docs quote, part 1 (run through gpurun; then `python scripts/collect_profiles.py r06` in the container)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/prof_round.sh > gpurun_out/final_prof.log 2>&1
bash scripts/pmc_step.sh > gpurun_out/final_pmc.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_final.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_final.log
tail -3 gpurun_out/final_prof.log; tail -3 gpurun_out/final_pmc.log; tail -4 gpurun_out/pytest_gpu_final.log
