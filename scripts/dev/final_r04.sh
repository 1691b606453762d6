#!/bin/bash
# round 4: the data the docs quote (run through gpurun; summaries are copied to profiles/ by scripts/collect_profiles.py r04)
mkdir -p gpurun_out
bash scripts/prof_r04.sh > gpurun_out/final_prof.log 2>&1
bash scripts/pmc_step.sh > gpurun_out/final_pmc.log 2>&1
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_r04_final.json
C="--shipped --visfirst --optimizer fused --row-lazy --geometry-rows --only fused --reps 1 --converge --grad-thresh 1e-3 --clear-radius 12"
(timeout 900 python scripts/mtgs_like_train.py $C --traversals 8 --steps 1200 --refine-every 100 --densify-from 500 --steady 100 500 --train-graph --one-graph --trace-refinements 2>&1 | grep -v amdgpu.ids
 echo "--- three traversals"
 timeout 900 python scripts/mtgs_like_train.py $C --steps 1000 --refine-every 100 --densify-from 250 --steady 60 260 --train-graph --one-graph 2>&1 | grep -E "refine|timing|steady|converge"
 echo "--- three traversals, eager loop"
 timeout 900 python scripts/mtgs_like_train.py $C --steps 1000 --refine-every 100 --densify-from 250 --steady 60 260 2>&1 | grep -E "timing|steady|converge"
) > gpurun_out/final_training.log 2>&1
python scripts/dev/touched.py 2>&1 | grep -v amdgpu > gpurun_out/final_touched.log
tail -3 gpurun_out/final_prof.log; tail -3 gpurun_out/final_pmc.log
