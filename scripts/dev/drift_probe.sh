#!/bin/bash
# Development: how far do two runs of ONE mtgs_like_train.py command (the single-process --accumulate 2 job of
# tests/test_gpu_mtgs_contract.py::test_mtgs_like_training_dp_rows_all_the_way_equals_accumulation) drift apart by themselves, with and
# without the zeroed-outputs form of the projection backward?  Prints the refinement sizes of every run.
cd $GRAFT_REPO_ROOT
COMMON="--accumulate 2 --n-background 60000 --n-road 20000 --traversals 3 --width 320 --height 200 --steps 400 --refine-every 50 --densify-from 120 --reps 1 --only fused --shipped --converge --grad-thresh 1e-3 --clear-radius 12"
for z in 1 0; do for i in 1 2 3 4; do
  echo -n "MTGS_ZEROED_OUTPUTS=$z run $i: "
  MTGS_ZEROED_OUTPUTS=$z python scripts/mtgs_like_train.py $COMMON 2>/dev/null | grep -o "N \[.*\]"
done; done
