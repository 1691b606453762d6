#!/bin/bash
# Development: refresh the round-2 text profiles (whole-iteration graph numbers, data-parallel cost model).
mkdir -p gpurun_out
( timeout 300 python scripts/mtgs_like_train.py --graph --only fused
  timeout 300 python scripts/mtgs_like_train.py --graph --only fused --shipped
  timeout 400 python scripts/mtgs_like_train.py --graph --only fused --shipped --objects 100 ) 2>&1 | grep -v amdgpu.ids | grep "fused iteration" > gpurun_out/r02_mtgs_like_graph.txt
timeout 300 python scripts/dp_cost.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_dp_cost.txt
cut -c1-230 gpurun_out/r02_mtgs_like_graph.txt; cat gpurun_out/r02_dp_cost.txt
