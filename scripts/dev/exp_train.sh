#!/bin/bash
cd $GRAFT_REPO_ROOT
MTGS_TORCH_PROFILE=1 timeout 900 python scripts/mtgs_like_train.py --shipped --visfirst --optimizer fused --row-lazy --geometry-rows --only fused --reps 2 --graph --objects 100 --traversals 8 2>&1 | grep -E "^aten::" | cut -c1-400 | head -30
