#!/bin/bash
# rocprofv3 kernel-trace summaries of the WHOLE fused MTGS-like iteration with the shipped option set INCLUDING the optimizer
# step: torch.optim.Adam(foreach) vs mtgs_amd.optim.FusedAdam -> gpurun_out/prof_opt_{torch,fused}/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kind in torch fused; do
  rm -rf $R/gpurun_out/prof_opt_$kind && mkdir -p $R/gpurun_out/prof_opt_$kind
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_opt_$kind -o it -- python3 $R/scripts/mtgs_like_train.py --shipped --only fused --reps 10 --optimizer $kind > $R/gpurun_out/prof_opt_$kind/log.txt 2>&1
  tail -1 $R/gpurun_out/prof_opt_$kind/log.txt | cut -c1-200
done
