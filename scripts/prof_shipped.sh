#!/bin/bash
# rocprofv3 kernel-trace summaries of the fused MTGS-like iteration with the shipped option set (960x540, 2M Gaussians):
# dense node path and visibility-first, without and with the fused optimizer -> gpurun_out/prof_shipped_*/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in ${TAGS:-dense visfirst visfirst_adam visfirst_adam_rowlazy}; do
  case $tag in dense) extra="";; visfirst) extra="--visfirst";; visfirst_adam) extra="--visfirst --optimizer fused";; visfirst_adam_rowlazy) extra="--visfirst --optimizer fused --row-lazy";; esac
  rm -rf $R/gpurun_out/prof_shipped_$tag && mkdir -p $R/gpurun_out/prof_shipped_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_shipped_$tag -o shipped -- python3 $R/scripts/mtgs_like_train.py --shipped --only fused --reps 10 $extra > $R/gpurun_out/prof_shipped_$tag/log.txt 2>&1
  tail -1 $R/gpurun_out/prof_shipped_$tag/log.txt | cut -c1-200
done
