#!/bin/bash
# rocprofv3 kernel-trace summaries of the neighbour rows (node activations, masked SSIM) and of the DP kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in node_bench loss_bench dp_cost; do
  rm -rf $R/gpurun_out/prof_$s && mkdir -p $R/gpurun_out/prof_$s
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$s -o $s -- python3 $R/scripts/$s.py > $R/gpurun_out/prof_$s/log.txt 2>&1
  tail -3 $R/gpurun_out/prof_$s/log.txt | cut -c1-200
done
