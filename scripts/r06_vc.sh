#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_vc
rm -rf $OUT && mkdir -p $OUT
bash scripts/ab2.sh "- vc4 vc2 vc16 vcnt vc4nt" --sh > $OUT/ab2.txt 2>&1
cat $OUT/ab2.txt | cut -c1-400
