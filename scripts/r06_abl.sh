#!/bin/bash
# Round 6: where the compositing backward's time goes -- ablation builds of blend.hip (dev variants), same box, interleaved
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_abl
rm -rf $OUT && mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_sh_prefill.py -x -q -m gpu > $OUT/pytest_prefill.txt 2>&1
tail -5 $OUT/pytest_prefill.txt
bash scripts/ab2.sh "- noatomic noreduce nored_noat notrans lean" > $OUT/ab2.txt 2>&1
cat $OUT/ab2.txt | cut -c1-330
