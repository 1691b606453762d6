#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_vc2
rm -rf $OUT && mkdir -p $OUT
bash scripts/ab2.sh "- vcold" --sh > $OUT/ab2.txt 2>&1
cat $OUT/ab2.txt | cut -c1-400
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt
