#!/bin/bash
# Round 6: the LEAN packed backward against the round-5 kernel (base) and the same code at four waves per SIMD (lean4)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_lean
rm -rf $OUT && mkdir -p $OUT
bash scripts/ab2.sh "base lean4 -" > $OUT/ab2.txt 2>&1
cat $OUT/ab2.txt
for v in timeline timeline4; do
timeout 600 python3 scripts/dev/blend_timeline.py mtgs_amd/_variants/libmtgs_rast_$v.so 1920 1080 $OUT/$v.npz > $OUT/$v.txt 2>&1
grep -A3 "== bwd" $OUT/$v.txt | cut -c1-300
done
timeout 900 python -m pytest tests/test_gpu_dp.py -x -q -m gpu -k "traversal or rccl or single_process" > $OUT/pytest_dp.txt 2>&1
tail -15 $OUT/pytest_dp.txt
