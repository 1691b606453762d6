#!/bin/bash
# Round 6: the LEAN packed backward against the round-5 kernel (base) and the same code at four waves per SIMD (lean4)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_lean
rm -rf $OUT && mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
bash scripts/ab2.sh "base lean4 -" > $OUT/ab2.txt 2>&1
cat $OUT/ab2.txt
timeout 600 python3 scripts/dev/blend_timeline.py mtgs_amd/_variants/libmtgs_rast_timeline.so 1920 1080 $OUT/timeline.npz > $OUT/timeline.txt 2>&1
grep -A3 "== bwd" $OUT/timeline.txt
