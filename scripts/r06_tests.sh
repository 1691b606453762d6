#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_tests
rm -rf $OUT && mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -8 $OUT/pytest.txt
python scripts/dp_cost.py --no-render-leg > $OUT/dp_cost.txt 2>&1; grep "touched" $OUT/dp_cost.txt
