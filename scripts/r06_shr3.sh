#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_shr3
rm -rf $OUT && mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_sh_raster.py tests/test_gpu_sh_lazy.py tests/test_gpu_sh_prefill.py tests/test_gpu_dp.py tests/test_gpu_nodes.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -25 $OUT/pytest.txt
python scripts/dp_cost.py --worlds 2 > $OUT/dp_cost.txt 2>&1; grep "render leg\|single-GPU\|data-parallel" $OUT/dp_cost.txt
