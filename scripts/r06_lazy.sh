#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_lazy
rm -rf $OUT && mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_sh_lazy.py tests/test_gpu_sh_prefill.py tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_graphs.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -12 $OUT/pytest.txt
timeout 900 python bench.py --cpu-steps 0 --no-also > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; cut -c1-500 $OUT/bench.json; tail -3 $OUT/bench.err
