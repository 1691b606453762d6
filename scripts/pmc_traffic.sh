#!/bin/bash
# HBM traffic of the compositing kernels from PMC counters (separate --pmc passes, no trace domains).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_traffic
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "blend" --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-steps 0 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "blend" --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-steps 0 > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "blend" --output-format csv -d $OUT/req -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-steps 0 > $OUT/req.log 2>&1
find $OUT -name "*counter_collection.csv" | head
tail -2 $OUT/req.log | cut -c1-200
