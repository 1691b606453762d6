#!/bin/bash
# Counter-based HBM traffic of EVERY kernel of the benchmark step (separate --pmc passes, no trace domains), plus a
# calibration of the two counters on kernels with a known byte count (scripts/dev/read_bench.hip --calibrate).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_step
rm -rf $OUT && mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/read_bench $R/scripts/dev/read_bench.hip > $OUT/build.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/cal_$c -- $OUT/read_bench --calibrate > $OUT/cal_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-steps 0 --no-also --no-tight > $OUT/$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-steps 0 --no-also --no-tight > $OUT/sq.log 2>&1
rm -f $OUT/read_bench
find $OUT -name "*counter_collection.csv" | head -8
