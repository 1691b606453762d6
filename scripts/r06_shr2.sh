#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_shr2
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-also --no-tight > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:22]:
    print("%-90s %6s %10.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3))
PY
cp $f $OUT/kernel_stats.csv
