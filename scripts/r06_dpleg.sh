#!/bin/bash
# kernel trace of the data-parallel render leg (scripts/dp_cost.py, render leg only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_dpleg
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/scripts/dp_cost.py --worlds > $OUT/log.txt 2>&1
tail -5 $OUT/log.txt
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:30]:
    print("%-100s %6s %10.1f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3))
PY
