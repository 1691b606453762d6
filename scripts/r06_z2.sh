#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_z2
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/scripts/fbench.py --sh > $OUT/log.txt 2>&1
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:22]:
    print("%-80s %6s %10.1f" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e3))
PY
