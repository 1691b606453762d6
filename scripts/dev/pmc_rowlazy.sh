#!/bin/bash
# Development: VALU counters of the row-lazy optimizer kernels inside the MTGS-like iteration -> gpurun_out/pmc_rowlazy/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_rowlazy
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT -- python3 $R/scripts/mtgs_like_train.py --shipped --only fused --reps 6 --visfirst --optimizer fused --row-lazy > $OUT/log.txt 2>&1
python3 - <<EOF
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "adam" in r["Kernel_Name"] or "vis_color" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    busy = m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (m["GRBM_GUI_ACTIVE"] / 8)
    print(k, len(next(iter(d.values()))), "VALU insts %.1f M" % (m["SQ_INSTS_VALU"] / 1e6), "waves %d" % m["SQ_WAVES"],
          "GUI cycles/XCD %d" % (m["GRBM_GUI_ACTIVE"] / 8), "VALU busy %.2f" % busy)
EOF
