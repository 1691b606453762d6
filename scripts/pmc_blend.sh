#!/bin/bash
# PMC counters of the compositing kernels inside the fused rasterization step (separate passes; no trace domains with --pmc)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_blend
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex "blend" --output-format csv -d $OUT/p1 -- python3 $R/scripts/fbench.py --reps 3 "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM --kernel-include-regex "blend" --output-format csv -d $OUT/p2 -- python3 $R/scripts/fbench.py --reps 3 "$@" > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            print(k)
            for c, v in sorted(d.items()):
                print("   %-24s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
