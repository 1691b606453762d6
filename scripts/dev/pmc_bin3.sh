#!/bin/bash
# Development: SQ counters of the bin3 kernels (scripts/fbench.py), separate passes, no trace domains.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_bin3
rm -rf $OUT && mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 scripts/fbench.py --reps 4 > $OUT/p$i.log 2>&1)
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "bin3" in n or "blend_fwd" in n:
            acc[n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:26s} {sum(v)/len(v):14.0f}  (n={len(v)})")
PY
