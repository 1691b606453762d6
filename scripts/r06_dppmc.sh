#!/bin/bash
# Counters of the receiver's reduction (dp_reduce_kernel) at eight senders: bytes fetched / written and the SQ view
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_dppmc
rm -rf $OUT && mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $R/scripts/dp_cost.py --worlds 8 --pmc --no-render-leg > $OUT/$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/sq -- python3 $R/scripts/dp_cost.py --worlds 8 --pmc --no-render-leg > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $OUT/lds -- python3 $R/scripts/dp_cost.py --worlds 8 --pmc --no-render-leg > $OUT/lds.log 2>&1
tail -2 $OUT/sq.log
python3 - <<PY
import csv, glob, collections
for d in ("FETCH_SIZE", "WRITE_SIZE", "sq", "lds"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "dp_reduce_kernel" in r["Kernel_Name"]]
        by = collections.OrderedDict()
        for r in rows:
            by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        ids = list(by)
        print(d, "dispatches", len(ids))
        for i in ids[-6:]:
            print("  ", i, {k: round(v) for k, v in by[i].items()})
PY
