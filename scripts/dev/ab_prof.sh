#!/bin/bash
# Development: per-kernel times (rocprofv3 kernel trace) of scripts/fbench.py for the in-tree library and every variant
# under mtgs_amd/_variants; prints the kernels whose name matches $1 (default bin3).
pat=${1:-bin3}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "" $R/mtgs_amd/_variants/*.so; do
  name=$(basename "${lib:-in-tree}" .so)
  out=$R/gpurun_out/ab_prof/$name
  rm -rf $out; mkdir -p $out
  (cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 scripts/fbench.py ${lib:+--lib $lib} "$@" > $out/log.txt 2>&1)
  echo "== $name"
  grep "whole step\|bin3_build" $out/log.txt
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$pat" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"] or "zero_kernel" in r["Name"]:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"   {n[:60]:60s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:8.1f}us")
PY
done
