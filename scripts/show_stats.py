#!/usr/bin/env python3
"""Development: print the newest rocprofv3 kernel_stats.csv under gpurun_out/<dir> (default prof_bench)."""
import csv, glob, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "prof_bench"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 26
f = max(glob.glob(f"gpurun_out/{d}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
print(f)
for r in list(csv.DictReader(open(f)))[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{name[:72]:72s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:8.1f}us tot%={r['Percentage']}")
