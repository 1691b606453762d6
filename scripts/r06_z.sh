#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_z
mkdir -p $OUT
for z in 1 0 1 0 1 0; do echo "== MTGS_ZEROED_OUTPUTS=$z"; MTGS_ZEROED_OUTPUTS=$z timeout 300 python scripts/fbench.py --sh 2>&1 | grep -E "mtgs_|whole" | awk '{printf "%s %s | ", $1, $3} END {print ""}'; done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
