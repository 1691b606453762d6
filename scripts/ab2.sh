#!/bin/bash
# Development: A/B the library variants under mtgs_amd/_variants with fbench, interleaved (box drift cancels): all stage times.
# usage: scripts/ab2.sh "<variant names, space separated; '-' = in-tree>" [fbench args]
names=$1; shift
for rep in 1 2 3; do
  for v in $names; do
    lib=""; [ "$v" != "-" ] && lib="--lib mtgs_amd/_variants/libmtgs_rast_$v.so"
    echo "== $v (run $rep)"
    timeout 300 python scripts/fbench.py $lib "$@" 2>&1 | grep -E "mtgs_|whole" | awk '{printf "%s %s | ", $1, $3} END {print ""}'
  done
done
