#!/bin/bash
# Development: A/B the library variants under mtgs_amd/_variants with fbench (stage times of the fused path).
# usage: scripts/ab.sh [fbench args]
for lib in "" mtgs_amd/_variants/*.so; do
  for rep in 1 2; do
    echo "== ${lib:-current} (run $rep)"
    timeout 300 python scripts/fbench.py ${lib:+--lib $lib} "$@" 2>&1 | grep -E "blend|whole" | tr '\n' ' '
    echo
  done
done
