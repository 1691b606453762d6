#!/bin/bash
# Development: A/B the library variants under mtgs_amd/_variants with kbench (blend stages only shown).
# usage: scripts/ab.sh [kbench args]
for lib in "" mtgs_amd/_variants/*.so; do
  for rep in 1 2; do
    echo "== ${lib:-current} (run $rep)"
    timeout 300 python scripts/kbench.py ${lib:+--lib $lib} "$@" 2>&1 | grep -E "blend|total|isect|sh_|project" | tr '\n' ' '
    echo
  done
done
