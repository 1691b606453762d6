#!/bin/bash
# Round 6: the compositing forward with one wave per tile (MTGS_PPL=4) at 4 / 5 / 6 waves per SIMD against the shipped two waves per tile
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== in-tree (fwd: 2 waves per tile)"; timeout 300 python scripts/fbench.py 2>&1 | grep -E "blend_fwd|whole"
  for w in 4 5 6; do
    echo "== fwdw$w, MTGS_PPL=4"; MTGS_PPL=4 timeout 300 python scripts/fbench.py --lib mtgs_amd/_variants/libmtgs_rast_fwdw$w.so 2>&1 | grep -E "blend_fwd|whole"
  done
done
