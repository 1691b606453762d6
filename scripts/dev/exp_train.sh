#!/bin/bash
# scratch: launch-bound sweeps of the compositing kernels with the tight lists (variant builds on the box)
cd $GRAFT_REPO_ROOT
for v in "bw3 -DMTGS_BWD_WAVES=3" "bw5 -DMTGS_BWD_WAVES=5" "fw4 -DMTGS_FWD_WAVES=4" "fw8 -DMTGS_FWD_WAVES=8"; do set -- $v
python scripts/build_variant.py $1 $2 > /dev/null 2>&1 &
done; wait
ls mtgs_amd/_variants/*.so
for rep in 1 2; do
for lib in "" mtgs_amd/_variants/libmtgs_rast_bw3.so mtgs_amd/_variants/libmtgs_rast_bw5.so mtgs_amd/_variants/libmtgs_rast_fw4.so mtgs_amd/_variants/libmtgs_rast_fw8.so; do
echo -n "rep $rep ${lib:-current}: "
timeout 300 python scripts/fbench.py ${lib:+--lib $lib} --reps 30 2>&1 | grep -E "blend" | tr '\n' ' '
echo
done; done
