#!/bin/bash
# scratch: forward with two waves per tile at the headline size (variant build with -DMTGS_DEV on the box; MTGS_PPL applies to both kernels,
# so the backward is read from the PPL=4 runs and the forward from either)
cd $GRAFT_REPO_ROOT
python scripts/build_variant.py dev -DMTGS_DEV > /dev/null 2>&1
for rep in 1 2 3; do
for ppl in 4 2; do
echo -n "rep $rep PPL=$ppl: "
MTGS_PPL=$ppl timeout 300 python scripts/fbench.py --lib mtgs_amd/_variants/libmtgs_rast_dev.so --reps 30 2>&1 | grep -E "blend_fwd" | tr '\n' ' '
echo
done; done
