# Development: run-to-run variation of the configs[4]-scale data-parallel harness run (8 ranks on one GPU) against the accumulation
cd $GRAFT_REPO_ROOT
C="--traversals 4 --width 960 --height 540 --steps 24 --refine-every 10 --reps 1 --only fused --shipped"
for i in 1 2 3 4; do
MTGS_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $((29633+i)) scripts/mtgs_like_train.py --dp --dp-exchange sparse $C 2>&1 | grep -E "refine|loss:" | cut -c1-120 | tr '\n' '|'; echo
done
for i in 1 2; do python scripts/mtgs_like_train.py --accumulate 8 $C 2>&1 | grep -E "refine|loss:" | cut -c1-120 | tr '\n' '|'; echo; done
