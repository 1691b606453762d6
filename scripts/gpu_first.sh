set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python __graft_entry__.py --smoke 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 2>&1 | tail -3 | tee gpurun_out/bench_first.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o r01_first -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof1.log
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof1 | head
