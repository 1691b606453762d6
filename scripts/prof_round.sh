#!/bin/bash
# rocprofv3 kernel-trace summaries of a round (r04, r05: `scripts/prof_round.sh` then `python scripts/collect_profiles.py rNN`): (1) bench.py's headline step, (2) the MTGS-style iteration in its fastest
# configuration (shipped options, visibility first, row-lazy optimizer, geometry rows; eager launches: kernel times are the
# graph's) -> gpurun_out/prof_bench/, gpurun_out/prof_iter/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_bench $R/gpurun_out/prof_iter; mkdir -p $R/gpurun_out/prof_bench $R/gpurun_out/prof_iter
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 bench.py --steps 10 --warmup 3 --cpu-steps 0 --no-also --no-tight > $R/gpurun_out/prof_bench/bench.log 2>&1
echo "bench rc=$?"; tail -1 $R/gpurun_out/prof_bench/bench.log | cut -c1-120
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_iter -- python3 scripts/mtgs_like_train.py --shipped --visfirst --optimizer fused --row-lazy --geometry-rows --only fused --reps 30 ${ITER_EXTRA} > $R/gpurun_out/prof_iter/log.txt 2>&1
echo "iter rc=$?"; tail -1 $R/gpurun_out/prof_iter/log.txt | cut -c1-160
python3 scripts/show_stats.py prof_bench 24
python3 scripts/show_stats.py prof_iter 44
