#!/bin/bash
# Development: the headline step (eager ms, graph ms) on library variants, interleaved.  usage: ab_bench.sh "<names; '-' = in-tree>" [reps]
for rep in $(seq 1 ${2:-3}); do
  for v in $1; do
    lib="-"; [ "$v" != "-" ] && lib="mtgs_amd/_variants/libmtgs_rast_$v.so"
    printf "%-10s " "$v"
    python scripts/dev/bench_lib.py $lib --steps 20 --warmup 5 --cpu-steps 0 --no-also --no-tight 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step_eager'], d['ms_per_step_graph'])"
  done
done
