#!/bin/bash
# Development: zeros riding on the compositing forward against a separate fill in front of it, at image sizes with few tiles
for wh in "1920 1080" "960 540" "480 270" "320 208"; do
  set -- $wh
  for v in ride sep; do
    printf "%sx%s %-5s " $1 $2 $v
    python scripts/dev/bench_lib.py mtgs_amd/_variants/libmtgs_rast_$v.so --width $1 --height $2 --steps 20 --warmup 5 --cpu-steps 0 --no-also --no-tight 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step_eager'], d['ms_per_step_graph'])"
  done
done
