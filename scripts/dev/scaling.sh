#!/bin/bash
# Development: the "scaling in N" table of DESIGN.md section 4 (bench.py at other Gaussian counts) + configs[0].
for n in 500000 2000000 8000000 32000000; do
  timeout 600 python bench.py --n-gaussians $n --steps 10 --warmup 3 --cpu-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('N=%d vis=%d M=%d ms=%.3f Mpix/s=%.0f fwd_only_ms=%s' % (d['config']['n_gaussians'], d['config']['n_visible'], d['config']['n_intersections'], d['ms_per_step'], d['value'], d['also'].get('fwd_only_ms')))"
done
timeout 300 python bench.py --n-gaussians 100000 --width 640 --height 480 --variant lean --steps 20 --warmup 5 --cpu-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('configs[0]: ms=%.3f fwd_only_ms=%s' % (d['ms_per_step'], d['also'].get('fwd_only_ms')))"
