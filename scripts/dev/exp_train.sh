#!/bin/bash
# scratch script of round 4 (training / exchange experiments run through gpurun); the last content: the final bench line
mkdir -p gpurun_out
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_r04_final.json
python -c "
import json; d=json.load(open('gpurun_out/bench_r04_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['avg_launch_ms'], d['roofline'].get('valu_busy_frac')); print(json.dumps(d['also'])); print(d['cpu_baseline'])"
