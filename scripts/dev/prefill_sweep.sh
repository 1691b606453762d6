#!/bin/bash
# Development: the headline step with the SH backward's zero fill overlapped (mtgs_amd/wrapper.py::_Prefill) for several sizes of the
# background fill, interleaved repeats (box drift cancels).  usage: prefill_sweep.sh "<cfg>;<cfg>;..." reps
IFS=';' read -ra CFGS <<< "${1:-MTGS_SH_PREFILL=0;MTGS_PREFILL_BLOCKS=32;MTGS_PREFILL_BLOCKS=64}"
for rep in $(seq 1 ${2:-3}); do
  for cfg in "${CFGS[@]}"; do
    printf "%-50s " "$cfg"
    env $cfg python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-also --no-tight 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step_eager'], d['ms_per_step_graph'])"
  done
done
