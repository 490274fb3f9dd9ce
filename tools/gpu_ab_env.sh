#!/bin/bash
# same-box A/B of one environment switch on the default bench line: tools/gpu_ab_env.sh VAR [steps]
var=$1; steps=${2:-60}
mkdir -p gpurun_out/r4
for rep in 1 2; do
  for mode in off on; do
    if [ $mode = on ]; then export $var=1; else unset $var; fi
    python bench.py --steps $steps --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$mode', round(d['value']), round(d['ms_per_step'], 4), {k: round(v['avg_ms'], 3) for k, v in d['kernel_ms'].items()})"
  done
done
