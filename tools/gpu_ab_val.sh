#!/bin/bash
# same-box comparison of several values of one environment variable: tools/gpu_ab_val.sh VAR "v1 v2 ..." [bench args]
var=$1; vals=$2; shift; shift
for rep in 1 2; do
  for v in $vals; do
    if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
    python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v', round(d['value']), round(d['ms_per_step'], 4))"
  done
done
