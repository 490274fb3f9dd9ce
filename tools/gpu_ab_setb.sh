#!/bin/bash
# same-box A/B of one environment switch on ragged batches: tools/gpu_ab_setb.sh VAR mols [api]
var=$1; mols=${2:-128}; api=${3:-trainer}
for rep in 1 2; do
  for mode in off on; do
    if [ $mode = on ]; then export $var=1; else unset $var; fi
    python bench.py --set B --mols $mols --api $api --steps 200 --warmup 20 --max-batches 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$mode mols=$mols $api', round(d['value']), round(d['ms_per_step'], 4))"
  done
done
