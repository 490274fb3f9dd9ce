#!/bin/bash
# headline bench under different values of one environment variable, interleaved twice:
#   tools/gpu_env_ab.sh VAR v1 v2 ...     ("-" = unset)   [BENCH_ARGS="--mols 128"]
var=$1; shift
out=gpurun_out/envab; mkdir -p $out; rm -f $out/*.json
for i in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $var; name=unset; else export $var=$v; name=$v; fi
    python bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 15 $BENCH_ARGS | tail -1 > $out/${name}_$i.json
  done
done
unset $var
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/envab/*.json')):
    try:
        d=json.loads(open(f).read())
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4))
    except Exception as e:
        print(f, 'FAILED', open(f).read()[:200])
PY
