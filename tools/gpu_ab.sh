#!/bin/bash
# A/B of two library builds on one box: tools/gpu_ab.sh <old .so> [bench args]   (the new build is the in-tree one)
old=$1; shift
out=gpurun_out/ab; mkdir -p $out
[ -x scratch/split_probe ] && scratch/split_probe > $out/split_probe.txt 2>&1
for i in 1 2; do
  GEOSSL_HIP_LIB=$PWD/$old python bench.py --no-cpu-baseline --no-secondary "$@" | tail -1 > $out/old_$i.json
  python bench.py --no-cpu-baseline --no-secondary "$@" | tail -1 > $out/new_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/*.json')):
    d=json.loads(open(f).read())
    print(f, round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d.get('kernel_ms',{}).items()})
PY
cat $out/split_probe.txt
