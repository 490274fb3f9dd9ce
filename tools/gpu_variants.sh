#!/bin/bash
# headline bench with alternative library builds, interleaved twice: tools/gpu_variants.sh a.so b.so ...  ("-" = in-tree)
out=gpurun_out/variants; mkdir -p $out; rm -f $out/*.json
for i in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then name=intree; unset GEOSSL_HIP_LIB; else name=$(basename $lib .so); export GEOSSL_HIP_LIB=$PWD/$lib; fi
    python bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 15 $BENCH_ARGS | tail -1 > $out/${name}_$i.json
  done
done
unset GEOSSL_HIP_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/variants/*.json')):
    d=json.loads(open(f).read())
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k.replace('geossl_',''):round(v['avg_ms'],4) for k,v in d.get('kernel_ms',{}).items()})
PY
