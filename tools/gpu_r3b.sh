#!/bin/bash
# round 3: fused CFConv tests + A/B bench, profile of the reference-shaped loop
out=gpurun_out/r3b; mkdir -p $out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_round3.py -x -q > $out/pytest3.log 2>&1; echo "rc $?" >> $out/pytest3.log
GEOSSL_FUSED=full python -m pytest tests/test_gpu_parity.py -x -q -k "golden or full_size or ddm" > $out/pytest_fused_full.log 2>&1; echo "rc $?" >> $out/pytest_fused_full.log
for mode in "" fwd full ""; do
  GEOSSL_FUSED=$mode python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > $out/bench_fused_${mode:-off}_$RANDOM.json 2>> $out/bench.err
done
python tools/ref_loop_profile.py 1024 > $out/ref_loop_1024.txt 2>&1
python tools/ref_loop_profile.py 128 > $out/ref_loop_128.txt 2>&1
tail -4 $out/pytest3.log; tail -3 $out/pytest_fused_full.log
for f in $out/bench_fused_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read()); print(d['value'], d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['kernel_ms'].items()})"; done
cat $out/ref_loop_1024.txt $out/ref_loop_128.txt
