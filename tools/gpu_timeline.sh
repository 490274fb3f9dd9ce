#!/bin/bash
# ordered launch list of one step at bs = 1024 and bs = 128 -> gpurun_out/timeline/
export TMPDIR=/tmp
out=gpurun_out/timeline; mkdir -p $out
for m in 1024 128; do
  rocprofv3 --kernel-trace -d $out/t$m -o t -- python3 bench.py --mols $m --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $out/bench_$m.log 2>&1
  python tools/step_timeline.py "$(ls $out/t$m/*.db $out/t$m/*/*.db 2>/dev/null | head -1)" 2 > $out/timeline_$m.txt 2>&1
  rm -rf $out/t$m
done
tail -3 $out/timeline_1024.txt
