#!/bin/bash
# ordered launch list of one step -> gpurun_out/timeline/timeline_<tag>.txt;  tools/gpu_timeline.sh <tag> [bench args]
export TMPDIR=/tmp
tag=$1; shift
marker=${MARKER:-k_adam}
out=gpurun_out/timeline; mkdir -p $out
rocprofv3 --kernel-trace -d $out/t$tag -o t -- python3 bench.py "$@" --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $out/bench_$tag.log 2>&1
python tools/step_timeline.py "$(ls $out/t$tag/*.db $out/t$tag/*/*.db 2>/dev/null | head -1)" ${BACK:-2} $marker > $out/timeline_$tag.txt 2>&1
rm -rf $out/t$tag
tail -2 $out/timeline_$tag.txt
