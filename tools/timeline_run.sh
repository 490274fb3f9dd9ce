#!/bin/bash
# launch list of one replayed step: tools/timeline_run.sh <name> <bench args...>   -> gpurun_out/timeline_<name>.txt
name=$1; shift
export TMPDIR=/tmp
out=gpurun_out/tl_$name
rocprofv3 --kernel-trace -d $out -o t -- python3 bench.py "$@" --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
{ echo "# one replayed step of python3 bench.py $* (rocprofv3 --kernel-trace, tools/step_timeline.py)"
  python tools/step_timeline.py $(ls $out/*/*.db $out/*.db 2>/dev/null | head -1) 2 | cut -c1-72; } > gpurun_out/timeline_$name.txt
find $out -name "*.db" -delete
