#!/bin/bash
# Per-block time line of the layer loop (wall-clock marks inside k_layer_loop, -DLOOP_TIMING) at the reference's batch
# size: builds a debug copy of the library next to the real one and runs tools/loop_timing.py on it.
#   tools/gpu_loop_timing.sh <git head>   -> profiles/r06_loop_timing_{A,B}128.txt
head=${1:-unknown}
set -e
# the debug library is built where the objects are (the build container; *.o do not travel to the GPU box):
#   hipcc ... -DLOOP_TIMING -c geossl_amd/csrc/chain.hip -o scratch/chain_timing.o
#   hipcc --offload-arch=gfx950 -shared -fPIC -o geossl_amd/lib/libgeossl_timing.so scratch/chain_timing.o <the other objects>
LIB=geossl_amd/lib/libgeossl_timing.so
[ -f $LIB ] || { echo "build $LIB first (see the header)"; exit 1; }
mkdir -p gpurun_out
for set_ in A B; do
  { echo "# wall-clock marks (s_memrealtime) of wave 1 of block 5 of k_layer_loop over one replayed step: python tools/loop_timing.py <debug lib> 128 $set_"
    echo "# (trainer, bs = 128, set $set_: forward pass, then backward pass; chain.hip built with -DLOOP_TIMING) @ $head"
    python tools/loop_timing.py $LIB 128 $set_ 2>/dev/null | grep -v amdgpu.ids; } > profiles/r06_loop_timing_${set_}128.txt
done
mkdir -p gpurun_out/lt; cp profiles/r06_loop_timing_*128.txt gpurun_out/lt/
wc -l profiles/r06_loop_timing_*128.txt
