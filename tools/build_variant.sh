#!/bin/bash
# Build an ablation / A-B variant of libgeossl_hip.so: one source recompiled with extra flags, linked with the other
# objects of the current build.   tools/build_variant.sh chain.hip out.so -DCHAIN_ABLATE_DMA
set -e
src=$1; out=$2; shift 2
here=$(cd "$(dirname "$0")/.." && pwd)
obj=$(mktemp /tmp/variant_XXXX.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I "$here/include" -I "$here/geossl_amd/csrc" -Wno-pass-failed "$@" -c "$here/geossl_amd/csrc/$src" -o "$obj"
others=$(ls "$here"/geossl_amd/lib/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $others "$obj"
rm -f "$obj"
