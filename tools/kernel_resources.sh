#!/bin/bash
# Register / scratch use of every kernel of one csrc translation unit (code-object metadata of the gfx950 build):
#   tools/kernel_resources.sh ncsn_bwd.hip [extra hipcc flags]
src=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/kres_XXXX)
(cd $tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I "$here/include" -I "$here/geossl_amd/csrc" -Wno-pass-failed "$@" \
  -c "$here/geossl_amd/csrc/$src" -o x.o -save-temps=obj 2>/dev/null)
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|private_segment_fixed_size|agpr_count|sgpr_spill_count):" $tmp/*gfx950.s | \
  awk '{v[$1]=$2} /vgpr_spill_count/ {printf "%-90s vgpr %3s agpr %3s spill %3s scratch %4s sgpr_spill %s\n", substr(v[".name:"],1,90), v[".vgpr_count:"], v[".agpr_count:"], v[".vgpr_spill_count:"], v[".private_segment_fixed_size:"], v[".sgpr_spill_count:"]}'
rm -rf $tmp
