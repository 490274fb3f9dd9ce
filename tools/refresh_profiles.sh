#!/bin/bash
# Regenerate the measured artefacts under profiles/ on an MI355X box (run from the repository root):
#   tools/refresh_profiles.sh r02 v1 <git head of the build>
# 1. default bench line (with the CPU baseline)          -> profiles/<round>_bench_<tag>.json
# 2. rocprofv3 kernel trace of the same command          -> profiles/<round>_bench_<tag>_kernel_stats.txt
# 3. HBM traffic per launch, two separate PMC passes     -> profiles/<round>_hbm_traffic_pmc.json
# PMC passes are never combined with the sys/runtime/hip/hsa trace domains, and the program itself follows `--`.
set -e
round=${1:-r02}; tag=${2:-vX}; head=${3:-unknown}
out=gpurun_out/refresh_$tag
mkdir -p "$out" profiles
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$out/trace" -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary, summarised by tools/rocpd_stats.py"
  python tools/rocpd_stats.py "$(ls "$out"/trace/*.db | head -1)" 60; } > profiles/${round}_bench_${tag}_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_f" -o f -- python3 tools/prof_step.py 4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_w" -o w -- python3 tools/prof_step.py 4 > /dev/null 2>&1
python tools/pmc_traffic.py "$out/pmc_f/f_counter_collection.csv" "$out/pmc_w/w_counter_collection.csv" 4 1024 \
  "schnet/ddm-step/mols=1024/set=A/cutoff=5" "$head" > profiles/${round}_hbm_traffic_pmc.json
# the bench line last: its roofline.traffic is read from the PMC file written above (same build)
python bench.py | tail -1 > profiles/${round}_bench_${tag}.json
echo "wrote profiles/${round}_bench_${tag}.json, ${round}_bench_${tag}_kernel_stats.txt, ${round}_hbm_traffic_pmc.json"
