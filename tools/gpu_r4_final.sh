#!/bin/bash
# round 4: regenerate the measured artefacts of the final build (run from the repository root on an MI355X box):
#   tools/gpu_r4_final.sh <git head> <tag>
head=$1; tag=${2:-v5}
out=gpurun_out/r4final; mkdir -p $out profiles; export TMPDIR=/tmp
tools/refresh_profiles.sh r04 $tag $head > $out/refresh.log 2>&1
SQ="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $out/sq -o sq -- python3 tools/prof_step.py 4 > $out/sq.log 2>&1
python tools/pmc_sq.py $(ls $out/sq/sq_counter_collection.csv $out/sq/*/sq_counter_collection.csv 2>/dev/null | head -1) \
  "schnet/ddm-step/mols=1024/set=A/cutoff=5, tools/prof_step.py 4 (eager steps)" "$head" k_filter_bwd k_filter_fwd k_ncsn k_row_chain k_aggregate k_wgrad > profiles/r04_pmc_sq_${tag}.txt 2>> $out/sq.log
# launch list of one replayed step (bs = 1024 and the reference's bs = 128)
for m in 1024 128; do
  rocprofv3 --kernel-trace -d $out/tl$m -o t -- python3 bench.py --mols $m --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  { echo "# one replayed step of python3 bench.py --mols $m (rocprofv3 --kernel-trace, tools/step_timeline.py) @ $head"
    python tools/step_timeline.py $(ls $out/tl$m/*/*.db $out/tl$m/*.db 2>/dev/null | head -1) 2 | cut -c1-72; } > profiles/r04_step_timeline_mols${m}_${tag}.txt
done
# force evaluations: kernels that are not the library's
{ echo "# Kernels that are NOT this library's in rocprofv3 --kernel-trace runs of a force evaluation (tools/aten_in_trace.py) @ $head"
  echo "# (zero fills, copies and the integer index preparation of a batch at its first sighting; the scripts' own print)"
  for m in bench schnet painn; do
    case $m in
      bench) cmd="python3 bench.py --forces --steps 5 --warmup 2";;
      *) cmd="python3 tools/force_trace.py $m 256 5";;
    esac
    rocprofv3 --kernel-trace -d $out/ft_$m -o t -- $cmd > $out/ft_$m.log 2>&1
    echo; echo "## $cmd"; python tools/aten_in_trace.py $out/ft_$m | cut -c1-170
  done; } > profiles/r04_force_eval_non_library_kernels.txt
python tools/force_trace.py painn 1024 40 2>&1 | tail -1 > $out/force_painn_1024.txt
python tools/force_trace.py schnet 1024 40 2>&1 | tail -1 > $out/force_schnet_1024.txt
mkdir -p $out/profiles; cp profiles/r04_*_${tag}*.json profiles/r04_*_${tag}*.txt profiles/r04_hbm_traffic_pmc.json profiles/r04_force_eval_non_library_kernels.txt $out/profiles/ 2>/dev/null
find $out -name "*.db" -delete; find $out -name "*.csv" -delete
ls $out/profiles; tail -2 $out/refresh.log; cat $out/force_*_1024.txt
