#!/bin/bash
# round 5: regenerate the measured artefacts of the final build (run from the repository root on an MI355X box):
#   tools/gpu_r5_final.sh <git head> <tag>
head=$1; tag=${2:-v1}
out=gpurun_out/r5final; mkdir -p $out profiles; export TMPDIR=/tmp
# PMC traffic of the second backbone and of the forward-only pass FIRST: the bench line reads them (roofline.traffic)
for what in painn forward; do
  case $what in
    painn) prog="tools/prof_step_painn.py 4"; wl="painn/ddm-step/mols=1024/set=A/cutoff=5"; steps=4;;
    forward) prog="tools/prof_forward.py 4"; wl="schnet/forward/mols=1024/set=A/cutoff=5"; steps=4;;
  esac
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_f_$what -o f -- python3 $prog > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_w_$what -o w -- python3 $prog > /dev/null 2>&1
  python tools/pmc_traffic.py "$(ls $out/pmc_f_$what/f_counter_collection.csv $out/pmc_f_$what/*/f_counter_collection.csv 2>/dev/null | head -1)" \
    "$(ls $out/pmc_w_$what/w_counter_collection.csv $out/pmc_w_$what/*/w_counter_collection.csv 2>/dev/null | head -1)" $steps 1024 "$wl" "$head" \
    > profiles/r05_${what}_hbm_traffic_pmc.json 2> $out/pmc_$what.err
done
tools/refresh_profiles.sh r05 $tag $head > $out/refresh.log 2>&1
SQ="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $out/sq -o sq -- python3 tools/prof_step.py 4 > $out/sq.log 2>&1
python tools/pmc_sq.py $(ls $out/sq/sq_counter_collection.csv $out/sq/*/sq_counter_collection.csv 2>/dev/null | head -1) \
  "schnet/ddm-step/mols=1024/set=A/cutoff=5, tools/prof_step.py 4 (eager steps)" "$head" k_filter_bwd k_filter_fwd k_ncsn k_row_chain k_aggregate k_wgrad > profiles/r05_pmc_sq_${tag}.txt 2>> $out/sq.log
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $out/sqp -o sq -- python3 tools/prof_step_painn.py 4 > $out/sqp.log 2>&1
python tools/pmc_sq.py $(ls $out/sqp/sq_counter_collection.csv $out/sqp/*/sq_counter_collection.csv 2>/dev/null | head -1) \
  "painn/ddm-step/mols=1024/set=A/cutoff=5, tools/prof_step_painn.py 4 (eager steps)" "$head" k_painn k_row_chain k_wgrad > profiles/r05_pmc_sq_painn_${tag}.txt 2>> $out/sqp.log
# launch lists of one replayed step: the headline, the reference's batch size, and what the reference's script feeds
tl() {  # name, bench args
  name=$1; shift
  rocprofv3 --kernel-trace -d $out/tl_$name -o t -- python3 bench.py "$@" --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  { echo "# one replayed step of python3 bench.py $* (rocprofv3 --kernel-trace, tools/step_timeline.py) @ $head"
    python tools/step_timeline.py $(ls $out/tl_$name/*/*.db $out/tl_$name/*.db 2>/dev/null | head -1) 2 | cut -c1-72; } > profiles/r05_step_timeline_${name}_${tag}.txt
}
tl mols1024
tl mols128 --mols 128
tl setC_cutoff10_mols128 --mols 128 --set C --cutoff 10
tl painn_mols1024 --model painn
tl painn_setC_mols128 --model painn --mols 128 --set C
# kernel stats of the PaiNN step
rocprofv3 --kernel-trace -d $out/trace_painn -o t -- python3 bench.py --model painn --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace -- python3 bench.py --model painn --steps 10 --warmup 3 --no-cpu-baseline, summarised by tools/rocpd_stats.py @ $head"
  python tools/rocpd_stats.py "$(ls $out/trace_painn/*.db $out/trace_painn/*/*.db 2>/dev/null | head -1)" 40; } > profiles/r05_bench_painn_${tag}_kernel_stats.txt
python bench.py --model painn 2>/dev/null | tail -1 > profiles/r05_bench_painn_${tag}.json
python bench.py --forward-only 2>/dev/null | tail -1 > profiles/r05_bench_forward_only_${tag}.json
# train-on-forces steps: which kernels are not the library's (second order on the tape)
bash tools/gpu_force_train_trace.sh $head 256 6 > /dev/null 2>&1
cut -c1-200 gpurun_out/force_train_non_library_kernels.txt > profiles/r05_force_train_non_library_kernels.txt
# long runs on never-repeating batches (memory, RSS and step time flat; captures)
{ python tools/soak.py 3000 128 trainer painn C 2>&1 | tail -6; python tools/soak.py 3000 128 reference schnet C 2>&1 | tail -6; } > profiles/r05_soak_${tag}.txt
mkdir -p $out/profiles; cp profiles/r05_* $out/profiles/ 2>/dev/null
find $out -name "*.db" -delete; find $out -name "*.csv" -delete
ls $out/profiles; tail -2 $out/refresh.log
