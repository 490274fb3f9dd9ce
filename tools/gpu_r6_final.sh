#!/bin/bash
# round 6: regenerate the measured artefacts of the final build (run from the repository root on an MI355X box):
#   tools/gpu_r6_final.sh <git head> <tag>
head=$1; tag=${2:-v1}; R=r06
out=gpurun_out/r6final; mkdir -p $out profiles; export TMPDIR=/tmp
first_csv() { ls $1/$2_counter_collection.csv $1/*/$2_counter_collection.csv 2>/dev/null | head -1; }
# ---- HBM traffic per kernel (PMC, separate passes) for the three workloads the bench line reports traffic for.  FIRST: the
# bench lines below read these files (stamped with the hash of csrc/ + include/: a summary of another build is not reported)
for what in schnet painn forward; do
  case $what in
    schnet) prog="tools/prof_step.py 4"; wl="schnet/ddm-step/mols=1024/set=A/cutoff=5"; name=${R}_hbm_traffic_pmc.json;;
    painn) prog="tools/prof_step_painn.py 4"; wl="painn/ddm-step/mols=1024/set=A/cutoff=5"; name=${R}_painn_hbm_traffic_pmc.json;;
    forward) prog="tools/prof_forward.py 4"; wl="schnet/forward/mols=1024/set=A/cutoff=5"; name=${R}_forward_hbm_traffic_pmc.json;;
  esac
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_f_$what -o f -- python3 $prog > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_w_$what -o w -- python3 $prog > /dev/null 2>&1
  python tools/pmc_traffic.py "$(first_csv $out/pmc_f_$what f)" "$(first_csv $out/pmc_w_$what w)" 4 1024 "$wl" "$head" \
    > profiles/$name 2> $out/pmc_$what.err
done
# ---- SQ counters of the two backbones' steps
SQ="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $out/sq -o sq -- python3 tools/prof_step.py 4 > $out/sq.log 2>&1
python tools/pmc_sq.py "$(first_csv $out/sq sq)" "schnet/ddm-step/mols=1024/set=A/cutoff=5, tools/prof_step.py 4 (eager steps)" "$head" \
  k_filter_bwd k_filter_fwd k_ncsn k_row_chain k_aggregate k_wgrad > profiles/${R}_pmc_sq_${tag}.txt 2>> $out/sq.log
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $out/sqp -o sq -- python3 tools/prof_step_painn.py 4 > $out/sqp.log 2>&1
python tools/pmc_sq.py "$(first_csv $out/sqp sq)" "painn/ddm-step/mols=1024/set=A/cutoff=5, tools/prof_step_painn.py 4 (eager steps)" "$head" \
  k_painn k_row_chain k_wgrad > profiles/${R}_pmc_sq_painn_${tag}.txt 2>> $out/sqp.log
# ---- kernel stats of the default bench command (what the driver runs) and of the PaiNN step
rocprofv3 --kernel-trace -d $out/trace -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary, summarised by tools/rocpd_stats.py @ $head"
  python tools/rocpd_stats.py "$(ls $out/trace/*.db $out/trace/*/*.db 2>/dev/null | head -1)" 60; } > profiles/${R}_bench_${tag}_kernel_stats.txt
rocprofv3 --kernel-trace -d $out/trace_painn -o t -- python3 bench.py --model painn --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace -- python3 bench.py --model painn --steps 10 --warmup 3 --no-cpu-baseline, summarised by tools/rocpd_stats.py @ $head"
  python tools/rocpd_stats.py "$(ls $out/trace_painn/*.db $out/trace_painn/*/*.db 2>/dev/null | head -1)" 40; } > profiles/${R}_bench_painn_${tag}_kernel_stats.txt
# ---- launch lists of one replayed step
tl() {  # name, bench args
  name=$1; shift
  rocprofv3 --kernel-trace -d $out/tl_$name -o t -- python3 bench.py "$@" --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  { echo "# one replayed step of python3 bench.py $* (rocprofv3 --kernel-trace, tools/step_timeline.py) @ $head"
    python tools/step_timeline.py $(ls $out/tl_$name/*/*.db $out/tl_$name/*.db 2>/dev/null | head -1) 2 | cut -c1-72; } > profiles/${R}_step_timeline_${name}_${tag}.txt
}
tl mols1024
tl mols128 --mols 128
tl setB_mols128 --mols 128 --set B
tl setC_cutoff10_mols128 --mols 128 --set C --cutoff 10
tl painn_mols1024 --model painn
tl painn_setC_mols128 --model painn --mols 128 --set C
# ---- the train-on-forces step: kernels of the captured graph
for bb in schnet painn; do
  rocprofv3 --kernel-trace -d $out/ft_$bb -o t -- python3 tools/force_graph_trace.py $bb 256 10 > $out/ft_$bb.log 2>&1
  { echo "# rocprofv3 --kernel-trace -- python3 tools/force_graph_trace.py $bb 256 10 (ForceTrainer: the finetune_md17.py:30-54 step replayed as one captured graph), tools/rocpd_stats.py @ $head"
    tail -1 $out/ft_$bb.log | cut -c1-200
    python tools/rocpd_stats.py $(ls $out/ft_$bb/*.db $out/ft_$bb/*/*.db 2>/dev/null | head -1) 30 | cut -c1-150; } > profiles/${R}_force_train_${bb}_${tag}_kernel_stats.txt
done
# ---- bench lines (compact line + detail)
python bench.py > profiles/${R}_bench_${tag}.json 2> $out/bench.err; cp gpurun_out/bench_detail.json profiles/${R}_bench_${tag}_detail.json
python bench.py --model painn --no-secondary > profiles/${R}_bench_painn_${tag}.json 2> $out/bench_painn.err
python bench.py --forward-only > profiles/${R}_bench_forward_only_${tag}.json 2> $out/bench_fwd.err
# ---- long runs on shuffled epochs over the device-resident dataset (memory, RSS and step time flat; captures)
{ python tools/soak.py 3000 128 trainer painn C dataset 2>&1 | tail -11; python tools/soak.py 3000 128 trainer schnet C dataset 2>&1 | tail -11;
  python tools/soak.py 3000 128 reference schnet B dataset 2>&1 | tail -11; } > profiles/${R}_soak_${tag}.txt
mkdir -p $out/profiles; cp profiles/${R}_* $out/profiles/ 2>/dev/null
find $out -name "*.db" -delete; find $out -name "*.csv" -delete
ls $out/profiles
