#!/bin/bash
# round 3: regenerate the measured artefacts of the final build (run from the repository root on an MI355X box)
head=$1; tag=${2:-v2}
out=gpurun_out/r3final; mkdir -p $out profiles; export TMPDIR=/tmp
tools/refresh_profiles.sh r03 $tag $head > $out/refresh.log 2>&1
SQ="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $out/sq -o sq -- python3 tools/prof_step.py 4 > $out/sq.log 2>&1
python tools/pmc_sq.py $(ls $out/sq/sq_counter_collection.csv $out/sq/*/sq_counter_collection.csv 2>/dev/null | head -1) \
  "schnet/ddm-step/mols=1024/set=A/cutoff=5, tools/prof_step.py 4 (eager steps)" "$head" k_filter_bwd k_filter_fwd k_ncsn k_row_chain k_aggregate k_wgrad > profiles/r03_pmc_sq_${tag}.txt 2>> $out/sq.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf -o f -- python3 tools/prof_step_painn.py 4 > $out/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pw -o w -- python3 tools/prof_step_painn.py 4 > $out/pw.log 2>&1
python tools/pmc_traffic.py $(ls $out/pf/f_counter_collection.csv $out/pf/*/f_counter_collection.csv 2>/dev/null | head -1) \
  $(ls $out/pw/w_counter_collection.csv $out/pw/*/w_counter_collection.csv 2>/dev/null | head -1) 4 1024 \
  "painn/ddm-step/mols=1024/set=A/cutoff=5" "$head" > profiles/r03_painn_hbm_traffic_pmc.json 2> $out/painn_pmc.err
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $out/sqp -o sq -- python3 tools/prof_step_painn.py 4 > $out/sqp.log 2>&1
python tools/pmc_sq.py $(ls $out/sqp/sq_counter_collection.csv $out/sqp/*/sq_counter_collection.csv 2>/dev/null | head -1) \
  "painn/ddm-step/mols=1024/set=A/cutoff=5, tools/prof_step_painn.py 4 (eager steps)" "$head" k_painn > profiles/r03_pmc_sq_painn_${tag}.txt 2>> $out/sqp.log
for extra in "--model painn --max-batches 8:painn" "--set B --max-batches 24:setB" "--cutoff 10:cutoff10" "--forward-only:forward_only" "--forces:forces" "--api reference:reference_api" "--mols 128:mols128" "--api reference --mols 128:reference_api_mols128"; do
  flags=${extra%%:*}; name=${extra##*:}
  python bench.py $flags --steps 50 --warmup 10 --no-secondary 2>> $out/bench.err | tail -1 > profiles/r03_bench_${name}_${tag}.json
done
ls -la profiles | grep r03; tail -3 $out/refresh.log
# only gpurun_out/ travels back from the GPU box: the profile files written above go there as well
mkdir -p $out/profiles; cp profiles/r03_*_${tag}*.json profiles/r03_*_${tag}*.txt profiles/r03_hbm_traffic_pmc.json profiles/r03_painn_hbm_traffic_pmc.json $out/profiles/ 2>/dev/null
