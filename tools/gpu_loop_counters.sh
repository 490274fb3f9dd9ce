#!/bin/bash
# counters of the layer-loop kernel (forced on for the eager steps of tools/prof_step.py): SQ activity and HBM traffic
head=$1
out=gpurun_out/loopc; mkdir -p $out; export TMPDIR=/tmp
export GEOSSL_LAYER_LOOP=1
SQ="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $out/sq -o sq -- python3 tools/prof_step.py 4 > $out/sq.log 2>&1
python tools/pmc_sq.py $(ls $out/sq/sq_counter_collection.csv $out/sq/*/sq_counter_collection.csv 2>/dev/null | head -1) \
  "schnet/ddm-step/mols=1024/set=A/cutoff=5, tools/prof_step.py 4 (eager steps, GEOSSL_LAYER_LOOP=1)" "$head" k_layer_loop k_ncsn > $out/r03_pmc_sq_layer_loop.txt 2>> $out/sq.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf -o f -- python3 tools/prof_step.py 4 > $out/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pw -o w -- python3 tools/prof_step.py 4 > $out/pw.log 2>&1
python tools/pmc_traffic.py $(ls $out/pf/f_counter_collection.csv $out/pf/*/f_counter_collection.csv 2>/dev/null | head -1) \
  $(ls $out/pw/w_counter_collection.csv $out/pw/*/w_counter_collection.csv 2>/dev/null | head -1) 4 1024 \
  "schnet/ddm-step/mols=1024/set=A/cutoff=5/layer-loop" "$head" > $out/r03_hbm_traffic_pmc_layer_loop.json 2> $out/pmc.err
ls -la $out
