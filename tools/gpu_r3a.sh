#!/bin/bash
# round 3, first GPU call: all GPU tests, the default bench line (with secondaries), SQ counters of the round-2 kernels
# ("before" of the filter-backward work), HBM traffic of the PaiNN step
out=gpurun_out/r3a; mkdir -p $out; export TMPDIR=/tmp
head=$1
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
python bench.py > $out/bench.json 2> $out/bench.err
SQ="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $out/sq -o sq -- python3 tools/prof_step.py 4 > $out/sq.log 2>&1
python tools/pmc_sq.py $(ls $out/sq/*/sq_counter_collection.csv $out/sq/sq_counter_collection.csv 2>/dev/null | head -1) \
  "schnet/ddm-step/mols=1024/set=A/cutoff=5, tools/prof_step.py 4 (eager steps)" "$head" k_filter_bwd k_filter_fwd k_ncsn k_row_chain k_aggregate k_wgrad > $out/pmc_sq.txt 2>> $out/sq.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf -o f -- python3 tools/prof_step_painn.py 4 > $out/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pw -o w -- python3 tools/prof_step_painn.py 4 > $out/pw.log 2>&1
python tools/pmc_traffic.py $(ls $out/pf/*/f_counter_collection.csv $out/pf/f_counter_collection.csv 2>/dev/null | head -1) \
  $(ls $out/pw/*/w_counter_collection.csv $out/pw/w_counter_collection.csv 2>/dev/null | head -1) 4 1024 \
  "painn/ddm-step/mols=1024/set=A/cutoff=5" "$head" > $out/painn_hbm_traffic_pmc.json 2> $out/painn_pmc.err
tail -3 $out/pytest.log; head -c 1500 $out/bench.json; tail -5 $out/bench.err
