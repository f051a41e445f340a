#!/bin/bash
# Round 3, final session at HEAD: full GPU suite, smoke, the bench lines (4 / 16 / 128 pairs, data-parallel schedule), kernel
# traces (eager + whole-step graph), the three PMC passes of the step at 4 pairs, the stall-breakdown pass of the GEMMs at 128 pairs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
BUDGET=${BUDGET:-960}
OUT=gpurun_out/r03z
mkdir -p $OUT
P=$PWD
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
t=$(lim 480)
(timeout $t python -m pytest tests/test_model_gpu.py -m gpu -q --durations=10 -p no:cacheprovider > $OUT/pytest_model.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_model.log) &
P1=$!
(timeout $t python -m pytest tests -m gpu -q --durations=8 --ignore=tests/test_model_gpu.py -p no:cacheprovider > $OUT/pytest_rest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_rest.log) &
P2=$!
wait $P1 $P2
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_model.log | tail -20; grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_rest.log | tail -10
stamp "pytest done"
t=$(lim 60); [ $t -gt 0 ] && { (timeout $t python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log); tail -2 $OUT/smoke.log; }
t=$(lim 150); [ $t -gt 0 ] && { timeout $t python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json; }
stamp "bench done"
ab() {   # name env... -- args
  local name=$1 t; shift
  t=$(lim 70); [ $t -gt 10 ] || return
  env "$@" timeout $t python bench.py --steps $STEPS --warmup 15 --no-cpu-baseline $EXTRA > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"graph_mode": "[a-z]*"' $OUT/bench_$name.json) $(grep -o '"exposed_ms": [0-9.]*' $OUT/bench_$name.json) $(grep -o '"algbw_gbs": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/bench_summary.txt
}
STEPS=150
EXTRA="--no-extras" ab b4 UNIVL_X=0
EXTRA="--no-extras --no-pipeline" ab b4_nopipe UNIVL_X=0
EXTRA="--no-extras --batch 16" ab b16 UNIVL_X=0
STEPS=60
EXTRA="--no-extras --batch 128" ab b128 UNIVL_X=0
STEPS=150
EXTRA="--force-dp" ab b4_dp_rccl_world1 UNIVL_X=0
EXTRA="--no-extras --force-dp" ab b4_dp_dryrun UNIVL_DP_DRYRUN=1
EXTRA="--no-extras --force-dp" ab b4_dp_pg_segmented UNIVL_DP_CAPTURE=0
stamp "bench lines done"
t=$(lim 90); [ $t -gt 20 ] && { (cd /tmp && timeout $t rocprofv3 --kernel-trace --stats -d $P/$OUT/prof -o eager --output-format csv -- python $P/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-extras > $P/$OUT/prof_bench.json 2> $P/$OUT/prof_bench.err)
  find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/eager_kernel_stats.csv \; ; rm -rf $OUT/prof; }
t=$(lim 90); [ $t -gt 20 ] && { (cd /tmp && timeout $t rocprofv3 --kernel-trace --stats -d $P/$OUT/profg -o graph --output-format csv -- python $P/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $P/$OUT/profg_bench.json 2> $P/$OUT/profg_bench.err)
  find $OUT/profg -name "*kernel_stats.csv" -exec cp {} $OUT/graph_kernel_stats.csv \; ; find $OUT/profg -name "*kernel_trace.csv" -exec gzip -c {} \; > $OUT/graph_kernel_trace.csv.gz; rm -rf $OUT/profg; }
stamp "traces done"
t=$(lim 70); [ $t -gt 25 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/$OUT/pmc_fetch --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_fetch.log 2>&1)
t=$(lim 70); [ $t -gt 25 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/$OUT/pmc_write --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_write.log 2>&1)
t=$(lim 70); [ $t -gt 25 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/$OUT/pmc_mfma --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_mfma.log 2>&1)
EL=$(grep -o "[0-9]* flat elements" $OUT/pmc_fetch.log | grep -o "^[0-9]*")
python scripts/pmc_step_parse.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma ${EL:-153784064} 4 $OUT/gemm_pmc.json > $OUT/pmc_parse.log 2>&1
for k in fetch write mfma; do find $OUT/pmc_$k -name "*counter_collection.csv" -exec gzip -c {} \; > $OUT/pmc_$k.csv.gz; rm -rf $OUT/pmc_$k; done
stamp "pmc b4 done"
t=$(lim 90); [ $t -gt 30 ] && { (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $P/$OUT/pmc_stall128 --output-format csv -- python $P/scripts/pmc_step.py 128 > $P/$OUT/pmc_stall128.log 2>&1)
  python scripts/pmc_stall_parse.py $OUT/pmc_stall128 > $OUT/pmc_stall_b128.txt 2>&1; find $OUT/pmc_stall128 -name "*counter_collection.csv" -exec gzip -c {} \; > $OUT/pmc_stall128.csv.gz; rm -rf $OUT/pmc_stall128; cat $OUT/pmc_stall_b128.txt | head -30; }
stamp "end"
