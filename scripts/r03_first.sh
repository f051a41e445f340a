#!/bin/bash
# First GPU session of round 3 (prepared at the end of round 2, when the GPU budget was spent): everything HEAD of round 2 could
# not be given any more -- the full GPU suite with the weight gradients riding in their dgrad launches (default since a9a865f),
# the bench lines at 4 / 16 / 128 pairs, kernel traces (eager + whole-step graph), the three PMC passes of the new kernels, and
# the same-box A/B of UNIVL_WGRAD_RIDE at 4 and 16 pairs.  ~9 minutes of box time; every step bounded by what is left of BUDGET.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
BUDGET=${BUDGET:-560}
OUT=gpurun_out/r03a
mkdir -p gpurun_out/r03a
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
P=$PWD
t=$(lim 420)
(timeout $t python -m pytest tests/test_model_gpu.py -m gpu -q --durations=8 > $OUT/pytest_model.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_model.log) &
P1=$!
(timeout $t python -m pytest tests -m gpu -q --durations=8 --ignore=tests/test_model_gpu.py > $OUT/pytest_rest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_rest.log) &
P2=$!
wait $P1 $P2
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
tail -3 $OUT/pytest_model.log; tail -3 $OUT/pytest_rest.log
stamp "pytest done"
t=$(lim 60); [ $t -gt 0 ] && { (timeout $t python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log); tail -2 $OUT/smoke.log; }
t=$(lim 120); [ $t -gt 0 ] && { timeout $t python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-260 $OUT/bench.json; }
stamp "bench done"
ab() {   # name batch env...
  local name=$1 batch=$2 t; shift 2
  t=$(lim 60); [ $t -gt 0 ] || return
  env "$@" timeout $t python bench.py --batch $batch --steps 150 --warmup 15 --no-cpu-baseline --no-extras > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt
}
ab b4_ride 4 UNIVL_X=0
ab b4_grouped 4 UNIVL_WGRAD_RIDE=0
ab b4_ride2 4 UNIVL_X=0
ab b4_grouped2 4 UNIVL_WGRAD_RIDE=0
ab b16_ride 16 UNIVL_X=0
ab b16_grouped 16 UNIVL_WGRAD_RIDE=0
ab b128 128 UNIVL_X=0
t=$(lim 60); [ $t -gt 0 ] && { timeout $t python bench.py --loopback --steps 150 --warmup 15 --no-cpu-baseline --no-extras > $OUT/ab_b4_loopback.json 2> $OUT/ab_b4_loopback.err; echo "b4_loopback: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_b4_loopback.json)" | tee -a $OUT/ab_summary.txt; }
stamp "A/B done"
# EXPERIMENTAL (written blind at the end of round 2, never run on a GPU): the BertAdam update riding with the next forward
t=$(lim 150); [ $t -gt 30 ] && { (UNIVL_EXPERIMENTAL=1 timeout $t python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "adam_update_riding" > $OUT/pytest_adam_ride.log 2>&1; echo "rc=$?" >> $OUT/pytest_adam_ride.log); tail -3 $OUT/pytest_adam_ride.log; }
abp() {   # name env...   (pipelined optimizer forms)
  local name=$1 t; shift
  t=$(lim 60); [ $t -gt 0 ] || return
  env "$@" timeout $t python bench.py --pipeline --steps 150 --warmup 15 --no-cpu-baseline --no-extras > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json) $(grep -o '"last_loss": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt
}
abp b4_adam_ride UNIVL_ADAM_RIDE=1
abp b4_adam_ride_cap64 UNIVL_ADAM_RIDE=1 UNIVL_ADAM_BLOCKS=64
abp b4_adam_sidestream UNIVL_ADAM_RIDE=0
stamp "adam ride done"
t=$(lim 100); [ $t -gt 20 ] && { (cd /tmp && timeout $t rocprofv3 --kernel-trace --stats -d $P/gpurun_out/r03a/prof -o eager --output-format csv -- python $P/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-extras > $P/$OUT/prof_bench.json 2> $P/$OUT/prof_bench.err)
  find gpurun_out/r03a/prof -name "*kernel_stats.csv" -exec cp {} $OUT/eager_kernel_stats.csv \; ; rm -rf gpurun_out/r03a/prof; }
t=$(lim 100); [ $t -gt 20 ] && { (cd /tmp && timeout $t rocprofv3 --kernel-trace --stats -d $P/gpurun_out/r03a/profg -o graph --output-format csv -- python $P/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $P/$OUT/profg_bench.json 2> $P/$OUT/profg_bench.err)
  find gpurun_out/r03a/profg -name "*kernel_stats.csv" -exec cp {} $OUT/graph_kernel_stats.csv \; ; find gpurun_out/r03a/profg -name "*kernel_trace.csv" -exec gzip -c {} \; > $OUT/graph_kernel_trace.csv.gz; rm -rf gpurun_out/r03a/profg; }
stamp "traces done"
t=$(lim 80); [ $t -gt 25 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/gpurun_out/r03a/pmc_fetch --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_fetch.log 2>&1)
t=$(lim 80); [ $t -gt 25 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/gpurun_out/r03a/pmc_write --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_write.log 2>&1)
t=$(lim 80); [ $t -gt 25 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/gpurun_out/r03a/pmc_mfma --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_mfma.log 2>&1)
EL=$(grep -o "[0-9]* flat elements" $OUT/pmc_fetch.log | grep -o "^[0-9]*")
python scripts/pmc_step_parse.py gpurun_out/r03a/pmc_fetch gpurun_out/r03a/pmc_write gpurun_out/r03a/pmc_mfma ${EL:-153784064} 4 $OUT/gemm_pmc.json > $OUT/pmc_parse.log 2>&1
find gpurun_out/r03a/pmc_fetch -name "*counter_collection.csv" -exec gzip -c {} \; > $OUT/pmc_fetch.csv.gz
find gpurun_out/r03a/pmc_write -name "*counter_collection.csv" -exec gzip -c {} \; > $OUT/pmc_write.csv.gz
find gpurun_out/r03a/pmc_mfma -name "*counter_collection.csv" -exec gzip -c {} \; > $OUT/pmc_mfma.csv.gz
rm -rf gpurun_out/r03a/pmc_fetch gpurun_out/r03a/pmc_write gpurun_out/r03a/pmc_mfma
stamp "end"
