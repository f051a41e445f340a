#!/bin/bash
# Round 3, session d: captured RCCL exchange after the unique-id fix, cotangent tests with diagnostics, DP bench lines, b128 profile.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
BUDGET=${BUDGET:-420}
OUT=gpurun_out/r03d
mkdir -p $OUT
P=$PWD
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
t=$(lim 200)
(timeout $t python -m pytest tests/test_model_gpu.py -m gpu -q --durations=5 -p no:cacheprovider -k "cotangent or (golden and pretrain)" > $OUT/pytest_model.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_model.log) &
P1=$!
(timeout $t python -m pytest tests/test_ddp_gpu.py -m gpu -q --durations=5 -p no:cacheprovider -k "rccl" > $OUT/pytest_ddp.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_ddp.log) &
P2=$!
wait $P1 $P2
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
grep -E "passed|failed|^FAILED|^ERROR|worst tensor" $OUT/pytest_model.log | tail -40; grep -E "passed|failed|^FAILED|^ERROR|Error|error" $OUT/pytest_ddp.log | tail -10
stamp "pytest done"
ab() {   # name env... -- args
  local name=$1 t; shift
  t=$(lim 70); [ $t -gt 10 ] || return
  env "$@" timeout $t python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-extras $EXTRA > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json) $(grep -o '"graph_mode": "[a-z]*"' $OUT/ab_$name.json) $(grep -o '"exposed_ms": [0-9.]*' $OUT/ab_$name.json) $(grep -o '"algbw_gbs": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt
}
EXTRA="--no-pipeline" ab b4_nopipe UNIVL_X=0
EXTRA="--force-dp" ab b4_dp_captured UNIVL_X=0
EXTRA="--force-dp" ab b4_dp_pg UNIVL_DP_CAPTURE=0
EXTRA="--batch 16 --force-dp" ab b16_dp_captured UNIVL_X=0
EXTRA="--batch 16 --no-pipeline" ab b16_nopipe UNIVL_X=0
tail -3 $OUT/ab_b4_dp_captured.err
stamp "A/B done"
t=$(lim 110); [ $t -gt 30 ] && { (cd /tmp && timeout $t rocprofv3 --kernel-trace --stats -d $P/$OUT/prof128 -o b128 --output-format csv -- python $P/bench.py --batch 128 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $P/$OUT/prof128_bench.json 2> $P/$OUT/prof128_bench.err)
  find $OUT/prof128 -name "*kernel_stats.csv" -exec cp {} $OUT/b128_graph_kernel_stats.csv \; ; rm -rf $OUT/prof128; head -12 $OUT/b128_graph_kernel_stats.csv | cut -c1-200; }
stamp "end"
