#!/bin/bash
# Round 3, session f: lazy word rows (test + A/B), riding update under the captured exchange, b128 wgrad diagnostics.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
BUDGET=${BUDGET:-420}
OUT=gpurun_out/r03f
mkdir -p $OUT
P=$PWD
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
t=$(lim 200)
(timeout $t python -m pytest tests/test_model_gpu.py -m gpu -q --durations=5 -p no:cacheprovider -k "lazy or sparse_word or riding or clip_and_bert or checkpoint_resume or bf16_shadow" > $OUT/pytest_model.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_model.log) &
P1=$!
(timeout $t python -m pytest tests/test_ddp_gpu.py tests/test_kernels_gpu.py -m gpu -q --durations=5 -p no:cacheprovider -k "rccl or rows or adam or optim" > $OUT/pytest_ddp.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_ddp.log) &
P2=$!
wait $P1 $P2
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_model.log | tail -20; grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_ddp.log | tail -10
stamp "pytest done"
ab() {   # name env... -- args
  local name=$1 t; shift
  t=$(lim 60); [ $t -gt 10 ] || return
  env "$@" timeout $t python bench.py --steps $STEPS --warmup 15 --no-cpu-baseline --no-extras $EXTRA > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json) $(grep -o '"graph_mode": "[a-z]*"' $OUT/ab_$name.json) $(grep -o '"last_loss": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt
}
STEPS=200
EXTRA="" ab b4_lazy UNIVL_X=0
EXTRA="" ab b4_nolazy UNIVL_ADAM_LAZY_ROWS=0
EXTRA="" ab b4_lazy2 UNIVL_X=0
EXTRA="" ab b4_nolazy2 UNIVL_ADAM_LAZY_ROWS=0
EXTRA="--force-dp" ab b4_dp_ride UNIVL_X=0
EXTRA="--force-dp" ab b4_dp_ride_dryrun UNIVL_DP_DRYRUN=1
stamp "A/B done"
for v in base groupbig; do
  if [ $v = groupbig ]; then export UNIVL_GEMM_GROUP_BIG_MIN=256; fi
  t=$(lim 80); [ $t -gt 30 ] && { (cd /tmp && timeout $t rocprofv3 --kernel-trace --stats -d $P/$OUT/prof_$v -o b128 --output-format csv -- python $P/bench.py --batch 128 --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-pipeline > $P/$OUT/prof_${v}_bench.json 2> $P/$OUT/prof_${v}_bench.err)
    find $OUT/prof_$v -name "*kernel_stats.csv" -exec cp {} $OUT/b128_${v}_kernel_stats.csv \; ; rm -rf $OUT/prof_$v; head -5 $OUT/b128_${v}_kernel_stats.csv | cut -c1-220; }
done
unset UNIVL_GEMM_GROUP_BIG_MIN
stamp "end"
