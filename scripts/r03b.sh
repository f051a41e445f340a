#!/bin/bash
# Round 3, session b: the GPU suite with the deterministic mode + cotangent goldens (no -x: every failure is wanted), then the
# A/B of the BertAdam update riding with the next forward.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
BUDGET=${BUDGET:-540}
OUT=gpurun_out/r03b
mkdir -p $OUT
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
t=$(lim 420)
(timeout $t python -m pytest tests/test_model_gpu.py -m gpu -q --durations=12 -p no:cacheprovider > $OUT/pytest_model.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_model.log) &
P1=$!
(timeout $t python -m pytest tests -m gpu -q --durations=8 --ignore=tests/test_model_gpu.py -p no:cacheprovider > $OUT/pytest_rest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_rest.log) &
P2=$!
wait $P1 $P2
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_model.log | tail -30; grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_rest.log | tail -10
stamp "pytest done"
ab() {   # name env... -- args
  local name=$1 t; shift
  t=$(lim 60); [ $t -gt 10 ] || return
  env "$@" timeout $t python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-extras $EXTRA > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json) $(grep -o '"last_loss": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt
}
EXTRA="" ab b4_plain UNIVL_X=0
EXTRA="--pipeline" ab b4_adam_ride UNIVL_ADAM_RIDE=1
EXTRA="" ab b4_plain2 UNIVL_X=0
EXTRA="--pipeline" ab b4_adam_ride2 UNIVL_ADAM_RIDE=1
EXTRA="--batch 16" ab b16_plain UNIVL_X=0
EXTRA="--batch 16 --pipeline" ab b16_adam_ride UNIVL_ADAM_RIDE=1
stamp "end"
