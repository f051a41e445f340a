#!/bin/bash
# Round 3, session g: lazy-rows test after the chunking fix, the DP + riding-update crash with faulthandler, bias gradients through
# the column-sum kernel at 128 pairs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
BUDGET=${BUDGET:-300}
OUT=gpurun_out/r03g
mkdir -p $OUT
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
t=$(lim 120)
(timeout $t python -m pytest tests/test_model_gpu.py -m gpu -q --durations=5 -p no:cacheprovider -k "lazy or (golden and joint_b128)" > $OUT/pytest_model.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_model.log) &
P1=$!
(timeout 100 python -X faulthandler bench.py --force-dp --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/dp_ride.json 2> $OUT/dp_ride.err; echo "rc=$?" >> $OUT/dp_ride.err) &
P2=$!
wait $P1 $P2
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_model.log | tail -10
grep -v "amdgpu.ids\|socket.cpp" $OUT/dp_ride.err | tail -40
stamp "tests done"
ab() {   # name env... -- args
  local name=$1 t; shift
  t=$(lim 60); [ $t -gt 10 ] || return
  env "$@" timeout $t python bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-extras $EXTRA > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json) $(grep -o '"last_loss": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt
}
STEPS=60
EXTRA="--batch 128" ab b128_colsum UNIVL_X=0
EXTRA="--batch 128" ab b128_ingemm UNIVL_DBIAS_COLSUM_MIN=100000
EXTRA="--batch 128" ab b128_colsum2 UNIVL_X=0
EXTRA="--batch 16" ab b16_default UNIVL_X=0
stamp "end"
