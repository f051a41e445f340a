#!/bin/bash
# Round 4, session h: riders one launch ahead (prologue = first quarter of a stack's first layer): bit-identity tests, A/B at 4 / 16 pairs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04h
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
timeout 600 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider -k "riding or lazy_word or graphed" > $OUT/pytest_ride.log 2>&1; tail -4 $OUT/pytest_ride.log; stamp "ride tests"
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 120 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; }
for r in 1 2 3; do
  line b4_launch_$r "UNIVL_RIDE_AHEAD=launch" --steps 150 --warmup 10
  line b4_layer_$r "UNIVL_RIDE_AHEAD=layer" --steps 150 --warmup 10
done
for r in 1 2; do
  line b16_launch_$r "UNIVL_RIDE_AHEAD=launch" --batch 16 --steps 100 --warmup 10
  line b16_layer_$r "UNIVL_RIDE_AHEAD=layer" --batch 16 --steps 100 --warmup 10
  line cap_launch_$r "UNIVL_RIDE_AHEAD=launch" --kind caption --steps 60 --warmup 10
  line cap_layer_$r "UNIVL_RIDE_AHEAD=layer" --kind caption --steps 60 --warmup 10
done
stamp "done"
