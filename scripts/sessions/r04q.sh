#!/bin/bash
# Round 4, session q: where the fold stops paying (8 / 12 pairs), the other kinds, and the model tests on the fence-free fold.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04q
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 150 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; tail -2 $OUT/bench_$name.err | grep -i -E "error|fail" ; }
for r in 1 2; do
  line b8_fold_$r "UNIVL_LN_FOLD=1" --batch 8 --steps 120 --warmup 10
  line b8_two_$r "UNIVL_LN_FOLD=0" --batch 8 --steps 120 --warmup 10
done
line b12_fold "UNIVL_LN_FOLD=1" --batch 12 --steps 100 --warmup 10
line b12_two "UNIVL_LN_FOLD=0" --batch 12 --steps 100 --warmup 10
line align_fold "UNIVL_LN_FOLD=1" --kind align --steps 60 --warmup 10
line align_two "UNIVL_LN_FOLD=0" --kind align --steps 60 --warmup 10
line cap_fold "UNIVL_LN_FOLD=1" --kind caption --steps 60 --warmup 10
line cap_two "UNIVL_LN_FOLD=0" --kind caption --steps 60 --warmup 10
line pre_fold "UNIVL_LN_FOLD=1" --kind pretrain --batch 6 --steps 40 --warmup 5
line pre_two "UNIVL_LN_FOLD=0" --kind pretrain --batch 6 --steps 40 --warmup 5
stamp "A/B done"
timeout 600 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider -k "atomic_mode or riding or graphed or lazy_word or unchanged_training_loop or dropout_training" > $OUT/pytest_model.log 2>&1; tail -4 $OUT/pytest_model.log; stamp "model tests"
stamp "done"
