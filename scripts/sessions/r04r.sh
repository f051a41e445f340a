#!/bin/bash
# Round 4, session r: the LayerNorm BACKWARD finished inside the pair launch of the dgrad that feeds it (univl_gemm_pair_ln).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04r
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
timeout 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -k "gemm_ln or pair_ln or gemm_pair" > $OUT/pytest_fold.log 2>&1; tail -6 $OUT/pytest_fold.log; stamp "fold kernel tests"
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 150 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; tail -2 $OUT/bench_$name.err | grep -i -E "error|fail" ; }
for r in 1 2 3; do
  line b4_both_$r "UNIVL_LN_FOLD_BWD=1" --steps 150 --warmup 10
  line b4_fwd_$r "UNIVL_LN_FOLD_BWD=0" --steps 150 --warmup 10
done
line b4_none "UNIVL_LN_FOLD=0" --steps 150 --warmup 10
line b6_both "UNIVL_LN_FOLD_BWD=1" --batch 6 --steps 120 --warmup 10
line b6_fwd "UNIVL_LN_FOLD_BWD=0" --batch 6 --steps 120 --warmup 10
line align_both "UNIVL_LN_FOLD_BWD=1" --kind align --steps 60 --warmup 10
line align_fwd "UNIVL_LN_FOLD_BWD=0" --kind align --steps 60 --warmup 10
line pre_both "UNIVL_LN_FOLD_BWD=1" --kind pretrain --batch 6 --steps 40 --warmup 5
line pre_fwd "UNIVL_LN_FOLD_BWD=0" --kind pretrain --batch 6 --steps 40 --warmup 5
stamp "A/B done"
timeout 600 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider -k "atomic_mode or riding or graphed or lazy_word or unchanged_training_loop or weight_gradient_packaging" > $OUT/pytest_model.log 2>&1; tail -4 $OUT/pytest_model.log; stamp "model tests"
stamp "done"
