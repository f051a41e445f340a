#!/bin/bash
# Round 4, session o: (1) attention: bitwise comparison of the new LDS images against the previous build over 200 cases, golden
# caption / FT-Align cases on the new kernels; (2) K8 / K10: the post-product LayerNorm finished inside the product's launch
# (univl_gemm_ln) -- kernel test with race screen, model tests, A/B of the step (UNIVL_LN_FOLD=0|1).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04o
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
BASE=$PWD/univl_amd/lib/libunivl_hip_base.so
timeout 200 python3 scripts/cmp_attention_libs.py 2>&1 | grep -v amdgpu.ids > $OUT/cmp_attention_libs.txt; tail -3 $OUT/cmp_attention_libs.txt; stamp "attention bitwise"
timeout 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -k "gemm_ln" > $OUT/pytest_gemm_ln.log 2>&1; tail -4 $OUT/pytest_gemm_ln.log; stamp "gemm_ln kernel test"
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 150 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; tail -2 $OUT/bench_$name.err | grep -i -E "error|fail" ; }
for r in 1 2; do
  line b4_fold_$r "UNIVL_LN_FOLD=1" --steps 150 --warmup 10
  line b4_two_$r "UNIVL_LN_FOLD=0" --steps 150 --warmup 10
done
line b16_fold "UNIVL_LN_FOLD=1" --batch 16 --steps 100 --warmup 10
line b16_two "UNIVL_LN_FOLD=0" --batch 16 --steps 100 --warmup 10
line cap_fold "UNIVL_LN_FOLD=1" --kind caption --steps 60 --warmup 10
line cap_two "UNIVL_LN_FOLD=0" --kind caption --steps 60 --warmup 10
line cap_two_again "UNIVL_LN_FOLD=0" --kind caption --steps 60 --warmup 10
line cap_base_lib "UNIVL_LN_FOLD=0 UNIVL_LIB=$BASE" --kind caption --steps 60 --warmup 10
line align_fold "UNIVL_LN_FOLD=1" --kind align --steps 60 --warmup 10
line align_two "UNIVL_LN_FOLD=0" --kind align --steps 60 --warmup 10
stamp "A/B done"
timeout 600 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider -k "atomic_mode or riding or graphed or lazy_word or caption_full-bf16 or align_full" > $OUT/pytest_model.log 2>&1; tail -4 $OUT/pytest_model.log; stamp "model tests"
stamp "done"
