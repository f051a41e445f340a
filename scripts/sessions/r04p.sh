#!/bin/bash
# Round 4, session p: K8 / K10 fold without agent-scope fences (atomics + s_waitcnt + agent-scope loads): kernel test, the pair alone, the step.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04p
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
timeout 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -k "gemm_ln" > $OUT/pytest_gemm_ln.log 2>&1; tail -4 $OUT/pytest_gemm_ln.log; stamp "gemm_ln kernel test"
timeout 120 python3 scripts/mb_gemm_ln.py 2>&1 | grep -v amdgpu.ids | tee $OUT/mb_gemm_ln.txt; stamp "pair alone"
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 150 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; tail -2 $OUT/bench_$name.err | grep -i -E "error|fail" ; }
for r in 1 2; do
  line b4_fold_$r "UNIVL_LN_FOLD=1" --steps 150 --warmup 10
  line b4_two_$r "UNIVL_LN_FOLD=0" --steps 150 --warmup 10
done
line b16_fold "UNIVL_LN_FOLD=1" --batch 16 --steps 100 --warmup 10
line b16_two "UNIVL_LN_FOLD=0" --batch 16 --steps 100 --warmup 10
stamp "done"
