#!/bin/bash
# Round 3, late session 2: the slot-fill tile policy (UNIVL_GEMM_RECT, gemm.hip choose) at 128 pairs -- whole-step A/B, interleaved
# twice -- plus every GEMM kernel test and the 128-pair model tests on the new default.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
BUDGET=${BUDGET:-150}
OUT=gpurun_out/r03v
mkdir -p $OUT
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
for rep in 1 2; do
for v in "UNIVL_GEMM_RECT=0" "UNIVL_GEMM_RECT=2" "UNIVL_GEMM_RECT=1"; do
  t=$(lim 30); [ $t -gt 12 ] || break
  env $v timeout $t python bench.py --batch 128 --steps 60 --warmup 8 --no-cpu-baseline --no-extras > $OUT/bench_b128_${v}_$rep.json 2> $OUT/bench_b128_${v}_$rep.err
  echo "$v rep $rep $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_b128_${v}_$rep.json)" | tee -a $OUT/ab_b128.txt
done
done
stamp "ab done"
t=$(lim 90); [ $t -gt 20 ] && timeout $t python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "gemm or b128" -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
stamp "end"
