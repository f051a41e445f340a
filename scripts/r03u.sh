#!/bin/bash
# Round 3, late session: the per-shape GEMM variant table at 128 pairs (M = 6144 rows; never run on the box before), with the
# half-width tiles (UnivlGemm.tile = 12864 / 64128), the no-store probe on the two candidates, the bit-identity test of the
# half-width tiles, and a whole-step A/B at 128 pairs (UNIVL_GEMM_RECT = 0 / 1 / 2, UNIVL_GEMM_BIG_MIN=300).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
BUDGET=${BUDGET:-170}
OUT=gpurun_out/r03u
mkdir -p $OUT
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
t=$(lim 70); [ $t -gt 20 ] && timeout $t python scripts/mb_gemm_variants.py --rows 6144 --kinds fwd,dgrad --out $OUT/mb_b128.json > $OUT/mb_b128.txt 2>&1
stamp "variants done"; tail -12 $OUT/mb_b128.txt | cut -c1-400
t=$(lim 40); [ $t -gt 15 ] && timeout $t python -m pytest tests/test_kernels_gpu.py -x -q -k half_width -p no:cacheprovider > $OUT/pytest_rect.log 2>&1; tail -3 $OUT/pytest_rect.log
stamp "test done"
for v in "UNIVL_GEMM_RECT=0" "UNIVL_GEMM_RECT=1" "UNIVL_GEMM_RECT=2" "UNIVL_GEMM_BIG_MIN=300"; do
  t=$(lim 30); [ $t -gt 12 ] || break
  env $v timeout $t python bench.py --batch 128 --steps 40 --warmup 8 --no-cpu-baseline --no-extras > $OUT/bench_b128_$v.json 2> $OUT/bench_b128_$v.err
  echo "$v $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_b128_$v.json)" | tee -a $OUT/ab_b128.txt
done
stamp "ab done"
t=$(lim 25); [ $t -gt 10 ] && UNIVL_GEMM_PROBE=1 timeout $t python scripts/mb_gemm_variants.py --rows 6144 --kinds fwd,dgrad --variants 128/2/8,12864/2/8 --check 0 --out $OUT/mb_b128_nostore.json > $OUT/mb_b128_nostore.txt 2>&1
stamp "end"
