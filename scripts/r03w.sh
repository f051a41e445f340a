#!/bin/bash
# Round 3, late session 3: a layer's grouped weight gradients at 6144 tokens per tile variant (isolated), and the whole step at
# 128 pairs with the group on the 256 x 128 tile (UNIVL_GEMM_GROUP_T256_MINK=1024, ring of 3 / 2 stages) against the default.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
BUDGET=${BUDGET:-110}
OUT=gpurun_out/r03w
mkdir -p $OUT
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
t=$(lim 40); timeout $t python scripts/mb_gemm_variants.py --rows "" --group-rows 6144 --out $OUT/mb_group.json > $OUT/mb_group.txt 2>&1
tail -3 $OUT/mb_group.txt | cut -c1-700
stamp "group variants done"
for rep in 1 2; do
for v in "UNIVL_X=0" "UNIVL_GEMM_GROUP_T256_MINK=1024" "UNIVL_GEMM_GROUP_T256_MINK=1024 UNIVL_GEMM_STAGES256=2"; do
  t=$(lim 30); [ $t -gt 12 ] || break
  n=$(echo "$v" | tr ' =' '__')
  env $v timeout $t python bench.py --batch 128 --steps 60 --warmup 8 --no-cpu-baseline --no-extras > $OUT/bench_b128_${n}_$rep.json 2> $OUT/bench_b128_${n}_$rep.err
  echo "$v rep $rep $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_b128_${n}_$rep.json) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_b128_${n}_$rep.json)" | tee -a $OUT/ab_b128.txt
done
done
stamp "end"
