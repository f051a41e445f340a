#!/bin/bash
# Round 3, late session 5: are the bias-gradient workgroups (column-0 tiles walk the staged A tile element by element) what holds
# the big-tile grouped weight gradients back?  Isolated group without fused bias gradients; whole step at 128 pairs with the bias
# gradients on the separate column-sum kernel (UNIVL_DBIAS_COLSUM_MIN) and the group on the 256 x 128 / 128 x 128 tile.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
BUDGET=${BUDGET:-75}
OUT=gpurun_out/r03y2
mkdir -p $OUT
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
t=$(lim 25); timeout $t python scripts/mb_gemm_variants.py --rows "" --group-rows 6144 --group-dbias 0 --out $OUT/mb_group_nodbias.json > $OUT/mb_group_nodbias.txt 2>&1
tail -2 $OUT/mb_group_nodbias.txt | cut -c1-700
stamp "group variants done"
for rep in 1 2; do
for v in "UNIVL_X=0" "UNIVL_DBIAS_COLSUM_MIN=1024 UNIVL_GEMM_GROUP_T256_MINK=1024" "UNIVL_DBIAS_COLSUM_MIN=1024 UNIVL_GEMM_GROUP_BIG_MIN=256"; do
  t=$(lim 25); [ $t -gt 10 ] || break
  n=$(echo "$v" | tr ' =' '__')
  env $v timeout $t python bench.py --batch 128 --steps 50 --warmup 8 --no-cpu-baseline --no-extras > $OUT/bench_b128_${n}_$rep.json 2> $OUT/bench_b128_${n}_$rep.err
  echo "$v rep $rep $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_b128_${n}_$rep.json) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_b128_${n}_$rep.json)" | tee -a $OUT/ab_b128.txt
done
done
stamp "end"
