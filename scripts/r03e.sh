#!/bin/bash
# Round 3, session e: A/B of existing switches at 128 pairs (grouped weight gradients on the 128 tile, 3-stage ring, LayerNorm rows
# per wave) and of the data-parallel schedule at 4 pairs (dry-run exchange, fewer exchange points).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
BUDGET=${BUDGET:-420}
OUT=gpurun_out/r03e
mkdir -p $OUT
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
ab() {   # name env... -- args
  local name=$1 t; shift
  t=$(lim 60); [ $t -gt 10 ] || return
  env "$@" timeout $t python bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-extras $EXTRA > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json) $(grep -o '"graph_mode": "[a-z]*"' $OUT/ab_$name.json) $(grep -o '"last_loss": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt
}
STEPS=60
EXTRA="--batch 128" ab b128_base UNIVL_X=0
EXTRA="--batch 128" ab b128_groupbig UNIVL_GEMM_GROUP_BIG_MIN=256
EXTRA="--batch 128" ab b128_stages3 UNIVL_GEMM_STAGES=3
EXTRA="--batch 128" ab b128_groupbig_rpw1 UNIVL_GEMM_GROUP_BIG_MIN=256 UNIVL_LN_RPW=1
EXTRA="--batch 128" ab b128_groupbig_rpw2 UNIVL_GEMM_GROUP_BIG_MIN=256 UNIVL_LN_RPW=2
EXTRA="--batch 128" ab b128_base2 UNIVL_X=0
stamp "b128 done"
STEPS=150
EXTRA="--force-dp" ab b4_dp_dryrun UNIVL_DP_DRYRUN=1
EXTRA="--force-dp" ab b4_dp_bucket200 UNIVL_BUCKET_MB=200
EXTRA="--force-dp" ab b4_dp_dryrun_bucket200 UNIVL_DP_DRYRUN=1 UNIVL_BUCKET_MB=200
EXTRA="--force-dp" ab b4_dp_captured UNIVL_X=0
EXTRA="--no-pipeline" ab b4_nopipe UNIVL_X=0
stamp "end"
