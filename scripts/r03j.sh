#!/bin/bash
# Round 3, session j: split-K of the N = 768 products at 16 pairs (144 output tiles: currently unsplit, K loops of up to 24 steps).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03j
mkdir -p $OUT
ab() { local name=$1; shift
  env "$@" timeout 60 python bench.py --steps $STEPS --warmup 15 --no-cpu-baseline --no-extras $EXTRA > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json) $(grep -o '"last_loss": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt; }
STEPS=150
EXTRA="--batch 16" ab b16_base UNIVL_X=0
EXTRA="--batch 16" ab b16_split256 UNIVL_SPLITK_TILES=256
EXTRA="--batch 16" ab b16_split256_wg768 UNIVL_SPLITK_TILES=256 UNIVL_SPLITK_MAXWG=768
EXTRA="--batch 16" ab b16_base2 UNIVL_X=0
EXTRA="--batch 16" ab b16_split256_2 UNIVL_SPLITK_TILES=256
EXTRA="--batch 8" ab b8_base UNIVL_X=0
EXTRA="--batch 8" ab b8_wg768 UNIVL_SPLITK_MAXWG=768
EXTRA="" ab b4_base UNIVL_X=0
EXTRA="" ab b4_wg768 UNIVL_SPLITK_MAXWG=768
