#!/bin/bash
# Round 5, session x: LayerNorm-backward fold on the rectangular pair form (384 - 512 tokens): kernel tests, A/B on caption / pretrain / align / 8 + 10 pairs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r05x
mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -q -x -k "pair_ln_fold" -p no:cacheprovider 2>&1 | grep -v "Extension modules" | tail -8 | cut -c1-250 | tee $OUT/pytest_fold.txt
b() { local tag=$1; shift; local ab=$1; shift
  UNIVL_AB="$ab" timeout 150 python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-others --no-extras --no-preheat "$@" 2>$OUT/err_$tag.txt | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$tag: /" | tee -a $OUT/ab_fold_bwd_rect.txt; grep -v "Extension modules\|amdgpu.ids" $OUT/err_$tag.txt | tail -3 | cut -c1-200; }
for rep in 1 2; do
  b "caption_383_$rep" "" --kind caption
  b "caption_512_$rep" "ln_fold_bwd_max=512" --kind caption
  b "pretrain_383_$rep" "" --kind pretrain --batch 6
  b "pretrain_512_$rep" "ln_fold_bwd_max=512" --kind pretrain --batch 6
  b "b8_383_$rep" "" --batch 8
  b "b8_512_$rep" "ln_fold_bwd_max=512" --batch 8
  b "b10_383_$rep" "" --batch 10
  b "b10_512_$rep" "ln_fold_bwd_max=512" --batch 10
done
UNIVL_AB="ln_fold_bwd_max=512" timeout 400 python3 -m pytest tests/test_model_gpu.py -q -x -k "default_mode and (caption or pretrain)" -p no:cacheprovider 2>&1 | grep -v "Extension modules" | tail -5 | cut -c1-250 | tee $OUT/pytest_golden_default.txt
