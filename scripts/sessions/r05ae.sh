#!/bin/bash
# Round 5, session ae: fused attention FORWARD only at 1536 / 3072 tokens (attn_fuse_fwd_max_rows); slot numbering with the per-product carriers
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r05ae
mkdir -p $OUT
b() { local tag=$1; shift; local ab=$1; shift
  UNIVL_AB="$ab" timeout 150 python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-others --no-extras --no-preheat "$@" 2>$OUT/err_$tag.txt | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$tag: /" | tee -a $OUT/ab_attn_fuse_fwd_1536.txt; grep -v "Extension modules\|amdgpu.ids" $OUT/err_$tag.txt | tail -2 | cut -c1-200; }
for rep in 1 2; do
  b "b32_unfused_$rep" "" --batch 32
  b "b32_fwd_fused_$rep" "attn_fuse_fwd_max_rows=1536" --batch 32
done
b "b64_unfused" "" --batch 64
b "b64_fwd_fused" "attn_fuse_fwd_max_rows=3072" --batch 64
UNIVL_AB="attn_fuse_fwd_max_rows=1536" timeout 300 python3 -m pytest tests/test_model_gpu.py -q -x -k "rectangular_tile or (default_mode and joint_b32)" -p no:cacheprovider 2>&1 | grep -v "Extension modules" | tail -3 | tee $OUT/pytest_b32_fused.txt
