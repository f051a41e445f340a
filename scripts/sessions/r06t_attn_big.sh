#!/bin/bash
# Round 6: the fused attention forward for 65 .. 128 positions (attn_fwd_qkv_kernel<NT, BIG>): kernel tests, golden cases that use it, A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06t
mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "qkv_projection_inside or fused_with_operand_pairs" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "(golden) and (caption or pretrain_small or align_full or pretrain_full) and bfloat16 or riding_with_the_next" 2>&1 | tail -4
for cfg in "--kind caption" "--kind pretrain --batch 6" "--kind align" "--kind align --batch 16" "--batch 4"; do
  BENCH_ARGS="$cfg" bash scripts/ab2.sh $OUT/ab_attn_big.txt "" "attn_fuse_fwd_max_seq=64" > /dev/null 2>&1
done
cat $OUT/ab_attn_big.txt
