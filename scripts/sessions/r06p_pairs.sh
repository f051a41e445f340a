#!/bin/bash
# Round 6, operand pairs (UnivlGemm.A_lo / B_lo): kernel tests, the gradient error of the golden cases per pairs variant, and the step cost.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06p
mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "operand_pairs or lo_half or qkv_projection_inside or ln_fold_matches" > $OUT/kernel_tests.txt 2>&1
tail -5 $OUT/kernel_tests.txt
CASES="joint_full or joint_b16 or pretrain_full or caption_full or joint_small"
for v in "xw" "" "x" "w"; do
  UNIVL_AB="pairs=$v" timeout 900 python -m pytest tests/test_model_gpu.py -q -s -m gpu -k "(golden_default_mode or test_forward_backward_vs_reference_golden) and ($CASES) and not float32" 2>&1 | grep -E "^\[parity|passed|failed|Error" > $OUT/parity_pairs_$v.txt
  echo "== pairs=$v"; grep -E "passed|failed" $OUT/parity_pairs_$v.txt; grep -E "parity (joint_full|pretrain_full|joint_b16)" $OUT/parity_pairs_$v.txt | grep -o "^\[parity [^]]*\]\|gglobal=[0-9.e-]*" | paste - - | sort | uniq | head -12
done
bash scripts/ab2.sh $OUT/ab_pairs_b4.txt "pairs=" "pairs=xw" "pairs=x" "pairs=w" "pairs=xw,pairs_ks=terms" > /dev/null 2>&1
BENCH_ARGS="--batch 16" bash scripts/ab2.sh $OUT/ab_pairs_b16.txt "pairs=" "pairs=xw" "pairs=x" "pairs=w" > /dev/null 2>&1
BENCH_ARGS="--batch 8" bash scripts/ab2.sh $OUT/ab_pairs_b8.txt "pairs=" "pairs=xw" > /dev/null 2>&1
cat $OUT/ab_pairs_b4.txt $OUT/ab_pairs_b16.txt $OUT/ab_pairs_b8.txt
