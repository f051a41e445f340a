#!/bin/bash
# Round 3, closing check: the 16-byte column-sum kernel (bias gradients at thousands of tokens, now issued on the weight-gradient
# stream) -- kernel test, 128-pair parity tests, step time at 128 pairs against the element-per-lane kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03z3
mkdir -p $OUT
timeout 40 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "colsum or b128" -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for v in "UNIVL_COLSUM_VEC=1" "UNIVL_COLSUM_VEC=0"; do
  env $v timeout 20 python bench.py --batch 128 --steps 50 --warmup 8 --no-cpu-baseline --no-extras > $OUT/bench_b128_$v.json 2> $OUT/bench_b128_$v.err
  echo "$v $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_b128_$v.json) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_b128_$v.json)" | tee -a $OUT/ab_b128.txt
done
