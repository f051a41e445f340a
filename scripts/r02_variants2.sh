#!/bin/bash
# GPU session: 8-wave default, L2-blocked tile order (UNIVL_GEMM_GM), grouped weight gradients on the 128 tile -- parity + A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r02o
mkdir -p $OUT
python -c "
from univl_amd import _lib
L = _lib.lib()
missing = [n for n in _lib.EXPORTED if not hasattr(L, n)]
assert not missing, missing
print('preflight ok')" > $OUT/preflight.txt 2>&1 || { cat $OUT/preflight.txt; exit 7; }
(timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm" > $OUT/pytest_gemm.log 2>&1; echo "rc=$?" >> $OUT/pytest_gemm.log); tail -3 $OUT/pytest_gemm.log
(UNIVL_GEMM_GROUP_BIG_MIN=64 timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm" > $OUT/pytest_gemm_groupbig.log 2>&1; echo "rc=$?" >> $OUT/pytest_gemm_groupbig.log); tail -3 $OUT/pytest_gemm_groupbig.log
(timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "golden or clip" > $OUT/pytest_model.log 2>&1; echo "rc=$?" >> $OUT/pytest_model.log); tail -3 $OUT/pytest_model.log
ab() {   # name batch env...
  local name=$1 batch=$2; shift 2
  env "$@" timeout 240 python bench.py --batch $batch --steps 100 --warmup 15 --no-cpu-baseline --no-extras > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json)"
}
ab b4_base 4 UNIVL_X=0
ab b4_w4 4 UNIVL_GEMM_WAVES=4
ab b4_xcd0 4 UNIVL_GEMM_XCD=0
ab b4_base2 4 UNIVL_X=0
ab b4_w4_2 4 UNIVL_GEMM_WAVES=4
ab b16_base 16 UNIVL_X=0
ab b16_w4 16 UNIVL_GEMM_WAVES=4
ab b16_gm0 16 UNIVL_GEMM_GM=0
ab b16_groupbig 16 UNIVL_GEMM_GROUP_BIG_MIN=256
ab b128_base 128 UNIVL_X=0
ab b128_gm0 128 UNIVL_GEMM_GM=0
ab b128_gm4 128 UNIVL_GEMM_GM=4
ab b128_gm16 128 UNIVL_GEMM_GM=16
ab b128_groupbig 128 UNIVL_GEMM_GROUP_BIG_MIN=256
ab b128_groupbig_gm0 128 UNIVL_GEMM_GROUP_BIG_MIN=256 UNIVL_GEMM_GM=0
ab b128_w4 128 UNIVL_GEMM_WAVES=4
UNIVL_GEMM_GM=0 timeout 200 python scripts/mb_gemm_variants.py --rows 6144 --out $OUT/mb_gm0.json > $OUT/mb_gm0.txt 2>&1
UNIVL_GEMM_GM=8 timeout 200 python scripts/mb_gemm_variants.py --rows 6144 --out $OUT/mb_gm8.json > $OUT/mb_gm8.txt 2>&1
tail -13 $OUT/mb_gm8.txt | cut -c1-330
