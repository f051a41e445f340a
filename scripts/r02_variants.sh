#!/bin/bash
# GPU session: GEMM variants (three-stage ring, 8-wave workgroups, 256x128 tile) -- parity, per-shape micro-benchmark, in-situ A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r02n
mkdir -p $OUT
python -c "
from univl_amd import _lib
L = _lib.lib()
missing = [n for n in _lib.EXPORTED if not hasattr(L, n)]
assert not missing, missing
print('preflight ok')" > $OUT/preflight.txt 2>&1 || { cat $OUT/preflight.txt; exit 7; }
(timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm" > $OUT/pytest_gemm.log 2>&1; echo "rc=$?" >> $OUT/pytest_gemm.log); tail -5 $OUT/pytest_gemm.log
(timeout 900 python -m pytest tests -m gpu -q -rs --durations=6 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
tail -6 $OUT/pytest.log
timeout 300 python scripts/mb_gemm_variants.py --out $OUT/mb_gemm_variants.json > $OUT/mb_gemm_variants.txt 2>&1; tail -3 $OUT/mb_gemm_variants.txt
ab() {   # name batch env...
  local name=$1 batch=$2; shift 2
  env "$@" timeout 240 python bench.py --batch $batch --steps 100 --warmup 15 --no-cpu-baseline --no-extras > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json)"
}
ab b4_base 4 UNIVL_X=0
ab b4_s3 4 UNIVL_GEMM_STAGES=3
ab b4_w8 4 UNIVL_GEMM_WAVES=8
ab b4_s3w8 4 UNIVL_GEMM_STAGES=3 UNIVL_GEMM_WAVES=8
ab b4_base2 4 UNIVL_X=0
ab b16_base 16 UNIVL_X=0
ab b16_s3w8 16 UNIVL_GEMM_STAGES=3 UNIVL_GEMM_WAVES=8
ab b16_s3 16 UNIVL_GEMM_STAGES=3
ab b128_base 128 UNIVL_X=0
ab b128_t256 128 UNIVL_GEMM_T256_MIN=256
ab b128_t256all 128 UNIVL_GEMM_T256_MIN=100
ab b128_s3w8 128 UNIVL_GEMM_STAGES=3 UNIVL_GEMM_WAVES=8
ab b128_w8 128 UNIVL_GEMM_WAVES=8
P=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $P/$OUT/prof -o b128 --output-format csv -- python $P/bench.py --batch 128 --steps 8 --warmup 3 --no-graph --no-cpu-baseline --no-extras > $P/$OUT/prof_b128.json 2> $P/$OUT/prof_b128.err)
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/b128_eager_kernel_stats.csv \;
rm -rf $OUT/prof
head -12 $OUT/b128_eager_kernel_stats.csv
