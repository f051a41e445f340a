#!/bin/bash
# GPU call 3 of round 2: full GPU suite, A/B of background weight gradients / burst GEMMs, kernel trace (csv).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r02b
mkdir -p $OUT
python -c "
from univl_amd import _lib
L = _lib.lib()
missing = [n for n in _lib.EXPORTED if not hasattr(L, n)]
assert not missing, missing
print('preflight ok')" > $OUT/preflight.txt 2>&1 || { cat $OUT/preflight.txt; exit 7; }
(timeout 900 python -m pytest tests -m gpu -q -rs --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
B="timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > $OUT/$name.json 2> $OUT/$name.err; python - $OUT/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], r.get('family_ms_per_step'), (r.get('adam') or {}).get('avg_launch_ms'), (d.get('pcie_inclusive') or {}).get('ms_per_step'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
}
for r in 1 2; do
  run base_$r X=1
  run wg64_$r UNIVL_WGRAD_BLOCKS=64
  run wg128_$r UNIVL_WGRAD_BLOCKS=128
  run wg256_$r UNIVL_WGRAD_BLOCKS=256
  run burst_$r UNIVL_GEMM_BURST=1
  run burst_wg128_$r UNIVL_GEMM_BURST=1 UNIVL_WGRAD_BLOCKS=128
  run norows_$r UNIVL_SPARSE_ROWS=0
done
B="timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --batch 16"
run b16_base X=1
run b16_wg128 UNIVL_WGRAD_BLOCKS=128
(UNIVL_WGRAD_BLOCKS=128 UNIVL_GEMM_BURST=1 timeout 600 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "golden or gemm or schedules" > $OUT/pytest_variants.log 2>&1; echo "rc=$?" >> $OUT/pytest_variants.log)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o eager --output-format csv -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-extras > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof_bench.err)
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/eager_kernel_stats.csv \;
rm -rf $OUT/prof
tail -4 $OUT/pytest.log; tail -3 $OUT/pytest_variants.log
