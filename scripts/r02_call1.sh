#!/bin/bash
# GPU call 1 of round 2: full GPU test suite (no -x), bench with / without the pipelined optimizer, knob sweeps, rocprof.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r02a
mkdir -p $OUT
rocminfo | grep -m1 "Marketing Name" > $OUT/box.txt 2>&1; nproc >> $OUT/box.txt
# pre-flight: the library that travelled must export everything the binding declares
python -c "
from univl_amd import _lib
L = _lib.lib()
missing = [n for n in _lib.EXPORTED if not hasattr(L, n)]
assert not missing, missing
print('preflight ok')" > $OUT/preflight.txt 2>&1 || { cat $OUT/preflight.txt; exit 7; }
(timeout 120 scripts/mb/burst > $OUT/mb_burst.txt 2>&1)
(timeout 1500 python -m pytest tests -m gpu -q -rs --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
B="timeout 300 python bench.py --steps 100 --warmup 20"
$B > $OUT/bench_pipe.json 2> $OUT/bench_pipe.err
$B --no-pipeline --no-cpu-baseline > $OUT/bench_nopipe.json 2> $OUT/bench_nopipe.err
for nb in 256 512 1024; do
  UNIVL_ADAM_BLOCKS=$nb $B --no-cpu-baseline --no-extras > $OUT/bench_pipe_blocks$nb.json 2> $OUT/bench_pipe_blocks$nb.err
done
for sk in 128 256 768; do
  UNIVL_SPLITK_LEN=$sk $B --no-cpu-baseline --no-extras > $OUT/bench_splitk$sk.json 2> $OUT/bench_splitk$sk.err
done
for sk in 128 256 384 768; do
  UNIVL_GEMM_BURST=1 UNIVL_SPLITK_LEN=$sk $B --no-cpu-baseline > $OUT/bench_burst_splitk$sk.json 2> $OUT/bench_burst_splitk$sk.err
done
(UNIVL_GEMM_BURST=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm or attention or layernorm" > $OUT/pytest_burst.log 2>&1; echo "rc=$?" >> $OUT/pytest_burst.log)
$B --batch 16 --no-cpu-baseline > $OUT/bench_b16.json 2> $OUT/bench_b16.err
$B --batch 16 --no-pipeline --no-cpu-baseline --no-extras > $OUT/bench_b16_nopipe.json 2> $OUT/bench_b16_nopipe.err
timeout 300 python bench.py --steps 50 --warmup 10 --loopback --no-cpu-baseline --no-extras > $OUT/bench_loopback.json 2> $OUT/bench_loopback.err
# kernel trace of the eager step (every kernel its own dispatch)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o eager -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-extras > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof_bench.err)
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/eager_kernel_stats.csv \;
rm -rf $OUT/prof
tail -3 $OUT/pytest.log
cat $OUT/bench_pipe.json | cut -c1-400
cat $OUT/bench_nopipe.json | cut -c1-300
cat $OUT/mb_burst.txt
tail -3 $OUT/pytest_burst.log
