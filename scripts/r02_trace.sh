#!/bin/bash
# Short GPU session: the re-gated bookkeeping test, then an ORDERED kernel trace of a few whole-step graph replays (which
# node sits where -- the ~25 __amd_rocclr_copyBuffer dispatches per step are not in any plan) with the memory-copy trace beside it.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r02q
mkdir -p $OUT
(timeout 120 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "sparse_word_table" > $OUT/pytest_sparse.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_sparse.log); tail -2 $OUT/pytest_sparse.log
P=$PWD
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --memory-copy-trace -d $P/$OUT/tr -o g --output-format csv -- python $P/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $P/$OUT/tr_bench.json 2> $P/$OUT/tr_bench.err)
for f in $(find $OUT/tr -name "*.csv"); do gzip -c $f > $OUT/$(basename $f).gz; done
rm -rf $OUT/tr
ls -la $OUT
