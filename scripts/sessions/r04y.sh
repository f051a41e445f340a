#!/bin/bash
# Round 4, session y (last): after the pruning of the backward plan builder -- one quick bench line, then the full GPU suite and smoke().
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04y
mkdir -p $OUT
timeout 100 python3 bench.py --no-cpu-baseline --no-others --no-extras --steps 100 --warmup 10 > $OUT/bench_b4.json 2> $OUT/bench_b4.err
echo "b4: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_b4.json | head -1)" | tee $OUT/summary.txt
timeout 900 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
grep -E "passed|failed|^FAILED|^ERROR|rc=" $OUT/pytest_gpu.log | tail -6 | tee -a $OUT/summary.txt
timeout 60 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log | tee -a $OUT/summary.txt
