#!/bin/bash
# Round 4, session v: the full GPU suite and smoke() at the final HEAD (as the driver runs them: one process).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04v
mkdir -p $OUT
timeout 900 python3 -m pytest tests/ -x -q -m gpu --durations=12 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
grep -E "passed|failed|^FAILED|^ERROR|rc=" $OUT/pytest_gpu.log | tail -6
timeout 100 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
