#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r02e
mkdir -p $OUT
python -c "
from univl_amd import _lib
L = _lib.lib()
missing = [n for n in _lib.EXPORTED if not hasattr(L, n)]
assert not missing, missing
print('preflight ok')" > $OUT/preflight.txt 2>&1 || { cat $OUT/preflight.txt; exit 7; }
(timeout 120 python -X faulthandler scripts/mb_capture_forkjoin.py > $OUT/probe_forkjoin.txt 2>&1; echo "rc=$?" >> $OUT/probe_forkjoin.txt)
tail -8 $OUT/probe_forkjoin.txt
(timeout 400 python -m pytest tests/test_ddp_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "ddp or stock or group or zero_many or reducer or allreduce" > $OUT/pytest_a.log 2>&1; echo "rc=$?" >> $OUT/pytest_a.log)
tail -5 $OUT/pytest_a.log
