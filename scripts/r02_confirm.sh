#!/bin/bash
# GPU session: the tests the copy-node change touches that r02_copy.sh did not run (full-depth goldens, data-parallel, shim).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r02s
mkdir -p $OUT
(timeout 170 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "b16 or b128 or align or caption_full or pretrain_full or resume or schedules or shaped or sparse" > $OUT/pytest_model.log 2>&1; echo "rc=$?" >> $OUT/pytest_model.log) &
P1=$!
(timeout 170 python -m pytest tests/test_ddp_gpu.py tests/test_shim_gpu.py tests/test_eval_gpu.py -m gpu -x -q > $OUT/pytest_ddp.log 2>&1; echo "rc=$?" >> $OUT/pytest_ddp.log) &
P2=$!
wait $P1 $P2
tail -2 $OUT/pytest_model.log; tail -2 $OUT/pytest_ddp.log
