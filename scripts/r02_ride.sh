#!/bin/bash
# GPU session (last seconds of the round-2 budget): univl_gemm_pair parity, then ride vs default, then the model-level check.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r02v
mkdir -p $OUT
(timeout 25 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm_pair" > $OUT/pytest_pair.log 2>&1; echo "rc=$?" >> $OUT/pytest_pair.log); tail -4 $OUT/pytest_pair.log
UNIVL_WGRAD_RIDE=1 timeout 20 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras > $OUT/ab_ride.json 2> $OUT/ab_ride.err; grep -o '"ms_per_step": [0-9.]*' $OUT/ab_ride.json; tail -2 $OUT/ab_ride.err
timeout 20 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras > $OUT/ab_base.json 2> $OUT/ab_base.err; grep -o '"ms_per_step": [0-9.]*' $OUT/ab_base.json
(timeout 30 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "riding and joint_full" > $OUT/pytest_ride.log 2>&1; echo "rc=$?" >> $OUT/pytest_ride.log); tail -4 $OUT/pytest_ride.log
