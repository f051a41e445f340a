#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider -x -k "attention" > gpurun_out/r05q_pytest.log 2>&1
echo "pytest exit $?"; tail -n 4 gpurun_out/r05q_pytest.log | cut -c1-250
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider -k "(golden and (joint_full or align_full or caption_small or pretrain_small or joint_b16)) or riding or unchanged or schedules_match or atomic_mode or reproducible or packaging" > gpurun_out/r05q_pytest_model.log 2>&1
echo "pytest model exit $?"; tail -n 8 gpurun_out/r05q_pytest_model.log | cut -c1-250
