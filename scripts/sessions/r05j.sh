#!/bin/bash
# round 5, session j: asm transpose reads in gemm_tile (every dgrad / weight-gradient product): kernel tests, steps at 4 .. 128 pairs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider -x -k "gemm or fold" > gpurun_out/r05j_pytest.log 2>&1
echo "pytest exit $?"; tail -n 6 gpurun_out/r05j_pytest.log | cut -c1-300
for b in 4 16 32 64 128; do
  timeout 300 python bench.py --child --batch $b --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('batch $b ms/step', j['ms_per_step'], 'pairs/s', j['value'], 'preheat', j.get('preheat',{}).get('block_ms'))"
done 2>&1 | tee gpurun_out/r05j_steps.txt
for k in align caption; do
  timeout 300 python bench.py --child --kind $k --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$k ms/step', j['ms_per_step'], 'preheat', j.get('preheat',{}).get('block_ms'))"
done 2>&1 | tee -a gpurun_out/r05j_steps.txt
timeout 300 python bench.py --child --kind pretrain --batch 6 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('pretrain ms/step', j['ms_per_step'], 'preheat', j.get('preheat',{}).get('block_ms'))" | tee -a gpurun_out/r05j_steps.txt
