#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider -x -k "attention" > gpurun_out/r05r_pytest.log 2>&1
echo "pytest exit $?"; tail -n 14 gpurun_out/r05r_pytest.log | cut -c1-250
for ab in "" "attn_fuse_fwd=0" "attn_fuse_fwd=0,attn_fuse_bwd=0"; do
  for b in 4 16; do
    UNIVL_AB=$ab timeout 300 python bench.py --child --batch $b --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('[$ab] batch $b ms/step', j['ms_per_step'], j.get('preheat',{}).get('block_ms'), 'loss', j['config']['last_loss'])"
  done
done 2>&1 | tee gpurun_out/r05r_steps.txt
UNIVL_AB= timeout 300 python bench.py --child --steps 20 --warmup 5 --no-graph 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('unchanged loop ms/step', j['ms_per_step'], j.get('preheat',{}).get('block_ms'))" | tee -a gpurun_out/r05r_steps.txt
