#!/bin/bash
# round 5, session e: grouped weight gradients at 1536 / 3072 tokens, and the 32 / 64 / 128-pair steps with and without the 256 body.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python scripts/mb_gemm256.py --rows 1536,3072 --kinds group,fwd --out gpurun_out/r05e_mb.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05e_mb.txt
for b in 128 64 32; do
  for ab in "" "g256=0"; do
    echo "== batch $b  UNIVL_AB=$ab"
    UNIVL_AB=$ab timeout 300 python bench.py --child --batch $b --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(' ms/step', j['ms_per_step'], 'pairs/s', j['value'], 'preheat', j.get('preheat',{}).get('block_ms'), 'loss', j['config'].get('last_loss'))"
  done
done 2>&1 | tee gpurun_out/r05e_steps.txt
