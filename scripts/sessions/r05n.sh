#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for ab in "" "g256_min_rows=768"; do
  UNIVL_AB=$ab timeout 300 python bench.py --child --batch 16 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('[$ab] batch 16 ms/step', j['ms_per_step'], j.get('preheat',{}).get('block_ms'))"
done 2>&1 | tee gpurun_out/r05n_b16.txt
