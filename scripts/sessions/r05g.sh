#!/bin/bash
# round 5, session g: 16 / 32 / 64 / 128-pair steps, default plans vs g256=0.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 32 64 128; do
  for ab in "" "g256=0"; do
    echo "== batch $b  UNIVL_AB=$ab"
    UNIVL_AB=$ab timeout 300 python bench.py --child --batch $b --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(' ms/step', j['ms_per_step'], 'pairs/s', j['value'], 'preheat', j.get('preheat',{}).get('block_ms'), 'loss', j['config'].get('last_loss'))"
  done
done 2>&1 | tee gpurun_out/r05g_steps.txt
(cd /tmp && rm -rf /tmp/prof64 && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof64 -o r --output-format csv -- python "$OLDPWD/bench.py" --child --batch 64 --steps 10 --warmup 3 --no-graph > /tmp/prof64.log 2>&1; echo "rocprof exit $?")
f=$(find /tmp/prof64 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05g_bench_b64_kernel_stats.csv; head -14 "$f" | cut -c1-200
