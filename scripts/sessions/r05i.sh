#!/bin/bash
# round 5, session i: re-run of the model tests session h failed on, unchanged-loop bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider -k "default_mode or riding or unchanged or schedules_match or (golden and joint_b128 and bf16)" > gpurun_out/r05i_pytest.log 2>&1
echo "pytest exit $?"; tail -n 12 gpurun_out/r05i_pytest.log | cut -c1-300
for extra in "--no-graph" ""; do
  timeout 300 python bench.py --child --steps 50 --warmup 10 $extra 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$extra ms/step', j['ms_per_step'], 'pairs/s', j['value'], j['config'].get('graph_mode'), 'preheat', j.get('preheat',{}).get('block_ms'))"
done
