#!/bin/bash
# round 5, session h: model-level tests (golden cases in deterministic AND default mode, unchanged loop with the riding update), smoke,
# unchanged-loop bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider -k "golden or unchanged or atomic_mode or riding" > gpurun_out/r05h_pytest.log 2>&1
echo "pytest exit $?"; tail -n 40 gpurun_out/r05h_pytest.log | cut -c1-300
cp gpurun_out/parity_errors.json gpurun_out/r05h_parity_errors.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for extra in "--no-graph" ""; do
  timeout 300 python bench.py --child --steps 50 --warmup 10 $extra 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$extra ms/step', j['ms_per_step'], 'pairs/s', j['value'], j['config'].get('graph_mode'), 'preheat', j.get('preheat',{}).get('block_ms'))"
done
