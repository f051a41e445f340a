#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r02f
mkdir -p $OUT
python -c "
from univl_amd import _lib
L = _lib.lib()
missing = [n for n in _lib.EXPORTED if not hasattr(L, n)]
assert not missing, missing
print('preflight ok')" > $OUT/preflight.txt 2>&1 || { cat $OUT/preflight.txt; exit 7; }
(UNIVL_WGRAD_OFFLOAD=12 timeout 200 python -X faulthandler bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/dbg_graph.json 2> $OUT/dbg_graph.err; echo "rc=$?" >> $OUT/dbg_graph.err)
tail -6 $OUT/dbg_graph.err
if grep -q "rc=0" $OUT/dbg_graph.err; then
B="timeout 300 python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-extras"
run() { name=$1; shift; env "$@" $B > $OUT/$name.json 2> $OUT/$name.err; python - $OUT/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'])
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
}
for r in 1 2 3; do
  run base_$r X=1
  run off12_$r UNIVL_WGRAD_OFFLOAD=12
  run off8_$r UNIVL_WGRAD_OFFLOAD=8
done
(UNIVL_WGRAD_OFFLOAD=12 timeout 400 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "joint_full or joint_small or schedules or accumulation or unchanged" > $OUT/pytest_off.log 2>&1; echo "rc=$?" >> $OUT/pytest_off.log)
tail -4 $OUT/pytest_off.log
fi
