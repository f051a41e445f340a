#!/bin/bash
# GPU call 4 of round 2: find the background-wgrad crash (faulthandler), A/B of early loss read-back and non-temporal Adam.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r02c
mkdir -p $OUT
python -c "
from univl_amd import _lib
L = _lib.lib()
missing = [n for n in _lib.EXPORTED if not hasattr(L, n)]
assert not missing, missing
print('preflight ok')" > $OUT/preflight.txt 2>&1 || { cat $OUT/preflight.txt; exit 7; }
(UNIVL_WGRAD_BLOCKS=128 UNIVL_AUTO_GRAPH=0 timeout 200 python -X faulthandler bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-extras > $OUT/dbg_eager.json 2> $OUT/dbg_eager.err; echo "rc=$?" >> $OUT/dbg_eager.err)
(UNIVL_WGRAD_BLOCKS=128 timeout 200 python -X faulthandler bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/dbg_graph.json 2> $OUT/dbg_graph.err; echo "rc=$?" >> $OUT/dbg_graph.err)
tail -25 $OUT/dbg_eager.err; tail -25 $OUT/dbg_graph.err
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
  run sync_$r UNIVL_ASYNC_LOSS=0
  run async_$r UNIVL_ASYNC_LOSS=1
  run async_nt_$r UNIVL_ASYNC_LOSS=1 UNIVL_ADAM_NT=1
  run async_wg128_$r UNIVL_ASYNC_LOSS=1 UNIVL_WGRAD_BLOCKS=128
  run async_wg64_$r UNIVL_ASYNC_LOSS=1 UNIVL_WGRAD_BLOCKS=64
done
(timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "golden or schedules or unchanged" > $OUT/pytest_model.log 2>&1; echo "rc=$?" >> $OUT/pytest_model.log)
tail -4 $OUT/pytest_model.log
