#!/bin/bash
# GPU call 5 of round 2: background wgrad on alternating streams, NT Adam default, NT wgrad stores; suite subset incl. sharded.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r02d
mkdir -p $OUT
python -c "
from univl_amd import _lib
L = _lib.lib()
missing = [n for n in _lib.EXPORTED if not hasattr(L, n)]
assert not missing, missing
print('preflight ok')" > $OUT/preflight.txt 2>&1 || { cat $OUT/preflight.txt; exit 7; }
(UNIVL_WGRAD_BLOCKS=128 timeout 200 python -X faulthandler bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/dbg_graph.json 2> $OUT/dbg_graph.err; echo "rc=$?" >> $OUT/dbg_graph.err)
tail -12 $OUT/dbg_graph.err
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
  run nt0_$r UNIVL_ADAM_NT=0
  run wgnt_$r UNIVL_WGRAD_NT=1
  run wg128_$r UNIVL_WGRAD_BLOCKS=128
  run wg64_$r UNIVL_WGRAD_BLOCKS=64
  run wg256_$r UNIVL_WGRAD_BLOCKS=256
  run wg128_nt_$r UNIVL_WGRAD_BLOCKS=128 UNIVL_WGRAD_NT=1
done
(timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ddp_gpu.py -m gpu -q > $OUT/pytest_model.log 2>&1; echo "rc=$?" >> $OUT/pytest_model.log)
tail -6 $OUT/pytest_model.log
(UNIVL_WGRAD_BLOCKS=128 UNIVL_WGRAD_NT=1 timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "golden or schedules or accumulation or unchanged" > $OUT/pytest_wg.log 2>&1; echo "rc=$?" >> $OUT/pytest_wg.log)
tail -4 $OUT/pytest_wg.log
