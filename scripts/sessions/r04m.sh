#!/bin/bash
# Round 4, session m: (1) the golden parity tests under the tightened bf16 gates + the new "at least as close as the reference's own
# bf16 autocast run" gate; (2) A/B of the two encoder branches on ONE stream (UNIVL_SERIAL_BRANCHES=1) at 128 / 64 / 32 pairs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04m
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 150 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; tail -2 $OUT/bench_$name.err | grep -i -E "error|fail" ; }
for r in 1 2; do
  line b128_serial_$r "UNIVL_SERIAL_BRANCHES=1" --batch 128 --steps 30 --warmup 5
  line b128_forked_$r "UNIVL_SERIAL_BRANCHES=0" --batch 128 --steps 30 --warmup 5
done
line b64_serial "UNIVL_SERIAL_BRANCHES=1" --batch 64 --steps 40 --warmup 5
line b64_forked "UNIVL_SERIAL_BRANCHES=0" --batch 64 --steps 40 --warmup 5
line b32_serial "UNIVL_SERIAL_BRANCHES=1" --batch 32 --steps 60 --warmup 5
line b32_forked "UNIVL_SERIAL_BRANCHES=0" --batch 32 --steps 60 --warmup 5
stamp "A/B done"
timeout 900 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider -k "vs_reference_golden or cotangent_golden" > $OUT/pytest_golden.log 2>&1; tail -5 $OUT/pytest_golden.log; stamp "golden tests"
cp gpurun_out/parity_errors.json $OUT/parity_errors.json 2>/dev/null
stamp "done"
