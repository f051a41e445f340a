#!/bin/bash
# Round 4, session f: LayerNorm-backward rows-per-wave table; update slot A/B at 4 pairs; the full GPU suite in the driver's form.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04f
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
timeout 200 python3 scripts/mb_ln_bwd.py > $OUT/mb_ln_bwd.txt 2>&1; cat $OUT/mb_ln_bwd.txt | tail -6
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 120 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; }
for r in 1 2 3; do
  line b4_slot1_$r "UNIVL_UPDATE_SLOT0=1" --steps 150 --warmup 10
  line b4_slot0_$r "UNIVL_UPDATE_SLOT0=0" --steps 150 --warmup 10
done
stamp "ab done"
timeout 900 python3 -m pytest tests/ -x -q -m gpu --durations=12 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
grep -E "passed|failed|^FAILED|^ERROR|rc=" $OUT/pytest_gpu.log | tail -8
timeout 100 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
stamp "done"
