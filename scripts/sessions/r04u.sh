#!/bin/bash
# Round 4, session u: shares of the riding optimizer chunks per carrying launch now that two of the four end in a latency-bound fold tail.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04u
mkdir -p $OUT
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 150 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; tail -2 $OUT/bench_$name.err | grep -i -E "error|fail" ; }
for r in 1 2; do
  line equal_$r "A=1" --steps 150 --warmup 10
  line w06_14_$r "UNIVL_RIDE_SPLIT=0.6,1.4,0.6,1.4" --steps 150 --warmup 10
  line w03_17_$r "UNIVL_RIDE_SPLIT=0.3,1.7,0.3,1.7" --steps 150 --warmup 10
  line w14_06_$r "UNIVL_RIDE_SPLIT=1.4,0.6,1.4,0.6" --steps 150 --warmup 10
  line w1_1_05_15_$r "UNIVL_RIDE_SPLIT=1,1,0.5,1.5" --steps 150 --warmup 10
done
