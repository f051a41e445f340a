#!/bin/bash
# Round 4, session t: sanity of the last host-side change (the folds' arrival counters cleared with the split-K arenas, one launch).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04t
mkdir -p $OUT
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 150 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; tail -2 $OUT/bench_$name.err | grep -i -E "error|fail" ; }
line b4 "A=1" --steps 150 --warmup 10
line align "A=1" --kind align --steps 60 --warmup 10
line pre "A=1" --kind pretrain --batch 6 --steps 40 --warmup 5
timeout 700 python3 -m pytest tests/test_model_gpu.py tests/test_ddp_gpu.py -x -q -m gpu -p no:cacheprovider -k "atomic_mode or riding or graphed or unchanged_training_loop or small-bf16 or ddp or rccl or captured" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
