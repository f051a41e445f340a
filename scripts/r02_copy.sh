#!/bin/bash
# GPU session: copy-node removal (univl_copy_many, aliased loss buffer, gout invariant) -- parity, then same-session A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r02r
mkdir -p $OUT
(timeout 100 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "copy_many or zero_many" > $OUT/pytest_copy.log 2>&1; echo "rc=$?" >> $OUT/pytest_copy.log); tail -2 $OUT/pytest_copy.log
(timeout 200 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "joint_full or joint_small or accumulation or graphed or unchanged or dropout or clip or pretrain_small or caption_small" > $OUT/pytest_model.log 2>&1; echo "rc=$?" >> $OUT/pytest_model.log); tail -2 $OUT/pytest_model.log
(timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log); tail -2 $OUT/smoke.log
ab() {
  local name=$1 batch=$2; shift 2
  env "$@" timeout 60 python bench.py --batch $batch --steps 150 --warmup 15 --no-cpu-baseline --no-extras > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json) $(grep -o '"last_loss": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt
}
ab new1 4 UNIVL_X=0
ab old1 4 UNIVL_COPY_KERNEL=0
ab new2 4 UNIVL_X=0
ab old2 4 UNIVL_COPY_KERNEL=0
