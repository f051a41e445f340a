#!/bin/bash
# Round 4, session n: attention LDS images (one pitch per read direction, two images of the two-way matrices in the backward):
# kernel tests, the kernels alone per problem size for four builds / settings, the step with the old and the new library.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04n
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
BASE=$PWD/univl_amd/lib/libunivl_hip_base.so
TRACE=$PWD/univl_amd/lib/libunivl_hip_trace.so
timeout 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -k "attention" > $OUT/pytest_attention.log 2>&1; tail -3 $OUT/pytest_attention.log; stamp "attention tests"
{
  echo "== base (round-4 HEAD before this change: one 160-byte pitch)"; UNIVL_LIB=$BASE timeout 120 python3 scripts/mb_attention.py 2>&1 | grep -v amdgpu.ids
  echo "== new (product build: PK / PT images, two images up to 64 positions)"; timeout 120 python3 scripts/mb_attention.py 2>&1 | grep -v amdgpu.ids
  echo "== new, no two-image staging"; UNIVL_LIB=$TRACE UNIVL_ATTN_DUAL_MAX=0 timeout 120 python3 scripts/mb_attention.py 2>&1 | grep -v amdgpu.ids
  echo "== new, two images up to 128 positions"; UNIVL_LIB=$TRACE UNIVL_ATTN_DUAL_MAX=128 timeout 120 python3 scripts/mb_attention.py 2>&1 | grep -v amdgpu.ids
  echo "== new, two images up to 224 positions"; UNIVL_LIB=$TRACE UNIVL_ATTN_DUAL_MAX=224 timeout 120 python3 scripts/mb_attention.py 2>&1 | grep -v amdgpu.ids
} > $OUT/mb_attention.txt 2>&1
stamp "kernels alone"
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 150 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; tail -2 $OUT/bench_$name.err | grep -i -E "error|fail" ; }
for r in 1 2; do
  line b4_base_$r "UNIVL_LIB=$BASE" --steps 150 --warmup 10
  line b4_new_$r "A=1" --steps 150 --warmup 10
done
line b128_base "UNIVL_LIB=$BASE" --batch 128 --steps 30 --warmup 5
line b128_new "A=1" --batch 128 --steps 30 --warmup 5
line cap_base "UNIVL_LIB=$BASE" --kind caption --steps 60 --warmup 10
line cap_new "A=1" --kind caption --steps 60 --warmup 10
line align_base "UNIVL_LIB=$BASE" --kind align --steps 60 --warmup 10
line align_new "A=1" --kind align --steps 60 --warmup 10
stamp "done"
