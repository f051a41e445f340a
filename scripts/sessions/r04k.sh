#!/bin/bash
# Round 4, session k: the riding update carried by the layer's LIGHT launches (attention core, attention-output product, LayerNorms) instead
# of its four forward products: bit-identity tests, kernel tests, A/B (UNIVL_RIDE_ON=light|gemm) at 4 / 16 pairs and the other kinds.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04k
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
timeout 600 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider -k "riding or lazy_word or graphed" > $OUT/pytest_ride.log 2>&1; tail -4 $OUT/pytest_ride.log; stamp "ride tests"
timeout 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -k "attention or layernorm" > $OUT/pytest_kern.log 2>&1; tail -3 $OUT/pytest_kern.log; stamp "kernel tests"
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 120 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; tail -2 $OUT/bench_$name.err | grep -i -E "error|fail" ; }
for r in 1 2 3; do
  line b4_light_$r "UNIVL_RIDE_ON=light" --steps 150 --warmup 10
  line b4_gemm_$r "UNIVL_RIDE_ON=gemm" --steps 150 --warmup 10
done
for r in 1 2; do
  line b16_light_$r "UNIVL_RIDE_ON=light" --batch 16 --steps 100 --warmup 10
  line b16_gemm_$r "UNIVL_RIDE_ON=gemm" --batch 16 --steps 100 --warmup 10
done
line align_light "UNIVL_RIDE_ON=light" --kind align --steps 60 --warmup 10
line align_gemm "UNIVL_RIDE_ON=gemm" --kind align --steps 60 --warmup 10
line cap_light "UNIVL_RIDE_ON=light" --kind caption --steps 60 --warmup 10
line cap_gemm "UNIVL_RIDE_ON=gemm" --kind caption --steps 60 --warmup 10
line b128_light "UNIVL_RIDE_ON=light" --batch 128 --steps 30 --warmup 5
line b128_gemm "UNIVL_RIDE_ON=gemm" --batch 128 --steps 30 --warmup 5
stamp "done"
