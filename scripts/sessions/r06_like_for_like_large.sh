set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r06_lfl; mkdir -p $OUT
run() { local tag=$1 dir=$2; shift 2
  (cd $dir && timeout 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-others --no-extras "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$tag: /" | tee -a $OUT/like_for_like_large.txt); }
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "operand_pairs or lo_half or qkv_projection_inside or fused_with_operand or fold_matches" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "operand_pairs or riding_in_rectangular or riding_with_the_next" 2>&1 | tail -2
for rep in 1 2 3; do
  run "r05_b32_$rep" _r05 --batch 32
  run "r06_b32_$rep" . --batch 32
  run "r05_b64_$rep" _r05 --batch 64
  run "r06_b64_$rep" . --batch 64
  run "r05_b128_$rep" _r05 --batch 128
  run "r06_b128_$rep" . --batch 128
done
run "r05_b4" _r05
run "r06_b4" .
