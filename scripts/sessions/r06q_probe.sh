#!/bin/bash
# Round 6: (1) ceiling of a streaming optimizer beside the forward (scripts/probe_stream_optimizer.py), (2) A/B of the branch-free
# erf-GELU in gemm_tile's bf16 epilogues (product library vs the -DUNIVL_GELU_FAST=0 variant build), (3) the GELU-related tests.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06q
mkdir -p $OUT
timeout 400 python scripts/probe_stream_optimizer.py > $OUT/probe_stream_b4.txt 2>&1; tail -16 $OUT/probe_stream_b4.txt
timeout 400 python scripts/probe_stream_optimizer.py --batch 16 --rounds 2 > $OUT/probe_stream_b16.txt 2>&1; tail -11 $OUT/probe_stream_b16.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gelu or half_width or tile_and_wave" 2>&1 | tail -3
SLOW=$PWD/univl_amd/lib/libunivl_hip_slowgelu.so
for args in "" "--batch 16" "--kind caption"; do
  for r in 1 2 3; do
    for v in fast slow; do
      if [ $v = slow ]; then export UNIVL_LIB=$SLOW; else unset UNIVL_LIB; fi
      ms=$(timeout 300 python bench.py --child --steps 100 --warmup 10 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
      echo "round $r [gelu $v] $args ms/step: $ms" | tee -a $OUT/ab_gelu.txt
    done
  done
done
unset UNIVL_LIB
