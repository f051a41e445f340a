#!/bin/bash
# Builds and runs the two minimal reproducers on the GPU box; every run under its own timeout.  Output: gpurun_out/r06_repro.txt
OUT=${1:-gpurun_out/r06_repro.txt}
mkdir -p "$(dirname "$OUT")"
{
  hipcc --offload-arch=gfx950 -O2 scripts/repro/capture_pingpong.hip -o /tmp/pingpong || echo "BUILD FAILED pingpong"
  hipcc --offload-arch=gfx950 -O2 scripts/repro/graph_rccl_beside_streaming.hip -o /tmp/grccl -lrccl || echo "BUILD FAILED grccl"
  for mode in global threadlocal relaxed; do
    for alt in 4 64 512; do
      echo "== capture_pingpong alternations=$alt mode=$mode"; timeout 60 /tmp/pingpong $alt $mode; echo "rc=$?"
    done
  done
  for nk in 60 200 400; do
    echo "== graph_rccl_beside_streaming kernels=$nk buckets=7"; HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 120 /tmp/grccl $nk 7; echo "rc=$?"
  done
} > "$OUT" 2>&1
tail -40 "$OUT"
