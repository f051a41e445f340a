#!/bin/bash
# Round 4, session w: rocprofv3 kernel trace (per dispatch) of the whole-step graph replay at 4 pairs -- the node list of one step at HEAD.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04w
mkdir -p $OUT
P=$PWD
(cd /tmp && timeout 150 rocprofv3 --kernel-trace -d $P/$OUT/trace --output-format csv -- python3 $P/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-others --no-extras --no-preheat > $P/$OUT/trace.log 2>&1)
find $OUT/trace -name "*kernel_trace.csv" -exec gzip -c {} \; > $OUT/graph_replay_kernel_trace.csv.gz; rm -rf $OUT/trace
ls -la $OUT; tail -2 $OUT/trace.log | cut -c1-300
