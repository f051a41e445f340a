#!/bin/bash
# Round 3, session q: kernel-trace statistics of the whole-step graph replay at HEAD for 4 / 16 / 128 pairs and the caption / pretrain steps.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03q
mkdir -p $OUT
P=$PWD
prof() { local name=$1; shift
  (cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats -d $P/$OUT/prof_$name -o p --output-format csv -- python $P/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras "$@" > $P/$OUT/prof_${name}_bench.json 2> $P/$OUT/prof_${name}_bench.err)
  find $OUT/prof_$name -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_graph_kernel_stats.csv \; ; rm -rf $OUT/prof_$name; echo "== $name"; head -6 $OUT/${name}_graph_kernel_stats.csv | cut -c1-200; }
prof b4
prof b4_eager --no-graph --steps 20 --warmup 5
prof b16 --batch 16
prof b128 --batch 128
prof caption --kind caption
prof pretrain --kind pretrain --batch 6
