#!/bin/bash
# Round 5: round 4's HEAD (3448cf2, checked out into _r04tree and built there) against this HEAD on the SAME box, alternating, same command
# (`bench.py --steps 20 --warmup 5`, pre-heat on, no side measurements) -- the like-for-like figure box-to-box variance hides.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r05_lfl
mkdir -p $OUT
run() { local tag=$1 dir=$2; shift 2
  (cd $dir && timeout 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-others --no-extras "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$tag: /" | tee -a $OUT/like_for_like.txt); }
for rep in 1 2 3; do
  run "r04_b4_$rep" _r04tree
  run "r05_b4_$rep" .
done
for rep in 1 2; do
  run "r04_b16_$rep" _r04tree --batch 16
  run "r05_b16_$rep" . --batch 16
done
run "r04_b128" _r04tree --batch 128
run "r05_b128" . --batch 128
run "r04_b64" _r04tree --batch 64
run "r05_b64" . --batch 64
run "r04_caption" _r04tree --kind caption
run "r05_caption" . --kind caption
run "r04_pretrain" _r04tree --kind pretrain --batch 6
run "r05_pretrain" . --kind pretrain --batch 6
