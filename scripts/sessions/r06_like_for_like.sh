#!/bin/bash
# Round 6: round 5's HEAD (acd2f5e, checked out into _r05 and built there) against this HEAD on the SAME box, alternating, same command
# (`bench.py --steps 20 --warmup 5`, pre-heat on, no side measurements) -- the like-for-like figure box-to-box variance hides.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r06_lfl
mkdir -p $OUT
run() { local tag=$1 dir=$2; shift 2
  (cd $dir && timeout 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-others --no-extras "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$tag: /" | tee -a $OUT/like_for_like.txt); }
for rep in 1 2 3; do
  run "r05_b4_$rep" _r05
  run "r06_b4_$rep" .
done
for rep in 1 2; do
  run "r05_b16_$rep" _r05 --batch 16
  run "r06_b16_$rep" . --batch 16
  run "r05_b8_$rep" _r05 --batch 8
  run "r06_b8_$rep" . --batch 8
  run "r05_caption_$rep" _r05 --kind caption
  run "r06_caption_$rep" . --kind caption
  run "r05_pretrain_$rep" _r05 --kind pretrain --batch 6
  run "r06_pretrain_$rep" . --kind pretrain --batch 6
  run "r05_align_$rep" _r05 --kind align
  run "r06_align_$rep" . --kind align
done
run "r05_b12" _r05 --batch 12
run "r06_b12" . --batch 12
run "r05_b24" _r05 --batch 24
run "r06_b24" . --batch 24
run "r05_b32" _r05 --batch 32
run "r06_b32" . --batch 32
run "r05_b64" _r05 --batch 64
run "r06_b64" . --batch 64
run "r05_b128" _r05 --batch 128
run "r06_b128" . --batch 128
