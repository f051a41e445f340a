#!/bin/bash
# Round 5, session s: optimizer chunks of the last N text layers applied on the video stack's stream (ride_offload) -- sweep at 4 and 16
# pairs, the riding-update identity tests under the override, device stamps of the best candidate.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r05s
mkdir -p $OUT
b() { local tag=$1; shift; local ab=$1; shift
  UNIVL_AB="$ab" timeout 120 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-others --no-extras "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$tag: /" | tee -a $OUT/ab_ride_offload.txt; }
for rep in 1 2; do
  b "off0_$rep" ""
  for n in 3 5 7 9 11; do b "off${n}_$rep" "ride_offload=$n"; done
done
b "off5_cap128" "ride_offload=5,ride_offload_blocks=128"
b "off7_cap256" "ride_offload=7,ride_offload_blocks=256"
b "b16_off0" "" --batch 16
b "b16_off5" "ride_offload=5" --batch 16
b "b16_off8" "ride_offload=8" --batch 16
UNIVL_AB="ride_offload=5" timeout 300 python3 -m pytest tests/test_model_gpu.py -q -x -k "riding or unchanged or pipelined" -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest_offload5.txt
UNIVL_AB="ride_offload=5,stamps=1" timeout 120 python3 scripts/probe_branches.py --batch 4 --steps 60 > $OUT/probe_stamps_off5.txt 2>&1; tail -22 $OUT/probe_stamps_off5.txt
