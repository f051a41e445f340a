#!/bin/bash
# Round 4, session a: (1) the driver's exact bench command with the declared pre-heat + other_configs; (2) the same without pre-heat
# (shows the transient BENCH_r03 sat on); (3) branch-overlap probes of the captured step WITHOUT a profiler (stamps, skip probes).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04a
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
(rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -6) > $OUT/clocks_before.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
stamp "driver-form bench rc=$?"
(rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -6) > $OUT/clocks_after.txt
python3 - <<'PY' | tee -a $OUT/summary.txt
import json
j=json.loads([l for l in open("gpurun_out/r04a/bench_driver.json") if l.startswith("{")][-1])
print("headline", j["ms_per_step"], j["value"], "preheat", j["preheat"])
print("pcie", j["pcie_inclusive"] and j["pcie_inclusive"]["ms_per_step"], "family", j["roofline"].get("family_ms_per_step"), j["roofline"].get("frac"))
for r in j.get("other_configs") or []: print(r.get("name"), r.get("ms_per_step"), r.get("value"), r.get("error"), r.get("roofline"), r.get("child_wall_s"))
print("cpu", j["cpu_baseline"] and j["cpu_baseline"].get("value"))
PY
for i in 1 2; do
  timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-preheat --no-others --no-cpu-baseline --no-extras > $OUT/bench_nopreheat_$i.json 2>> $OUT/bench_nopreheat.err
  echo "no-preheat $i: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_nopreheat_$i.json | head -1)" | tee -a $OUT/summary.txt
done
stamp "no-preheat done"
UNIVL_STAMPS=1 timeout 200 python3 scripts/probe_branches.py > $OUT/probe_stamps_b4.txt 2>&1; tail -22 $OUT/probe_stamps_b4.txt
for sk in none visual bert; do
  env UNIVL_PROBE_SKIP=$sk timeout 200 python3 scripts/probe_branches.py > $OUT/probe_skip_${sk}_b4.txt 2>&1; grep -E "^batch|^wall" $OUT/probe_skip_${sk}_b4.txt | tee -a $OUT/summary.txt
done
stamp "b4 probes done"
UNIVL_STAMPS=1 timeout 200 python3 scripts/probe_branches.py --batch 16 > $OUT/probe_stamps_b16.txt 2>&1; tail -22 $OUT/probe_stamps_b16.txt
for sk in none visual; do
  env UNIVL_PROBE_SKIP=$sk timeout 200 python3 scripts/probe_branches.py --batch 16 > $OUT/probe_skip_${sk}_b16.txt 2>&1; grep -E "^batch|^wall" $OUT/probe_skip_${sk}_b16.txt | tee -a $OUT/summary.txt
done
stamp "done"
