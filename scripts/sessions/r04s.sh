#!/bin/bash
# Round 4, session s: the driver's exact bench command on the final sources (bench.py now reports the bare products as `frac` and the
# as-run launches with their folded LayerNorms as `as_run`); kernel sources unchanged since the closing session's PMC passes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04s
mkdir -p $OUT
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
echo "rc=$?"
python3 - <<'PY' | tee $OUT/summary.txt
import json
j=json.loads([l for l in open("gpurun_out/r04s/bench_driver.json") if l.startswith("{")][-1])
print("headline", j["ms_per_step"], j["value"], "preheat", j["preheat"]["block_ms"], j["preheat"]["stable"])
r=j["roofline"]; print("pcie", j["pcie_inclusive"]["ms_per_step"], "family", r.get("family_ms_per_step"), "frac", r.get("frac"), "traffic/step", r.get("traffic_per_step"), r.get("traffic_source"), "alg", r.get("algorithmic_bytes_per_step"))
print("as_run", r.get("as_run"))
print("adam", r["adam"]["frac"], "step", r["step"])
for o in j.get("other_configs") or []: print(o.get("name"), o.get("ms_per_step"), o.get("value"), o.get("error"), o.get("roofline"))
print("cpu", j["cpu_baseline"] and j["cpu_baseline"].get("value"), j["cpu_baseline"] and j["cpu_baseline"].get("cores"))
PY
