#!/bin/bash
# Round 6: the data-parallel supervisor of bench.py (supervise()) on the one-GPU box: world size 1 through the N > 1 code path (--force-dp),
# with injected failures of the first attempts.  Every run must end with exactly ONE JSON line on stdout that carries a value.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 UNIVL_BENCH_SUPERVISE=1
OUT=gpurun_out/r06s
mkdir -p $OUT
A="--force-dp --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-others --watchdog-s 120"
run() { local tag=$1; shift
  env "$@" timeout 600 python bench.py $A > $OUT/$tag.out 2> $OUT/$tag.err; echo "== $tag rc=$? stdout lines: $(wc -l < $OUT/$tag.out)"
  python - $OUT/$tag.out <<'PY'
import json, sys
lines = [l for l in open(sys.argv[1]).read().splitlines() if l.strip().startswith("{")]
d = json.loads(lines[-1])
print("   json lines %d  value %s  ms_per_step %s  graph %s  attempt %s" % (len(lines), d.get("value"), d.get("ms_per_step"),
      (d.get("config") or {}).get("graph_mode"), json.dumps(d.get("dp_attempt"))))
PY
  grep "\[bench\] rank" $OUT/$tag.err | head -5; }
run plain X=1
run crash0 UNIVL_BENCH_INJECT=crash:0
run crash0_hang1 UNIVL_BENCH_INJECT=crash:0,hang:1
unset UNIVL_BENCH_SUPERVISE
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-others --no-extras > $OUT/n1.out 2> $OUT/n1.err; echo "== n1 rc=$? lines $(wc -l < $OUT/n1.out)"; cut -c1-200 $OUT/n1.out
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-others --no-extras > $OUT/torchrun1.out 2> $OUT/torchrun1.err; echo "== torchrun n1 rc=$? lines $(wc -l < $OUT/torchrun1.out)"; cut -c1-160 $OUT/torchrun1.out
