#!/bin/bash
# round 5, session k: A/B of the fp32 saved GELU pre-activation: golden gradient errors (deterministic mode) and step time.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for ab in "" "gelu_pre_f32=1"; do
  UNIVL_AB=$ab timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "test_forward_backward_vs_reference_golden and bf16 and (joint_b128 or pretrain_full or joint_full or caption_full) and not default_mode" > gpurun_out/r05k_pytest_${ab:-default}.log 2>&1
  echo "pytest[$ab] exit $?"; tail -n 3 gpurun_out/r05k_pytest_${ab:-default}.log | cut -c1-200
  cp gpurun_out/parity_errors.json gpurun_out/r05k_parity_${ab:-default}.json
  for b in 4 128; do
    UNIVL_AB=$ab timeout 300 python bench.py --child --batch $b --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('[$ab] batch $b ms/step', j['ms_per_step'])"
  done
  UNIVL_AB=$ab timeout 300 python bench.py --child --kind pretrain --batch 6 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('[$ab] pretrain ms/step', j['ms_per_step'])"
done 2>&1 | tee gpurun_out/r05k_ab.txt
python - <<'PY'
import json
a=json.load(open("gpurun_out/r05k_parity_default.json")); b=json.load(open("gpurun_out/r05k_parity_gelu_pre_f32=1.json"))
for k in sorted(a):
    if "bfloat16" in a[k] and k in b:
        x,y=a[k]["bfloat16"],b[k]["bfloat16"]
        print("%-16s"%k, " ".join("%s %.3e->%.3e"%(m,x[m],y[m]) for m in ("gglobal","gmedian","gnorm","gtop")))
PY
