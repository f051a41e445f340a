"""Debugging aid: run one golden case with NaN-poisoned workspaces (UNIVL_POISON=1) and report where NaNs appear."""
import os, sys
os.environ.setdefault("UNIVL_AB", "poison=1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import univl_oracle as O
from make_golden import case_config
from test_model_gpu import build, call

name = sys.argv[1] if len(sys.argv) > 1 else "align_full"
dtype = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
cfg, rows, dseed = case_config(name)
model, P = build(cfg, dtype)
batch = O.synthetic_batch(cfg, rows, seed=dseed)
model.train()
loss = call(model, batch)
torch.cuda.synchronize()
print("loss", float(loss))


def scan(tag, obj, seen, depth=0):
    if id(obj) in seen or depth > 6:
        return
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        if obj.is_floating_point() and obj.is_cuda and obj.numel() > 0:
            n = int(torch.isnan(obj.float()).sum())
            if n:
                print("  NaN %-60s %8d / %d" % (tag, n, obj.numel()))
        return
    if isinstance(obj, dict):
        for k, v in obj.items():
            scan("%s[%r]" % (tag, k), v, seen, depth + 1)
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            scan("%s[%d]" % (tag, i), v, seen, depth + 1)
    elif hasattr(obj, "__dict__") and type(obj).__module__.startswith("univl_amd") and type(obj).__name__ not in ("UniVL", "FlatParams", "Plan", "Ctx"):
        for k, v in vars(obj).items():
            scan("%s.%s" % (tag, k), v, seen, depth + 1)


st = next(v for v in model._steps.values() if getattr(v, "kind", None) and v.cx.training)
print("--- after forward (NaN = allocated but not written by the forward; backward scratch is expected here)")
scan("step", st, set())
loss.backward()
torch.cuda.synchronize()
print("--- after backward")
scan("step", st, set())
bad = [n for n, p in model.named_parameters() if p.grad is not None and bool(torch.isnan(p.grad).any())]
print("gradients with NaN: %d of %d" % (len(bad), sum(1 for _, p in model.named_parameters() if p.grad is not None)), bad[:12])
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
names = [str(s) for s in g["grad_names"]]
params = dict(model.named_parameters())
rat = []
for i, n in enumerate(names):
    ref = float(g["grad_norms"][i])
    got = float(params[n].grad.double().norm())
    rat.append((got / ref if ref > 0 else float("nan"), n))
rat.sort()
print("norm ratio got/ref: min", rat[:3], "max", rat[-3:], "median", rat[len(rat) // 2])
