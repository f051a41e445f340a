"""Debugging aid: the golden parity test's sequence (eval surface, then one training forward/backward) with sentinel bands
around every workspace (UNIVL_GUARD=1); reports the buffers next to which a kernel wrote, and the gradient-norm ratios."""
import os, sys
os.environ.setdefault("UNIVL_AB", "guard=1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import univl_oracle as O
from make_golden import case_config
from test_model_gpu import build, call
from univl_amd import _ab as _uab
_uab.allow()
from univl_amd import engine

name = sys.argv[1] if len(sys.argv) > 1 else "align_full"
dtype = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
do_eval = len(sys.argv) < 4 or sys.argv[3] != "noeval"
cfg, rows, dseed = case_config(name)
model, P = build(cfg, dtype)
batch = O.synthetic_batch(cfg, rows, seed=dseed)
b = {k: v.to("cuda") for k, v in batch.items()}
if do_eval:
    model.eval()
    with torch.no_grad():
        seq, vis = model.get_sequence_visual_output(b["input_ids"], b["token_type_ids"], b["attention_mask"], b["video"], b["video_mask"])
        torch.cuda.synchronize(); print("guards after get_sequence_visual_output:", engine.check_guards())
        sim = model.get_similarity_logits(seq, vis, b["attention_mask"], b["video_mask"])
        torch.cuda.synchronize(); print("guards after get_similarity_logits:", engine.check_guards())
model.train()
loss = call(model, batch)
torch.cuda.synchronize(); print("loss", float(loss), "guards after training forward:", engine.check_guards())
loss.backward()
torch.cuda.synchronize(); print("guards after backward:", engine.check_guards())
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
names = [str(s) for s in g["grad_names"]]
params = dict(model.named_parameters())
gmax = float(np.max(g["grad_norms"]))
rat = sorted((float(params[n].grad.double().norm()) / float(g["grad_norms"][i]), n) for i, n in enumerate(names) if g["grad_norms"][i] > 1e-3 * gmax)
print("significant tensors: norm ratio got/ref min", rat[:2], "median", rat[len(rat) // 2], "max", rat[-2:])
import collections, re
grp = collections.defaultdict(list)
for r, n in rat:
    m = re.match(r"^(\w+)\.encoder\.layer\.(\d+)\.", n)
    key = "%s.L%s" % (m.group(1), m.group(2)) if m else n.split(".")[0] + "." + n.split(".")[1]
    grp[key].append(r)
print("per-group median ratio:", {k: round(float(np.median(v)), 3) for k, v in sorted(grp.items())})
