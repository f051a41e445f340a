"""Which tensors own the global gradient error of a golden case (tests/test_model_gpu.py: gglobal = ||all 256-samples - ref|| / ||ref||)?
    python scripts/diag_gglobal.py joint_full [default]        (UNIVL_AB selects the plan variant)
Prints per tensor: share of the error numerator, share of the reference norm, relative error of its sample.  Test infrastructure."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
from univl_amd import _ab          # noqa: E402
_ab.allow()
import test_model_gpu as T        # noqa: E402
import univl_oracle as O          # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "joint_full"
    default = len(sys.argv) > 2 and sys.argv[2] == "default"
    import univl_amd
    if not default:
        univl_amd.set_deterministic(True)
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    cfg, rows, dseed = T.case_config(name)
    model, P = T.build(cfg, torch.bfloat16)
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    model.train()
    loss = T.call(model, batch)
    loss.backward()
    names = [str(s) for s in g["grad_names"]]
    params = dict(model.named_parameters())
    rows_ = []
    for i, n in enumerate(names):
        gs = T.sample_exact(params[n].grad.float().cpu(), 256).astype(np.float64)
        rs = g["grad_samples"][i][:gs.size].astype(np.float64)
        rows_.append((float(((gs - rs) ** 2).sum()), float((rs ** 2).sum()), n, params[n].numel()))
    num = sum(r[0] for r in rows_)
    den = sum(r[1] for r in rows_)
    print("case %s  %s  UNIVL_AB=%r  gglobal %.3e  loss %.6f (ref %.6f)" % (name, "default" if default else "deterministic", os.environ.get("UNIVL_AB", ""),
                                                                           (num / den) ** 0.5, float(loss), float(g["loss"])))
    print("%-70s %9s %9s %9s %9s" % ("tensor", "err share", "ref share", "rel err", "numel"))
    for e, r, n, k in sorted(rows_, reverse=True)[:25]:
        print("%-70s %9.4f %9.4f %9.2e %9d" % (n, e / num, r / den, (e / max(r, 1e-300)) ** 0.5, k))
    # by kind
    kinds = {}
    for e, r, n, k in rows_:
        kind = ("LayerNorm" if "LayerNorm" in n else "bias" if n.endswith(".bias") else "embedding" if "embeddings" in n else "matrix")
        a = kinds.setdefault(kind, [0.0, 0.0])
        a[0] += e; a[1] += r
    for kind, (e, r) in sorted(kinds.items()):
        print("kind %-10s err share %.4f  ref share %.4f  rel err %.2e" % (kind, e / num, r / den, (e / max(r, 1e-300)) ** 0.5))


if __name__ == "__main__":
    main()
