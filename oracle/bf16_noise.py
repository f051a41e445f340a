"""TEST INFRASTRUCTURE -- measures the bf16 noise floor of THE REFERENCE ITSELF, the same way tests/test_model_gpu.py
measures the HIP path: the real reference classes (imported from /root/reference) run once in fp32 and once under
torch.autocast(device_type="cpu", dtype=torch.bfloat16) on the golden cases' weights and inputs, and the differences
(hidden states max-abs, similarity logits max-abs, loss, per-tensor relative gradient error on the golden sample
positions) are written to tests/golden/bf16_autocast_noise.json.  Build container only.

The numbers justify every bf16 gate of the GPU parity tests that sits above north_star's 1e-2: a
gate is never tighter than what torch's own bf16 autocast of the reference achieves against its fp32 self.

    python oracle/bf16_noise.py [case ...]
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_harness as H            # noqa: E402
import univl_oracle as O            # noqa: E402
import make_golden as MG            # noqa: E402

OUT = os.path.join(MG.GOLDEN_DIR, "bf16_autocast_noise.json")
DEFAULT = ["joint_small", "joint_full", "joint_b16", "align_full", "caption_full", "pretrain_full"]


def run(model, cfg, batch, autocast):
    ctx = torch.autocast(device_type="cpu", dtype=torch.bfloat16) if autocast else torch.autocast(device_type="cpu", enabled=False)
    out = {}
    model.eval()
    with torch.no_grad(), ctx:
        seq, vis = model.get_sequence_visual_output(batch["input_ids"], batch["token_type_ids"], batch["attention_mask"],
                                                    batch["video"], batch["video_mask"])
        sim = model.get_similarity_logits(seq, vis, batch["attention_mask"], batch["video_mask"])
        if cfg.has_decoder:
            out["logits"] = MG.sample(model.decoder_caption(seq, vis, batch["input_ids"], batch["attention_mask"],
                                                            batch["video_mask"], batch["input_caption_ids"],
                                                            batch["decoder_mask"], shaped=False, get_logits=True).float())
    out["seq"], out["vis"], out["sim"] = MG.sample(seq.float()), MG.sample(vis.float()), sim.float().numpy().copy()
    model.train()
    model.zero_grad(set_to_none=True)
    with ctx:
        loss = MG.reference_forward(model, cfg, batch)
    loss.backward()
    out["loss"] = float(loss)
    out["grads"] = {n: MG.sample_exact(p.grad, 256) for n, p in model.named_parameters() if p.grad is not None}
    out["gnorm"] = {n: float(p.grad.double().norm()) for n, p in model.named_parameters() if p.grad is not None}
    return out


def measure(name):
    cfg, rows, dseed = MG.case_config(name)
    model = H.build_reference_model(MG._task_ns(cfg), vocab_size=cfg.vocab_size, zero_dropout=True)
    MG.load_procedural_into_reference(model, cfg, seed=0)
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    a, b = run(model, cfg, batch, False), run(model, cfg, batch, True)
    rel, nrm = [], []
    gmax = max(a["gnorm"].values())
    for n, ga in a["grads"].items():
        ref = float(np.linalg.norm(ga))
        if a["gnorm"][n] > 1e-3 * gmax and ref > 0:            # tensors whose gradient is not itself rounding noise
            rel.append(float(np.linalg.norm(b["grads"][n] - ga)) / ref)
            nrm.append(abs(b["gnorm"][n] - a["gnorm"][n]) / a["gnorm"][n])
    rec = dict(hidden_max_abs=float(max(np.abs(a["seq"] - b["seq"]).max(), np.abs(a["vis"] - b["vis"]).max())),
               sim_max_abs=float(np.abs(a["sim"] - b["sim"]).max()), sim_scale=float(np.abs(a["sim"]).max()),
               loss_rel=abs(a["loss"] - b["loss"]) / max(1.0, abs(a["loss"])),
               grad_sample_rel_max=max(rel), grad_sample_rel_median=float(np.median(rel)),
               grad_norm_rel_max=max(nrm), tensors=len(rel))
    if "logits" in a:
        rec["logits_max_abs"] = float(np.abs(a["logits"] - b["logits"]).max())
    print("[bf16-noise] %s: %s" % (name, json.dumps(rec)))
    return rec


if __name__ == "__main__":
    assert H.reference_available()
    torch.set_num_threads(os.cpu_count())
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for nm in (sys.argv[1:] or DEFAULT):
        res[nm] = measure(nm)
    res["_how"] = ("reference (modules.modeling.UniVL, /root/reference) fp32 vs the same model under torch.autocast(cpu, bfloat16); "
                   "dropout 0; per-tensor gradient errors on the 256-element strided samples of make_golden.sample_exact, tensors "
                   "with norm > 1e-3 of the largest only")
    json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
