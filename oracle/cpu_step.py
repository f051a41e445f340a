"""TEST INFRASTRUCTURE -- the oracle's training step (CPU port of main_task_retrieval.py:333-353 on the retrieval FT-Joint
configuration).  Used by bench.py's `cpu_baseline` leg on the GPU box (where /root/reference does not exist) and by
oracle/time_reference.py, which times it next to the real reference in the build container."""
import torch

import univl_oracle as O


def make_step(batch_rows, dropout=0.1):
    cfg = O.OracleConfig(batch_size=batch_rows, dropout_prob=dropout)
    P = {k: v.requires_grad_(True) for k, v in O.procedural_params(cfg, 0).items()}
    batch = O.synthetic_batch(cfg, batch_rows, seed=1234, all_ones_mask=True)
    names = list(P)
    groups = O.param_groups(names, lr=3e-5, coef_lr=0.1)
    state = {n: dict(m=torch.zeros_like(P[n]), v=torch.zeros_like(P[n]), step=0) for n in names}

    def step():
        loss = O.univl_forward(P, cfg, batch, training=True)
        loss.backward()
        float(loss)
        with torch.no_grad():
            used = [n for n in names if P[n].grad is not None]
            O.clip_grad_norm_([P[n].grad for n in used], 1.0)
            for n in used:
                st = state[n]
                st["step"] = O.bert_adam_step(P[n], P[n].grad, st["m"], st["v"], st["step"], groups[n]["lr"], 0.1,
                                              100000, groups[n]["weight_decay"])
            for n in names:
                P[n].grad = None
    return step
