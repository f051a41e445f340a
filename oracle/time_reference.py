"""TEST INFRASTRUCTURE -- times the REAL reference training step (modules.modeling.UniVL + modules.optimization.BertAdam,
imported from /root/reference) and the oracle port (oracle/univl_oracle.py, what bench.py's cpu_baseline runs on the GPU
box, where /root/reference does not exist) on the same host cores, same inputs, same thread count, and writes the ratio to
tests/golden/cpu_port_ratio.json.  Build container only.

One step = main_task_retrieval.py:333-353: forward (dropout 0.1 active), backward, float(loss), clip_grad_norm_(1.0),
BertAdam.step, zero_grad; YouCookII retrieval FT-Joint, 12+6 layers, 48x48, all-ones masks, fp32.

    python oracle/time_reference.py [batch=4] [steps=8]
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_harness as H            # noqa: E402
import univl_oracle as O            # noqa: E402
import make_golden as MG            # noqa: E402
import cpu_step                     # noqa: E402


def reference_step_fn(batch_rows):
    cfg = O.OracleConfig(batch_size=batch_rows, dropout_prob=0.1)
    model = H.build_reference_model(MG._task_ns(cfg), vocab_size=cfg.vocab_size, zero_dropout=False)
    MG.load_procedural_into_reference(model, cfg, seed=0)
    model.train()
    batch = O.synthetic_batch(cfg, batch_rows, seed=1234, all_ones_mask=True)
    BertAdam = H.reference_bert_adam()
    groups = O.param_groups([n for n, _ in model.named_parameters()], lr=3e-5, coef_lr=0.1)
    pg = [{"params": [p], "weight_decay": groups[n]["weight_decay"], "lr": groups[n]["lr"]} for n, p in model.named_parameters()]
    opt = BertAdam(pg, lr=3e-5, warmup=0.1, schedule='warmup_linear', t_total=100000, weight_decay=0.01, max_grad_norm=1.0)

    def step():
        loss = MG.reference_forward(model, cfg, batch)
        loss.backward()
        float(loss)
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
    return step


def median_time(step, n):
    step()
    ts = []
    for _ in range(n):
        t0 = time.time()
        step()
        ts.append(time.time() - t0)
    ts.sort()
    return ts[len(ts) // 2]


if __name__ == "__main__":
    assert H.reference_available()
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    nt = os.cpu_count()
    torch.set_num_threads(nt)
    t_ref = median_time(reference_step_fn(rows), n)
    t_port = median_time(cpu_step.make_step(rows), n)
    rec = dict(batch=rows, threads=nt, reference_s_per_step=round(t_ref, 4), port_s_per_step=round(t_port, 4),
               reference_pairs_per_s=round(rows / t_ref, 3), port_pairs_per_s=round(rows / t_port, 3),
               port_over_reference_time=round(t_port / t_ref, 4),
               how="median of %d steps each after one warm-up, %d OpenMP threads, build container (torch %s)" % (n, nt, torch.__version__))
    print(json.dumps(rec))
    json.dump(rec, open(os.path.join(MG.GOLDEN_DIR, "cpu_port_ratio.json"), "w"), indent=1, sort_keys=True)
