"""run_univl_amd.py on the GPU box: a script with the reference's call pattern (tests/standin/train_like_reference.py -- the
reference checkout itself does not exist here; tests/test_shim_cpu.py drives the REAL scripts through the same shim in the
build container) runs unmodified through the launcher shim on one MI355X: modules.modeling / modules.optimization are
replaced, boto3 is stubbed, np.float / np.long exist again, `--local-rank` is translated, the import-time
init_process_group("nccl") gets its single-process rendezvous (RCCL, world size 1), the stock DistributedDataParallel
wrap works, and the loss trace equals the direct-API loop on the same data."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STANDIN = os.path.join(ROOT, "tests", "standin")


def test_reference_shaped_script_runs_through_the_shim(tmp_path):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "UNIVL_SHIM_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "run_univl_amd.py"), os.path.join(STANDIN, "train_like_reference.py"),
           "--local-rank", "0", "--output_dir", str(tmp_path), "--steps", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    tr = json.load(open(os.path.join(tmp_path, "trace.json")))
    assert tr["backend"] == "nccl" and tr["world"] == 1 and tr["model_class"] == "univl_amd.modeling"
    # the same loop through the direct API (no shim, no DDP wrap) on the same seeds
    sys.path.insert(0, STANDIN)
    sys.path.insert(0, ROOT)
    import run_univl_amd
    run_univl_amd.install_compat()                          # np.float / np.long, as the shim does for the script
    from synthetic_data import Synthetic
    import argparse
    from univl_amd import UniVL, BertAdam, clip_grad_norm_
    a = argparse.Namespace(local_rank=0, steps=4, batch_size=4, lr=1e-4, coef_lr=0.1, max_words=20, max_frames=12, video_dim=1024,
                           n_gpu=1, n_pair=1, margin=0.1, negative_weighting=1, hard_negative_rate=0.5, use_mil=False,
                           do_pretrain=False, task_type="retrieval", stage_two=False, train_sim_after_cross=False,
                           text_num_hidden_layers=2, visual_num_hidden_layers=1, cross_num_hidden_layers=1,
                           decoder_num_hidden_layers=1, dropout_prob=0.0, compute_dtype="fp32", seed=7)
    torch.manual_seed(a.seed)
    model = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base", task_config=a)
    model.to("cuda").train()
    named = list(model.named_parameters())
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    nd = [(n, p) for n, p in named if not any(x in n for x in no_decay)]
    dc = [(n, p) for n, p in named if any(x in n for x in no_decay)]
    groups = [{"params": [p for n, p in nd if "bert." in n], "weight_decay": 0.01, "lr": a.lr * a.coef_lr},
              {"params": [p for n, p in nd if "bert." not in n], "weight_decay": 0.01},
              {"params": [p for n, p in dc if "bert." in n], "weight_decay": 0.0, "lr": a.lr * a.coef_lr},
              {"params": [p for n, p in dc if "bert." not in n], "weight_decay": 0.0}]
    opt = BertAdam(groups, lr=a.lr, warmup=0.1, schedule="warmup_linear", t_total=a.steps * 2, weight_decay=0.01, max_grad_norm=1.0)
    data = Synthetic(a.batch_size * a.steps, a.max_words, a.max_frames, a.video_dim, seed=a.seed)
    losses = []
    for s in range(a.steps):
        items = [data[i] for i in range(s * a.batch_size, (s + 1) * a.batch_size)]
        cols = [torch.from_numpy(np.stack([it[c] for it in items])).to("cuda") for c in range(9)]
        ids, mask, seg, video, vmask, mtext, tlab, mvideo, vlab = cols
        loss = model(ids, seg, mask, video, vmask, pairs_masked_text=mtext, pairs_token_labels=tlab, masked_video=mvideo,
                     video_labels_index=vlab)
        loss.backward()
        losses.append(float(loss))
        clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
    np.testing.assert_allclose(tr["losses"], losses, rtol=2e-4, atol=2e-5)
    assert tr["lrs"][-1] == pytest.approx(sorted(set(opt.get_lr())))
    sd = torch.load(os.path.join(tmp_path, "pytorch_model.bin.0"), map_location="cpu")
    assert set(sd.keys()) == set(model.state_dict().keys())                 # checkpoint written by the script's own save path
    n = "bert.encoder.layer.1.output.dense.weight"
    assert float((sd[n] - model.state_dict()[n].cpu()).abs().max()) < 1e-5
