"""run_univl_amd.py (the launcher shim of INTEGRATION.md section A) against the REAL, unchanged reference script.
CPU only; needs /root/reference (build container), so it is skipped on the GPU box.  The script is executed up to, not
including, main(): its imports (dataloaders using np.float, file_utils importing boto3), its import-time
init_process_group (re-pointed at gloo here), get_args with the launcher's `--local-rank` spelling, init_model and
prep_optimizer -- everything that does not need a GPU kernel."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("UNIVL_REFERENCE_ROOT", "/root/reference")

CHILD = r'''
import json, os, runpy, sys
sys.path.insert(0, %(root)r)
import run_univl_amd as R
script = R.prepare(os.path.join(%(ref)r, %(script)r))
sys.argv = [script] + R.translate_argv(["--local-rank", "0", "--do_train", "--output_dir", %(out)r, "--bert_model",
                                        "bert-base-uncased", "--batch_size", "4", "--max_words", "48", "--max_frames", "48",
                                        "--lr", "3e-5", "--visual_num_hidden_layers", "2", "--text_num_hidden_layers", "2"])
ns = runpy.run_path(script, run_name="not_main")          # imports + import-time init_process_group + definitions
import torch, univl_amd
assert ns["UniVL"] is univl_amd.UniVL and ns["BertAdam"] is univl_amd.BertAdam
assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "gloo"
import numpy as np
assert np.float is float                                   # dataloaders/dataloader_youcook_retrieval.py:139
import modules.file_utils, modules.tokenization           # the reference's own modules still import (boto3 stubbed if absent)
args = ns["get_args"]()
assert args.local_rank == 0 and args.do_train
args.n_gpu, args.world_size = 1, 1
model = ns["init_model"](args, torch.device("cpu"), 1, 0)  # UniVL.from_pretrained(args.bert_model, ..., task_config=args)
assert isinstance(model, univl_amd.UniVL)
torch.nn.parallel.DistributedDataParallel = lambda m, **kw: m     # the CPU host has no device to wrap for
opt, sched, wrapped = ns["prep_optimizer"](args, model, 100, torch.device("cpu"), 1, 0, coef_lr=args.coef_lr)
assert isinstance(opt, univl_amd.BertAdam) and sched is None
names = dict((id(p), n) for n, p in model.named_parameters())
groups = [[names[id(p)] for p in g["params"]] for g in opt.param_groups]
out = dict(sizes=[len(g) for g in groups], lrs=[g["lr"] for g in opt.param_groups], wds=[g["weight_decay"] for g in opt.param_groups],
           bert_first=all(n.startswith("bert.") for n in groups[0]), nobert=all(not n.startswith("bert.") for n in groups[1]),
           clip_is_shim=torch.nn.utils.clip_grad_norm_.__module__ == "run_univl_amd", keys=len(model.state_dict()))
try:
    model(torch.zeros(1, 1, 48, dtype=torch.long), torch.zeros(1, 1, 48, dtype=torch.long), torch.ones(1, 1, 48, dtype=torch.long),
          torch.zeros(1, 1, 48, 1024, dtype=torch.float64), torch.ones(1, 1, 48, dtype=torch.long))
    out["cpu_forward"] = "ran"
except RuntimeError as e:
    out["cpu_forward"] = str(e)
torch.distributed.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "main_task_retrieval.py")), reason="reference checkout not mounted")
@pytest.mark.parametrize("script", ["main_task_retrieval.py", "main_task_caption.py"])
def test_unchanged_reference_script_through_the_shim(tmp_path, script):
    env = dict(os.environ, UNIVL_SHIM_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    code = CHILD % dict(root=ROOT, ref=REF, script=script, out=str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert len(res["sizes"]) == 4 and all(s > 0 for s in res["sizes"])       # prep_optimizer's four groups (:183-188)
    assert res["bert_first"] and res["nobert"]
    assert res["lrs"][0] == pytest.approx(3e-5 * 0.1) and res["lrs"][1] == pytest.approx(3e-5)
    assert res["wds"] == [0.01, 0.01, 0.0, 0.0]
    assert res["clip_is_shim"]
    assert "no CPU fallback" in res["cpu_forward"] or "HIP device" in res["cpu_forward"]     # the product path fails loudly on CPU
