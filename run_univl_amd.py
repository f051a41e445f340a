"""Launcher shim: run an UNCHANGED script of the reference (main_task_retrieval.py, main_task_caption.py, main_pretrain.py)
on top of univl_amd.

    python run_univl_amd.py /path/to/UniVL/main_task_retrieval.py --do_train --bert_model bert-base-uncased ...
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 run_univl_amd.py /path/to/UniVL/main_task_retrieval.py ...

Nothing of the reference is edited or copied.  Before the script runs, this file
  * puts the script's directory (the reference checkout) on sys.path;
  * makes `from modules.modeling import UniVL` / `from modules.optimization import BertAdam` (main_task_retrieval.py:16-17)
    resolve to univl_amd.modeling / univl_amd.optimization, and torch.nn.utils.clip_grad_norm_ (:347) to the fused clip --
    the rest of the reference's `modules` package (tokenization, file_utils) stays the reference's;
  * restores what newer libraries removed under the reference's feet: `np.float` & friends (used by every dataloader, e.g.
    dataloaders/dataloader_youcook_retrieval.py:139; gone since NumPy 1.24), `boto3` / `botocore` imports of
    modules/file_utils.py:20-21 (only needed for S3 downloads) and `nlgeval` of main_task_caption.py:12 when they are not
    installed (stubs that fail only when really used);
  * translates the `--local-rank` flag that torch >= 2.0 launchers pass into the `--local_rank` the scripts declare
    (main_task_retrieval.py:83), and fills it from LOCAL_RANK when the launcher passes neither;
  * supplies single-process rendezvous defaults (RANK=0, WORLD_SIZE=1, MASTER_ADDR=127.0.0.1, a free MASTER_PORT) so that the
    scripts' import-time `torch.distributed.init_process_group(backend="nccl")` (main_task_retrieval.py:23) also works when
    started without a launcher on one GPU.  On ROCm the "nccl" backend is RCCL.
"""
import os
import runpy
import socket
import sys
import types


def install_compat():
    """NumPy aliases and import stubs.  Idempotent."""
    import numpy as np
    for name, typ in (("float", float), ("int", int), ("bool", bool), ("object", object), ("long", int)):
        if name not in np.__dict__:
            setattr(np, name, typ)
    try:
        import boto3  # noqa: F401
    except ImportError:
        b3 = types.ModuleType("boto3")

        def _no_s3(*a, **k):
            raise RuntimeError("boto3 is not installed: s3:// paths are unavailable (run_univl_amd.py stub)")
        b3.resource = b3.client = _no_s3
        sys.modules["boto3"] = b3
    try:
        import botocore.exceptions  # noqa: F401
    except ImportError:
        bc, bce = types.ModuleType("botocore"), types.ModuleType("botocore.exceptions")
        bce.ClientError = type("ClientError", (Exception,), {})
        bc.exceptions = bce
        sys.modules["botocore"], sys.modules["botocore.exceptions"] = bc, bce
    try:
        import nlgeval  # noqa: F401
    except ImportError:
        ng = types.ModuleType("nlgeval")

        class NLGEval:              # main_task_caption.py:12,612: caption metrics need the Java-based package
            def __init__(self, *a, **k):
                pass

            def compute_metrics(self, *a, **k):
                raise RuntimeError("nlgeval is not installed: BLEU/METEOR/ROUGE/CIDEr are unavailable (run_univl_amd.py stub)")
        ng.NLGEval = NLGEval
        sys.modules["nlgeval"] = ng


def install_univl_amd():
    """modules.modeling / modules.optimization -> univl_amd; the fused clip behind torch.nn.utils.clip_grad_norm_."""
    import torch
    from univl_amd import modeling as amd_modeling, optimization as amd_opt
    sys.modules["modules.modeling"] = amd_modeling
    sys.modules["modules.optimization"] = amd_opt
    try:
        import modules                       # the reference's package, if importable: keep attribute access consistent
        modules.modeling, modules.optimization = amd_modeling, amd_opt
    except ImportError:
        pass
    stock_clip = torch.nn.utils.clip_grad_norm_

    def clip_grad_norm_(parameters, max_norm, norm_type=2.0, *a, **k):
        params = [parameters] if isinstance(parameters, torch.Tensor) else list(parameters)
        if params and all(p.is_cuda for p in params) and amd_opt.owns(params[0]):
            return amd_opt.clip_grad_norm_(params, max_norm, norm_type)     # device / kernel errors propagate (no silent second clip)
        return stock_clip(params, max_norm, norm_type, *a, **k)               # not parameters of a univl_amd model
    torch.nn.utils.clip_grad_norm_ = clip_grad_norm_
    return amd_modeling, amd_opt


def translate_argv(argv):
    """--local-rank[=N] -> --local_rank[=N]; add --local_rank from LOCAL_RANK when absent."""
    out, seen = [], False
    for a in argv:
        if a == "--local-rank" or a.startswith("--local-rank="):
            a = "--local_rank" + a[len("--local-rank"):]
        if a == "--local_rank" or a.startswith("--local_rank="):
            seen = True
        out.append(a)
    if not seen and "LOCAL_RANK" in os.environ:
        out += ["--local_rank", os.environ["LOCAL_RANK"]]
    return out


def rendezvous_defaults():
    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
        os.environ.setdefault("LOCAL_RANK", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def prepare(script):
    """Everything up to (not including) running the script.  Returns the absolute script path."""
    script = os.path.abspath(script)
    root = os.path.dirname(script)
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (root, here):
        if p not in sys.path:
            sys.path.insert(0, p)
    install_compat()
    install_univl_amd()
    rendezvous_defaults()
    backend = os.environ.get("UNIVL_SHIM_BACKEND")            # tests on CPU-only hosts: "gloo"
    if backend:
        import torch.distributed as dist
        real = dist.init_process_group

        def init_process_group(*a, **k):
            k.pop("backend", None)
            return real(backend, *a[1:], **k)
        dist.init_process_group = init_process_group
    return script


def main():
    if len(sys.argv) < 2:
        print(__doc__)
        sys.exit(2)
    script = prepare(sys.argv[1])
    sys.argv = [script] + translate_argv(sys.argv[2:])
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
