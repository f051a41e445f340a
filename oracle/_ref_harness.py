"""TEST INFRASTRUCTURE ONLY -- imports the real reference (microsoft/UniVL, mounted read-only at
/root/reference) so that golden vectors can be generated from, and the oracle restatement pinned against,
the reference's own classes.  Works only in the build container (the GPU box has no /root/reference).

Nothing in univl_amd/ may import this file.

What it does:
  * stubs `boto3` / `botocore.exceptions` (imported at modules/file_utils.py:20-21, not installed),
  * writes a BERT-base `bert_config.json` directory (values of modules/module_bert.py:61-72) and passes its
    absolute path as `pretrained_bert_name` (until_config.py:42 then resolves it unchanged),
  * builds `modules.modeling.UniVL` through `UniVL.from_pretrained(...)` exactly as
    main_task_retrieval.py:152-166 does, with an argparse.Namespace as task_config.
"""
import argparse
import json
import os
import sys
import tempfile
import types

REFERENCE_ROOT = os.environ.get("UNIVL_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "modules", "modeling.py"))


def _install_stubs():
    if "boto3" not in sys.modules:
        sys.modules["boto3"] = types.ModuleType("boto3")
    if "botocore" not in sys.modules:
        bc = types.ModuleType("botocore")
        bce = types.ModuleType("botocore.exceptions")

        class ClientError(Exception):
            pass

        bce.ClientError = ClientError
        bc.exceptions = bce
        sys.modules["botocore"] = bc
        sys.modules["botocore.exceptions"] = bce
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


_BERT_DIR = None


def bert_config_dir(vocab_size=30522, num_hidden_layers=12):
    """A directory holding bert_config.json with BERT-base values (module_bert.py:61-72)."""
    global _BERT_DIR
    d = tempfile.mkdtemp(prefix="univl_bertcfg_")
    cfg = {
        "attention_probs_dropout_prob": 0.1, "hidden_act": "gelu", "hidden_dropout_prob": 0.1,
        "hidden_size": 768, "initializer_range": 0.02, "intermediate_size": 3072,
        "max_position_embeddings": 512, "num_attention_heads": 12,
        "num_hidden_layers": num_hidden_layers, "type_vocab_size": 2, "vocab_size": vocab_size,
    }
    with open(os.path.join(d, "bert_config.json"), "w") as f:
        json.dump(cfg, f)
    _BERT_DIR = d
    return d


def task_namespace(**kw):
    """The attributes UniVL reads from task_config (SURVEY.md section 5 'Config / flags')."""
    d = dict(max_words=48, max_frames=48, video_dim=1024, batch_size=4, n_gpu=1, n_pair=1, margin=0.1,
             negative_weighting=1, hard_negative_rate=0.5, use_mil=False, do_pretrain=False,
             task_type="retrieval", stage_two=False, train_sim_after_cross=False,
             text_num_hidden_layers=12, visual_num_hidden_layers=6, cross_num_hidden_layers=2,
             decoder_num_hidden_layers=3, local_rank=0)
    d.update(kw)
    return argparse.Namespace(**d)


def build_reference_model(task_config, vocab_size=30522, seed=0, zero_dropout=True):
    """UniVL.from_pretrained as in main_task_retrieval.py:161-162; dropout p forced to 0 for parity."""
    import torch
    _install_stubs()
    from modules.modeling import UniVL  # noqa: the reference's class
    torch.manual_seed(seed)
    model = UniVL.from_pretrained(bert_config_dir(vocab_size), "visual-base", "cross-base", "decoder-base",
                                  cache_dir=None, state_dict=None, task_config=task_config)
    if zero_dropout:
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    return model


def reference_bert_adam():
    _install_stubs()
    from modules.optimization import BertAdam
    return BertAdam
