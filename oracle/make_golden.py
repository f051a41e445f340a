"""TEST INFRASTRUCTURE -- generates tests/golden/*.npz by EXECUTING THE REAL REFERENCE (modules.modeling.UniVL,
modules.optimization.BertAdam imported from /root/reference) on the procedural parameters and synthetic inputs
defined in oracle/univl_oracle.py.  Run in the build container only:

    python oracle/make_golden.py            # all cases
    python oracle/make_golden.py joint_full # one case

The fixtures are small (sampled tensors, per-parameter gradient norms + leading elements); parameters and
inputs are re-derived from seeds by the tests, so no state_dict is shipped.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_harness as H          # noqa: E402
import univl_oracle as O          # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")

# name -> (OracleConfig kwargs, rows, data seed)
CASES = {
    # cfg1/cfg2 of BASELINE.json: YouCookII retrieval FT-Joint, 12+6 layers, 48x48, bs=4
    "joint_full": (dict(batch_size=4), 4, 1234),
    # same path, 2+1 layers, ragged non-multiple-of-16 lengths
    "joint_small": (dict(batch_size=3, text_num_hidden_layers=2, visual_num_hidden_layers=1,
                         max_words=20, max_frames=12), 3, 7),
    # all-ones masks (throughput shape) on a shallow stack
    "joint_ones": (dict(batch_size=4, text_num_hidden_layers=1, visual_num_hidden_layers=1), 4, 11),
    # FT-Align: --train_sim_after_cross (B^2 pairs through the 2-layer cross encoder, chunks of 5 text rows)
    "align_small": (dict(batch_size=6, text_num_hidden_layers=1, visual_num_hidden_layers=1,
                         cross_num_hidden_layers=2, train_sim_after_cross=True,
                         max_words=16, max_frames=16), 6, 21),
    # caption stage-two: cross + causal decoder + tied vocab classifier
    "caption_small": (dict(batch_size=2, text_num_hidden_layers=1, visual_num_hidden_layers=1,
                           cross_num_hidden_layers=1, decoder_num_hidden_layers=2, stage_two=True,
                           task_type="caption", max_words=24, max_frames=16), 2, 31),
    # pretrain stage-two: 5 losses, MIL-NCE, n_pair=3
    "pretrain_small": (dict(batch_size=2, n_pair=3, text_num_hidden_layers=1, visual_num_hidden_layers=1,
                            cross_num_hidden_layers=1, decoder_num_hidden_layers=1, stage_two=True,
                            do_pretrain=True, use_mil=True, task_type="retrieval",
                            max_words=16, max_frames=16), 2, 41),
    # ---- BASELINE.json configurations at full depth (12 + 6 (+ 2 cross, + 3 decoder) layers)
    # cfg3 (MSRVTT retrieval, global bs 128): 16 rows per GPU on 8 GPUs, 128 rows on one GPU
    "joint_b16": (dict(batch_size=16), 16, 1316),
    "joint_b128": (dict(batch_size=128), 128, 13128),
    # ... and its 4- / 2-GPU shares: 32 / 64 rows per GPU (1536 / 3072 tokens: where the plans change to the 256 x 256 GEMM body, round 5)
    "joint_b32": (dict(batch_size=32), 32, 1332),
    "joint_b64": (dict(batch_size=64), 64, 1364),
    # FT-Align (--train_sim_after_cross) at 48x48: 16 (text, video) pairs x 96 tokens through the 2-layer cross encoder
    # (data seed picked so that no hinge argument of the 4x4 max-margin loss, margin + s_ij - s_ii, is closer than 0.07 to
    #  zero: with seed 2104 one sits at 8.5e-4, inside the bf16 noise of the logits (2e-3), and whether that hinge counts --
    #  i.e. an O(30 %) step in every gradient of this 16-pair batch -- is a coin flip of the rounding in ANY bf16
    #  implementation; the loss is not differentiable there, so no gradient parity is defined)
    "align_full": (dict(batch_size=4, train_sim_after_cross=True), 4, 2105),
    # cfg4: caption finetune stage two, 128 x 96, 3 decoder layers, 4 rows per GPU
    "caption_full": (dict(batch_size=4, stage_two=True, task_type="caption", max_words=128, max_frames=96), 4, 3104),
    # cfg5: pretrain stage two, 2 videos x n_pair 3 = 6 rows, 48 x 64, five losses
    "pretrain_full": (dict(batch_size=2, n_pair=3, stage_two=True, do_pretrain=True, use_mil=True,
                           task_type="retrieval", max_words=48, max_frames=64), 2, 4106),
}
FULL_CASES = ["joint_b16", "joint_b32", "joint_b64", "joint_b128", "align_full", "caption_full", "pretrain_full"]


def case_config(name):
    kw, rows, seed = CASES[name]
    return O.OracleConfig(**kw), rows, seed


def _task_ns(cfg):
    return H.task_namespace(
        max_words=cfg.max_words, max_frames=cfg.max_frames, video_dim=cfg.video_dim, batch_size=cfg.batch_size,
        n_gpu=cfg.n_gpu, n_pair=cfg.n_pair, margin=cfg.margin, negative_weighting=cfg.negative_weighting,
        hard_negative_rate=cfg.hard_negative_rate, use_mil=cfg.use_mil, do_pretrain=cfg.do_pretrain,
        task_type=cfg.task_type, stage_two=cfg.stage_two, train_sim_after_cross=cfg.train_sim_after_cross,
        text_num_hidden_layers=cfg.text_num_hidden_layers, visual_num_hidden_layers=cfg.visual_num_hidden_layers,
        cross_num_hidden_layers=cfg.cross_num_hidden_layers,
        decoder_num_hidden_layers=cfg.decoder_num_hidden_layers)


def load_procedural_into_reference(model, cfg, seed=0):
    P = O.procedural_params(cfg, seed)
    named = dict(model.named_parameters())
    assert list(named.keys()) == list(P.keys()), "oracle param inventory != reference named_parameters()"
    with torch.no_grad():
        for n, p in named.items():
            assert tuple(p.shape) == tuple(P[n].shape), (n, p.shape, P[n].shape)
            p.copy_(P[n])
    return P


def reference_forward(model, cfg, batch):
    kw = {}
    if cfg.stage_two and (cfg.do_pretrain or cfg.task_type == "caption"):
        kw = dict(input_caption_ids=batch["input_caption_ids"], decoder_mask=batch["decoder_mask"],
                  output_caption_ids=batch["output_caption_ids"])
    return model(batch["input_ids"], batch["token_type_ids"], batch["attention_mask"], batch["video"],
                 batch["video_mask"], pairs_masked_text=batch["pairs_masked_text"],
                 pairs_token_labels=batch["pairs_token_labels"], masked_video=batch["masked_video"],
                 video_labels_index=batch["video_labels_index"], **kw)


def sample(t, n=4096):
    """Deterministic strided subsample of a tensor (flattened)."""
    f = t.detach().reshape(-1)
    if f.numel() <= n:
        return f.to(torch.float32).numpy().copy()
    idx = torch.linspace(0, f.numel() - 1, n).long()
    return f[idx].to(torch.float32).numpy().copy()


def sample_exact(t, n):
    """Strided subsample with exact integer indices (float32 linspace of `sample` rounds beyond 2^24 elements)."""
    f = t.detach().reshape(-1)
    if f.numel() <= n:
        return f.to(torch.float32).numpy().copy()
    idx = (torch.arange(n, dtype=torch.int64) * (f.numel() - 1)) // (n - 1)
    return f[idx].to(torch.float32).numpy().copy()


def _pad(a, n):
    o = np.zeros(n, dtype=np.float32)
    o[:a.size] = a
    return o


def generate(name):
    cfg, rows, dseed = case_config(name)
    model = H.build_reference_model(_task_ns(cfg), vocab_size=cfg.vocab_size, zero_dropout=True)
    load_procedural_into_reference(model, cfg, seed=0)
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    out = {}
    # eval-mode public surface (modeling.py:299-313, 377-391)
    model.eval()
    with torch.no_grad():
        seq, vis = model.get_sequence_visual_output(batch["input_ids"], batch["token_type_ids"],
                                                    batch["attention_mask"], batch["video"], batch["video_mask"])
        sim = model.get_similarity_logits(seq, vis, batch["attention_mask"], batch["video_mask"])
        if cfg.has_decoder:
            logits = model.decoder_caption(seq, vis, batch["input_ids"], batch["attention_mask"],
                                           batch["video_mask"], batch["input_caption_ids"],
                                           batch["decoder_mask"], shaped=False, get_logits=True)
            out["decoder_logits_sample"] = sample(logits)
    out["sequence_output_sample"] = sample(seq)
    out["visual_output_sample"] = sample(vis)
    if seq.numel() <= 200000:
        out["sequence_output"] = seq.numpy().copy()
        out["visual_output"] = vis.numpy().copy()
    out["sim_matrix"] = sim.numpy().copy()
    # train-mode loss + grads (dropout p = 0)
    model.train()
    loss = reference_forward(model, cfg, batch)
    loss.backward()
    out["loss"] = np.array(float(loss), dtype=np.float64)
    names, norms, heads, sums, nograd = [], [], [], [], []
    for n, p in model.named_parameters():
        if p.grad is None:
            nograd.append(n)
            continue
        names.append(n)
        norms.append(float(p.grad.norm()))
        sums.append(float(p.grad.double().sum()))
        h = torch.zeros(8)
        k = min(8, p.grad.numel())
        h[:k] = p.grad.reshape(-1)[:k]
        heads.append(h.numpy())
    # strided samples of the gradients themselves: 256 elements of EVERY tensor, 4096 of the ten with the largest norm
    gdict = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    out["grad_samples"] = np.stack([_pad(sample_exact(gdict[n], 256), 256) for n in names]).astype(np.float32)
    top = sorted(range(len(names)), key=lambda i: -norms[i])[:10]
    out["grad_top_index"] = np.array(top, dtype=np.int64)
    out["grad_top_samples"] = np.stack([_pad(sample_exact(gdict[names[i]], 4096), 4096) for i in top]).astype(np.float32)
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms, dtype=np.float64)
    out["grad_sums"] = np.array(sums, dtype=np.float64)
    out["grad_heads"] = np.stack(heads).astype(np.float32)
    out["nograd_names"] = np.array(nograd)
    # one optimizer step exactly as main_task_retrieval.py:347-353 + prep_optimizer :168-195
    BertAdam = H.reference_bert_adam()
    groups = O.param_groups([n for n, _ in model.named_parameters()], lr=3e-5, coef_lr=0.1)
    pg = [{"params": [p], "weight_decay": groups[n]["weight_decay"], "lr": groups[n]["lr"]}
          for n, p in model.named_parameters()]
    opt = BertAdam(pg, lr=3e-5, warmup=0.1, schedule='warmup_linear', t_total=100, weight_decay=0.01,
                   max_grad_norm=1.0)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    for _ in range(2):   # two steps: step 0 has lr_scheduled = 0 under warmup_linear (optimization.py:156-159)
        total = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
    out["clip_total_norm"] = np.array(float(total), dtype=np.float64)
    dn, dh = [], []
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        d = (p.detach() - before[n])
        dn.append(float(d.double().norm()))
        h = torch.zeros(8)
        k = min(8, d.numel())
        h[:k] = d.reshape(-1)[:k]
        dh.append(h.numpy())
    out["adam_delta_norms"] = np.array(dn, dtype=np.float64)
    out["adam_delta_heads"] = np.stack(dh).astype(np.float32)
    out["config_json"] = np.array(json.dumps(cfg.to_dict()))
    out["rows"] = np.array(rows)
    out["data_seed"] = np.array(dseed)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"[golden] {name}: loss={float(loss):.6f} params_with_grad={len(names)} no_grad={nograd} -> {path} "
          f"({os.path.getsize(path) / 1024:.0f} KiB)")


# ------------------------------------------------------------------------------------------ cotangent goldens
# The FT-Align loss (hinge / CrossEn over B x B nearly equal cross-encoder scores) turns the backward into a DIFFERENCE of nearly
# equal per-pair gradients: under bf16 operand rounding the reference's own autocast run differs from its fp32 run by 40-80 % per
# tensor THROUGH THE LOSS, but only by ~1 % through (sim * W).sum() with a non-negative cotangent W.  These
# fixtures hold the reference's gradients for three seeded non-negative cotangents, so that the whole FT-Align backward (pooler,
# similarity_dense, cross encoder on B^2 pairs, both encoders, embeddings) has a WELL-POSED bf16 parity gate.  The reference's
# loss function for the cross-encoder similarity (modeling.py:209 / :265, `self.loss_fct`) is replaced by sim -> (sim * W).sum();
# everything else of UniVL.forward runs unchanged (on the pretrain path the other four losses stay in the gradient).
COT_CASES = ["align_small", "align_full", "pretrain_full"]
COT_KINDS = ["ones", "onehot", "absrandn"]


def cotangent(kind, n, seed=977):
    """Seeded non-negative [n, n] cotangent of the cross-encoder similarity matrix."""
    if kind == "ones":
        return torch.ones(n, n)
    if kind == "onehot":
        w = torch.zeros(n, n)
        w[1 % n, 2 % n] = 1.0
        return w
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, n, generator=g).abs()


def generate_cotangent(name):
    cfg, rows, dseed = case_config(name)
    model = H.build_reference_model(_task_ns(cfg), vocab_size=cfg.vocab_size, zero_dropout=True)
    load_procedural_into_reference(model, cfg, seed=0)
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    model.train()
    n = rows * cfg.n_pair
    out = {"kinds": np.array(COT_KINDS)}
    names = None
    for kind in COT_KINDS:
        Wc = cotangent(kind, n)
        seen = {}

        class Fct(torch.nn.Module):              # `loss_fct` is a registered child module of the reference model
            def forward(self, sim, Wc=Wc, seen=seen):
                assert tuple(sim.shape) == tuple(Wc.shape), (sim.shape, Wc.shape)
                seen["sim"] = sim.detach().clone()
                return (sim * Wc).sum()
        model.loss_fct = Fct()
        model.zero_grad()
        loss = reference_forward(model, cfg, batch)
        loss.backward()
        gd = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
        if names is None:
            names = list(gd)
            out["grad_names"] = np.array(names)
        assert names == list(gd)
        norms = [float(gd[k].norm()) for k in names]
        top = sorted(range(len(names)), key=lambda i: -norms[i])[:10]
        out["W_" + kind] = Wc.numpy().copy()
        out["sim_" + kind] = seen["sim"].numpy().copy()
        out["loss_" + kind] = np.array(float(loss), dtype=np.float64)
        out["grad_norms_" + kind] = np.array(norms, dtype=np.float64)
        out["grad_samples_" + kind] = np.stack([_pad(sample_exact(gd[k], 256), 256) for k in names]).astype(np.float32)
        out["grad_top_index_" + kind] = np.array(top, dtype=np.int64)
        out["grad_top_samples_" + kind] = np.stack([_pad(sample_exact(gd[names[i]], 4096), 4096) for i in top]).astype(np.float32)
        print(f"[golden] {name} cotangent {kind}: loss={float(loss):.6f} |g|={np.sqrt(np.sum(np.square(norms))):.4e}")
    path = os.path.join(GOLDEN_DIR, name + "_cot.npz")
    np.savez_compressed(path, **out)
    print(f"[golden] {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def dump_param_inventory():
    """Key lists of the reference's named_parameters()/state_dict per stage -> tests/golden/param_inventory.json."""
    inv = {}
    for name in ("joint_small", "align_small", "caption_small", "pretrain_small"):
        cfg, _, _ = case_config(name)
        model = H.build_reference_model(_task_ns(cfg), vocab_size=cfg.vocab_size)
        inv[name] = dict(named_parameters=[[n, list(p.shape)] for n, p in model.named_parameters()],
                         state_dict_keys=list(model.state_dict().keys()))
    with open(os.path.join(GOLDEN_DIR, "param_inventory.json"), "w") as f:
        json.dump(inv, f)
    print("[golden] param_inventory.json written")


# ------------------------------------------------------------------------------------------ beam search golden
def _reference_decode_functions():
    """The caption decoding helpers live in main_task_caption.py, which cannot be imported (it initialises NCCL at import
    time).  Their definitions are taken out of the script's syntax tree and executed here, in this container only: the
    reference's own code produces the golden hypotheses; nothing of it is stored in the repository."""
    import ast
    src = open(os.path.join(H.REFERENCE_ROOT, "main_task_caption.py")).read()
    want = {"get_inst_idx_to_tensor_position_map", "collect_active_part", "collate_active_info", "beam_decode_step",
            "collect_hypothesis_and_scores"}
    tree = ast.parse(src)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert {n.name for n in body} == want
    H._install_stubs()
    from modules.beam import Beam
    ns = {"torch": torch, "Beam": Beam}
    exec(compile(ast.Module(body=body, type_ignores=[]), "main_task_caption.py[decode helpers]", "exec"), ns)
    return ns, Beam


def generate_beam(name="caption_small", n_inst=3, n_bm=5, max_len=5, bos=101):
    """Runs the reference's caption decoding loop (main_task_caption.py:498-522) with the reference model on procedural
    weights and stores hypotheses / scores for two EOS settings: unreachable, and the token instance 0's top beam emits at
    its second step (so one instance stops early)."""
    cfg, rows, dseed = case_config(name)
    model = H.build_reference_model(_task_ns(cfg), vocab_size=cfg.vocab_size, zero_dropout=True)
    load_procedural_into_reference(model, cfg, seed=0)
    model.eval()
    ns, Beam = _reference_decode_functions()
    batch = O.synthetic_batch(cfg, n_inst, seed=dseed + 5)

    class _Tok:                       # Constants.from_tokenizer reads tokenizer.vocab[...]
        def __init__(self, eos):
            self.vocab = {"[PAD]": 0, "[UNK]": 100, "[CLS]": bos, "[SEP]": eos}

    def decode(eos, steps):
        dev = torch.device("cpu")
        with torch.no_grad():
            seq, vis = model.get_sequence_visual_output(batch["input_ids"], batch["token_type_ids"], batch["attention_mask"],
                                                        batch["video"], batch["video_mask"])
            n, len_s, d_h = seq.size()
            _, len_v, v_h = vis.size()
            input_ids = batch["input_ids"].view(-1, batch["input_ids"].shape[-1])
            input_mask = batch["attention_mask"].view(-1, batch["attention_mask"].shape[-1])
            video_mask = batch["video_mask"].view(-1, batch["video_mask"].shape[-1])
            tup = (seq.repeat(1, n_bm, 1).view(n * n_bm, len_s, d_h), vis.repeat(1, n_bm, 1).view(n * n_bm, len_v, v_h),
                   input_ids.repeat(1, n_bm).view(n * n_bm, len_s), input_mask.repeat(1, n_bm).view(n * n_bm, len_s),
                   video_mask.repeat(1, n_bm).view(n * n_bm, len_v))
            beams = [Beam(n_bm, device=dev, tokenizer=_Tok(eos)) for _ in range(n)]
            active = list(range(n))
            pos = ns["get_inst_idx_to_tensor_position_map"](active)
            for len_dec_seq in range(1, steps + 1):
                active = ns["beam_decode_step"](model.decoder_caption, beams, len_dec_seq, pos, n_bm, dev, tup)
                if not active:
                    break
                tup, pos = ns["collate_active_info"](tup, pos, active, n_bm, dev)
            hyp, scores = ns["collect_hypothesis_and_scores"](beams, 1)
        return [h[0] for h in hyp], [float(s[0]) for s in scores]

    out = {}
    hyp, sc = decode(-1, max_len)
    short, _ = decode(-1, 2)
    eos = short[0][1]
    hyp2, sc2 = decode(eos, max_len)
    pad = lambda hs: np.array([h + [-1] * (max_len - len(h)) for h in hs], dtype=np.int64)
    out.update(n_inst=n_inst, n_bm=n_bm, max_len=max_len, bos=bos, eos2=eos, hyp=pad(hyp), scores=np.array(sc),
               hyp2=pad(hyp2), scores2=np.array(sc2), data_seed=dseed + 5)
    np.savez(os.path.join(GOLDEN_DIR, "beam_" + name + ".npz"), **out)
    print("[golden] beam_%s.npz: hyp %s | eos=%d -> lengths %s" % (name, hyp, eos, [len(h) for h in hyp2]))


def generate_metrics():
    """metrics.compute_metrics of the reference (metrics.py:8-20) on seeded similarity matrices incl. ties with the diagonal."""
    H._install_stubs()
    import importlib
    ref_metrics = importlib.import_module("metrics")          # /root/reference/metrics.py
    out = {}
    for n in (7, 60, 333):
        g = torch.Generator().manual_seed(n)
        x = torch.randn(n, n, generator=g)
        x[3, 5] = x[3, 3]
        x[2, :] = 0.25
        x[1, 1] = x[1].max() + 1
        m = ref_metrics.compute_metrics(x.numpy())
        out["x%d" % n] = x.numpy()
        out["m%d" % n] = np.array([m["R1"], m["R5"], m["R10"], m["MR"]], dtype=np.float64)
    np.savez(os.path.join(GOLDEN_DIR, "metrics.npz"), **out)
    print("[golden] metrics.npz written")


if __name__ == "__main__":
    assert H.reference_available(), "reference not mounted; golden vectors can only be made in the build container"
    torch.set_num_threads(os.cpu_count())
    which = sys.argv[1:] or list(CASES) + ["beam", "metrics"] + [c + "_cot" for c in COT_CASES]
    for nm in which:
        if nm.endswith("_cot"):
            generate_cotangent(nm[:-4])
        elif nm == "beam":
            generate_beam()
        elif nm == "metrics":
            generate_metrics()
        else:
            generate(nm)
    if not sys.argv[1:]:
        dump_param_inventory()
