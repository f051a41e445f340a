"""Pins the oracle (oracle/univl_oracle.py) to golden vectors produced by the REAL reference classes
(oracle/make_golden.py).  CPU only.  Tolerance: fp32 re-association noise only (the oracle restates the same
fp32 arithmetic in a different op order) -> 2e-5 relative on loss / logits, 1e-4 relative on gradient norms."""
import json
import os

import numpy as np
import pytest
import torch

import univl_oracle as O
from make_golden import CASES, case_config, sample_exact

ALL = list(CASES)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def _sample(t, n=4096):
    f = t.detach().reshape(-1)
    if f.numel() <= n:
        return f.float().numpy()
    idx = torch.linspace(0, f.numel() - 1, n).long()
    return f[idx].float().numpy()


@pytest.mark.parametrize("name", ALL)
def test_oracle_matches_reference_golden(golden_dir, name):
    g = _load(golden_dir, name)
    cfg, rows, dseed = case_config(name)
    assert json.loads(str(g["config_json"])) == cfg.to_dict()
    P = {k: v.requires_grad_(True) for k, v in O.procedural_params(cfg, 0).items()}
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    # eval surface
    with torch.no_grad():
        seq, vis = O.get_sequence_visual_output(P, cfg, batch["input_ids"], batch["token_type_ids"],
                                                batch["attention_mask"], batch["video"], batch["video_mask"])
        v = lambda t: t.view(-1, t.shape[-1])
        sim = O.similarity_logits(seq, vis, v(batch["attention_mask"]), v(batch["video_mask"]), P, cfg, False)
    np.testing.assert_allclose(_sample(seq), g["sequence_output_sample"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(_sample(vis), g["visual_output_sample"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(sim.numpy(), g["sim_matrix"], rtol=2e-4, atol=2e-5)
    if cfg.has_decoder:
        with torch.no_grad():
            logits = O.decoder_caption(P, cfg, seq, vis, batch["attention_mask"], batch["video_mask"],
                                       batch["input_caption_ids"], batch["decoder_mask"])
        np.testing.assert_allclose(_sample(logits), g["decoder_logits_sample"], rtol=2e-4, atol=5e-5)
    # training loss + grads
    loss = O.univl_forward(P, cfg, batch, training=True)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= 2e-5 * max(1.0, abs(float(g["loss"])))
    names = [str(s) for s in g["grad_names"]]
    nograd = set(str(s) for s in g["nograd_names"])
    for n, p in P.items():
        if n in nograd:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
    for i, n in enumerate(names):
        gr = P[n].grad
        assert gr is not None, n
        ref = float(g["grad_norms"][i])
        assert abs(float(gr.norm()) - ref) <= 2e-4 * ref + 1e-7, (n, float(gr.norm()), ref)
        k = min(8, gr.numel())
        np.testing.assert_allclose(gr.reshape(-1)[:k].numpy(), g["grad_heads"][i][:k], rtol=2e-3, atol=2e-7,
                                   err_msg=n)
        # strided sample of the gradient itself (256 elements of every tensor): relative error of the sample
        rs = g["grad_samples"][i]
        got = sample_exact(gr, 256)
        d = float(np.linalg.norm(got - rs[:got.size]))
        assert d <= 5e-4 * float(np.linalg.norm(rs)) + 1e-6 * ref + 1e-9, (n, d, float(np.linalg.norm(rs)))
    for j, i in enumerate(g["grad_top_index"]):
        rs = g["grad_top_samples"][j]
        got = sample_exact(P[names[int(i)]].grad, 4096)
        assert float(np.linalg.norm(got - rs[:got.size])) <= 5e-4 * float(np.linalg.norm(rs)), names[int(i)]


@pytest.mark.parametrize("name", ["joint_small", "caption_small"])
def test_oracle_bert_adam_matches_reference(golden_dir, name):
    """clip_grad_norm_ + BertAdam (optimization.py:103-168), two steps, vs the reference's own optimizer."""
    g = _load(golden_dir, name)
    cfg, rows, dseed = case_config(name)
    P = {k: v.requires_grad_(True) for k, v in O.procedural_params(cfg, 0).items()}
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    O.univl_forward(P, cfg, batch, training=True).backward()
    names = [str(s) for s in g["grad_names"]]
    groups = O.param_groups(names, lr=3e-5, coef_lr=0.1)
    with torch.no_grad():
        before = {n: P[n].detach().clone() for n in names}
        state = {n: dict(m=torch.zeros_like(P[n]), v=torch.zeros_like(P[n]), step=0) for n in names}
        for _ in range(2):
            total = O.clip_grad_norm_([P[n].grad for n in names], 1.0)
            for n in names:
                st = state[n]
                st["step"] = O.bert_adam_step(P[n], P[n].grad, st["m"], st["v"], st["step"], groups[n]["lr"],
                                              0.1, 100, groups[n]["weight_decay"])
        assert abs(float(total) - float(g["clip_total_norm"])) <= 1e-4 * float(g["clip_total_norm"])
        for i, n in enumerate(names):
            d = (P[n].detach() - before[n])
            ref = float(g["adam_delta_norms"][i])
            assert abs(float(d.double().norm()) - ref) <= 2e-3 * ref + 1e-10, (n, float(d.norm()), ref)


def test_param_inventory_matches_reference(golden_dir):
    inv = json.load(open(os.path.join(golden_dir, "param_inventory.json")))
    for name, rec in inv.items():
        cfg, _, _ = case_config(name)
        shapes = O.param_shapes(cfg)
        assert [[n, list(s)] for n, s in shapes.items()] == rec["named_parameters"], name
        keys = set(shapes) | set(O.tied_aliases(cfg))
        assert keys == set(rec["state_dict_keys"]), name


def _golden_hyps(a):
    return [[int(t) for t in row if t >= 0] for row in a]


def test_oracle_beam_search_matches_reference_decode_loop(golden_dir):
    """oracle.beam_search_caption against hypotheses produced by the reference's OWN decode code (main_task_caption.py's
    beam_decode_step / collate_active_info / collect_hypothesis_and_scores and modules/beam.py, executed by
    oracle/make_golden.py::generate_beam on the reference model)."""
    g = _load(golden_dir, "beam_caption_small")
    cfg, rows, dseed = case_config("caption_small")
    n, nb, T, bos = int(g["n_inst"]), int(g["n_bm"]), int(g["max_len"]), int(g["bos"])
    P = O.procedural_params(cfg, 0)
    b = O.synthetic_batch(cfg, n, seed=int(g["data_seed"]))
    with torch.no_grad():
        so, vo = O.get_sequence_visual_output(P, cfg, b["input_ids"], b["token_type_ids"], b["attention_mask"], b["video"],
                                              b["video_mask"], training=False)
        am, vm = b["attention_mask"].view(n, -1), b["video_mask"].view(n, -1)
        hyp, sc = O.beam_search_caption(P, cfg, so, vo, am, vm, nb, T, bos, -1)
        hyp2, sc2 = O.beam_search_caption(P, cfg, so, vo, am, vm, nb, T, bos, int(g["eos2"]))
    assert hyp == _golden_hyps(g["hyp"]) and hyp2 == _golden_hyps(g["hyp2"])
    assert [len(h) for h in hyp2][0] == 2 and max(len(h) for h in hyp2) == T          # instance 0 stopped at its EOS
    np.testing.assert_allclose(sc, g["scores"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(sc2, g["scores2"], rtol=0, atol=2e-4)


def test_oracle_metrics_match_reference(golden_dir):
    """oracle.compute_metrics against values computed by the reference's metrics.compute_metrics (ties included)."""
    g = _load(golden_dir, "metrics")
    for n in (7, 60, 333):
        m = O.compute_metrics(g["x%d" % n])
        assert [m["R1"], m["R5"], m["R10"], float(m["MR"])] == list(g["m%d" % n])


@pytest.mark.parametrize("name", ["align_small"])
def test_oracle_matches_reference_cotangent_golden(golden_dir, name):
    """The cotangent fixtures (oracle/make_golden.py: the reference's gradients of (sim * W).sum() for seeded non-negative W, the
    well-posed form of the FT-Align backward) against the oracle's autograd."""
    from make_golden import COT_KINDS, cotangent
    g = _load(golden_dir, name + "_cot")
    cfg, rows, dseed = case_config(name)
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    names = [str(s) for s in g["grad_names"]]
    assert [str(k) for k in g["kinds"]] == COT_KINDS
    for kind in COT_KINDS:
        P = {k: v.requires_grad_(True) for k, v in O.procedural_params(cfg, 0).items()}
        Wc = cotangent(kind, rows * cfg.n_pair)
        np.testing.assert_array_equal(Wc.numpy(), g["W_" + kind])
        loss, parts = O.univl_forward(P, cfg, batch, training=True, return_parts=True, sim_loss_fct=lambda s: (s * Wc).sum())
        loss.backward()
        np.testing.assert_allclose(parts["sim_matrix"].detach().numpy(), g["sim_" + kind], rtol=2e-4, atol=2e-5)
        assert abs(float(loss) - float(g["loss_" + kind])) <= 2e-5 * max(1.0, abs(float(g["loss_" + kind])))
        gmax = float(np.max(g["grad_norms_" + kind]))
        for i, n in enumerate(names):
            ref = float(g["grad_norms_" + kind][i])
            got = float(P[n].grad.double().norm())
            assert abs(got - ref) <= 2e-4 * ref + 1e-6 * gmax, (kind, n, got, ref)
            rs = g["grad_samples_" + kind][i]
            gs = sample_exact(P[n].grad, 256)
            assert float(np.linalg.norm(gs - rs[:gs.size])) <= 5e-4 * float(np.linalg.norm(rs)) + 1e-6 * gmax, (kind, n)
