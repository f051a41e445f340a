"""Cached beam-search caption decoding (univl_amd.decode) against (a) the model's own full-recompute decoder_caption --
the call the reference's beam_decode_step makes every step (main_task_caption.py:450-452) -- and (b) the oracle's
restatement of the whole procedure (oracle.beam_search_caption)."""
import pytest
import torch

import univl_oracle as O
from make_golden import case_config

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from univl_amd.decode import CaptionBeamSearch
    from test_model_gpu import build

DEV = "cuda"


def _setup(dtype, n_inst=3):
    cfg, rows, dseed = case_config("caption_small")
    model, P = build(cfg, dtype)
    model.eval()
    b = O.synthetic_batch(cfg, n_inst, seed=dseed + 5)
    d = {k: v.to(DEV) for k, v in b.items()}
    with torch.no_grad():
        so, vo = model.get_sequence_visual_output(d["input_ids"], d["token_type_ids"], d["attention_mask"], d["video"], d["video_mask"])
    return cfg, model, P, b, d, so, vo


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cached_step_equals_full_recompute(dtype):
    """Feeding a fixed token sequence position by position through the cache gives the last-position log-probabilities
    of decoder_caption on the growing prefix, including after a beam permutation."""
    cfg, model, P, b, d, so, vo = _setup(dtype)
    n, nb, T = so.shape[0], 2, 6
    bs = CaptionBeamSearch(model, n, cfg.max_words, cfg.max_frames, n_bm=nb, max_len=T)
    am, vm = d["attention_mask"].view(n, -1), d["video_mask"].view(n, -1)
    bs.encode(so, vo, am, vm)
    g = torch.Generator().manual_seed(3)
    seqs = torch.randint(1000, 30000, (n * nb, T), generator=g).to(DEV)
    rep = lambda t: t.repeat_interleave(nb, dim=0)
    tol = 2e-3 if dtype == torch.float32 else 6e-2
    ident = torch.arange(n * nb, device=DEV)
    for t in range(T):
        parents = ident
        if t == 3:                                   # swap the two beams of every instance: caches must follow
            parents = ident.view(n, nb).flip(1).reshape(-1)
            seqs = seqs[parents]
        lp = bs.step_logprobs(t, seqs[:, t], parents if t > 0 else None)
        full = model.decoder_caption(rep(so), rep(vo), rep(d["input_ids"].view(n, -1)), rep(am), rep(vm), seqs[:, :t + 1],
                                     torch.ones_like(seqs[:, :t + 1]), shaped=True, get_logits=True)
        ref = torch.log_softmax(full[:, -1, :].float(), dim=-1)
        assert float((lp - ref).abs().max()) < tol, (t, float((lp - ref).abs().max()))


def test_beam_search_matches_reference_procedure():
    cfg, model, P, b, d, so, vo = _setup(torch.float32)
    n, nb, T = so.shape[0], 5, 5
    bs = CaptionBeamSearch(model, n, cfg.max_words, cfg.max_frames, n_bm=nb, max_len=T)
    am, vm = d["attention_mask"].view(n, -1), d["video_mask"].view(n, -1)
    so_c, vo_c = O.get_sequence_visual_output(P, cfg, b["input_ids"], b["token_type_ids"], b["attention_mask"], b["video"],
                                              b["video_mask"], training=False)
    am_c, vm_c = b["attention_mask"].view(n, -1), b["video_mask"].view(n, -1)
    bos = 101
    # 1) no EOS reachable: every instance runs the full length
    hyp, sc = bs(so, vo, am, vm, bos=bos, eos=-1)
    ref_hyp, ref_sc = O.beam_search_caption(P, cfg, so_c, vo_c, am_c, vm_c, nb, T, bos, -1)
    assert hyp == ref_hyp
    assert all(len(h) == T for h in hyp)
    assert max(abs(float(a) - b_) for a, b_ in zip(sc, ref_sc)) < 1e-3
    # 2) make the token instance 0's TOP beam emits at its second step the EOS: that instance must stop there
    short, _ = bs(so, vo, am, vm, bos=bos, eos=-1, max_len=2)
    eos = short[0][1]
    hyp2, sc2 = bs(so, vo, am, vm, bos=bos, eos=eos)
    ref2, ref_sc2 = O.beam_search_caption(P, cfg, so_c, vo_c, am_c, vm_c, nb, T, bos, eos)
    assert hyp2 == ref2
    assert len(hyp2[0]) == 2 and hyp2[0][-1] == eos and any(len(h) == T for h in hyp2)
    assert max(abs(float(a) - b_) for a, b_ in zip(sc2, ref_sc2)) < 1e-3


def test_beam_search_matches_reference_golden(golden_dir):
    """The cached GPU decoder reproduces the hypotheses the reference's own decode loop produced (tests/golden/
    beam_caption_small.npz, see oracle/make_golden.py::generate_beam)."""
    import os
    import numpy as np
    g = np.load(os.path.join(golden_dir, "beam_caption_small.npz"))
    cfg, rows, dseed = case_config("caption_small")
    n, nb, T, bos = int(g["n_inst"]), int(g["n_bm"]), int(g["max_len"]), int(g["bos"])
    model, P = build(cfg, torch.float32)
    model.eval()
    d = {k: v.to(DEV) for k, v in O.synthetic_batch(cfg, n, seed=int(g["data_seed"])).items()}
    with torch.no_grad():
        so, vo = model.get_sequence_visual_output(d["input_ids"], d["token_type_ids"], d["attention_mask"], d["video"], d["video_mask"])
    bs = CaptionBeamSearch(model, n, cfg.max_words, cfg.max_frames, n_bm=nb, max_len=T)
    am, vm = d["attention_mask"].view(n, -1), d["video_mask"].view(n, -1)
    unpad = lambda a: [[int(t) for t in row if t >= 0] for row in a]
    hyp, sc = bs(so, vo, am, vm, bos=bos, eos=-1)
    assert hyp == unpad(g["hyp"])
    assert float((sc.cpu() - torch.as_tensor(g["scores"], dtype=torch.float32)).abs().max()) < 1e-3
    hyp2, sc2 = bs(so, vo, am, vm, bos=bos, eos=int(g["eos2"]))
    assert hyp2 == unpad(g["hyp2"])
    assert float((sc2.cpu() - torch.as_tensor(g["scores2"], dtype=torch.float32)).abs().max()) < 1e-3
