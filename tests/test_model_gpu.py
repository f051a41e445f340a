"""End-to-end parity of the HIP path (through the Python surface that mirrors modules/modeling.py::UniVL and the
C ABI underneath) against (a) golden vectors produced by the REAL reference classes and (b) the oracle, on the
same procedural weights and synthetic inputs.  Dropout p = 0.  Needs an MI355X.

Tolerances (BASELINE.json north_star): fp32 mode 1e-3; bf16 mode 1e-2 on logits / loss / relative gradient error
(hidden states carry bf16 operand noise, gated at 5e-2 max-abs as BASELINE.md section 2 states)."""
import argparse
import os

import numpy as np
import pytest
import torch

import univl_oracle as O
from make_golden import case_config, sample_exact, FULL_CASES

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import univl_amd
    from univl_amd import UniVL, BertAdam, clip_grad_norm_

DEV = "cuda"
JOINT_CASES = ["joint_small", "joint_ones", "joint_full"]
ALL_CASES = JOINT_CASES + ["align_small", "caption_small", "pretrain_small"] + FULL_CASES   # FULL: BASELINE cfg3/FT-Align/cfg4/cfg5

# Every test of this module runs in the library's DETERMINISTIC mode unless it says otherwise (fixture below): fixed-order
# reductions instead of fp32 atomics, so that a measured error is a property of the code and not of one run.  The default
# (atomic) mode is held against the deterministic one in test_default_atomic_mode_matches_deterministic.
#
# bf16 gates.  north_star: 1e-2 on outputs (similarity logits, decoder logits, loss) and on the relative gradient error;
# hidden states carry bf16 operand noise that torch's own bf16 autocast of the REFERENCE shows too (measured the same way
# by oracle/bf16_noise.py -> tests/golden/bf16_autocast_noise.json, quoted in DESIGN.md section 2).  Every measured error
# is written to gpurun_out/parity_errors.json (copied to profiles/ per round).
#   gnorm    worst relative error of a per-tensor gradient norm          gsample  worst per-tensor relative error of the 256-sample
#   gmedian  median of those per-tensor sample errors                     gp95     their 95th percentile
#   gtop     worst error of the 4096-samples of the ten largest tensors   gglobal  ||all samples - ref|| / ||ref||
#   gcos     1 - cosine(all samples, ref)
GATES = {
    torch.float32: dict(hidden=1e-3, sim=1e-3, logits=1e-3, loss=1e-3, gnorm=1e-3, gsample=2e-3, gtop=2e-3, gmedian=1e-3,
                        gp95=1e-3, gglobal=1e-3, gcos=1e-5),
    # round 4: tightened to ~1.3 x the worst measured value of the cases that use them (cfg1-3, cfg4): gnorm 3.2e-3, gtop 1.3e-2,
    # gmedian 1.0e-2, gp95 1.5e-2, gglobal 1.2e-2, gcos 7.1e-5 -- and see check_against_reference_bf16 for the yardstick.
    # round 6: gnorm is back at 6e-3 for every case; the one case whose grouped weight gradients sum 6144 tokens on the
    # 256 x 256 body (32x32x16 MFMA chunks: another fp32 summation order) has its own entry in BF16_CASE_GATES below
    torch.bfloat16: dict(hidden=5e-2, sim=1e-2, logits=1e-2, loss=1e-2, gnorm=6e-3, gsample=4e-2, gtop=2e-2, gmedian=1.5e-2,
                         gp95=2e-2, gglobal=1.3e-2, gcos=1.5e-4),
}
# per-case bf16 gates: joint_b128 measures gnorm 6.6e-3 with the 256 body (4.6e-3 on the 128 tile), 0.47 x the reference's own bf16
# autocast run on that case (1.41e-2, tests/golden/bf16_autocast_noise.json) -- which stays the binding gate
BF16_CASE_GATES = {"joint_b128": dict(gnorm=8.5e-3)}
# Branch-specific bf16 GRADIENT gates.
#  * caption: the 30522-way softmax gradient through the tied table leaves a few small tensors at 3e-2 (measured 3.0-3.4e-2).
#  * align / pretrain THROUGH THE LOSS: the hinge / CrossEn over B x B nearly equal cross-encoder scores makes the gradient a
#    DIFFERENCE of nearly equal per-pair gradients -- the reference's own bf16 autocast run differs from its fp32 run by 38 %
#    (median tensor) / 80 % (worst) there (oracle/bf16_noise.py).  A worst-of-300-tensors statistic has no power in
#    that regime, so through the loss these branches are gated on the outputs (loss, sim at 1e-2), on the hinge-activity pattern, and
#    on aggregate gradient statistics (cosine, global error, median); the per-tensor gates at the 1e-2 level are applied where the
#    problem is well posed: test_backward_vs_reference_cotangent_golden (the reference's gradients of (sim * W).sum(), W >= 0).
BF16_GRAD_GATES = {
    "caption": dict(gsample=6e-2, gtop=5e-2, gp95=4e-2),
    "align": dict(gnorm=0.3, gsample=None, gtop=0.5, gmedian=0.35, gp95=None, gglobal=0.35, gcos=6e-2),
    "pretrain": dict(gnorm=3e-2, gsample=None, gtop=4e-2, gmedian=4e-2, gp95=0.15, gglobal=3e-2, gcos=1e-3),
}
# the cotangent form of the same branches (bf16): worst tensor / median / global
COT_GATES = {
    # align_small: a one-hot cotangent sends the whole gradient through ONE 32-token sequence of a 1+1+2-layer model; the
    # position-embedding gradient of the cross encoder (2 % of the largest norm) then carries 3.6e-2 of bf16 operand noise
    "align_small": dict(gnorm=1e-2, gsample=5e-2, gtop=1.5e-2, gmedian=1e-2, gp95=2e-2, gglobal=1e-2, gcos=5e-5, sim=1e-2),
    "align": dict(gnorm=1e-2, gsample=2.5e-2, gtop=1.5e-2, gmedian=1e-2, gp95=2e-2, gglobal=1e-2, gcos=5e-5, sim=1e-2),
    "pretrain": dict(gnorm=1.5e-2, gsample=8e-2, gtop=3e-2, gmedian=2e-2, gp95=4e-2, gglobal=1.5e-2, gcos=2e-4, sim=1e-2),
}


def gates_for(name, dtype):
    g = dict(GATES[dtype])
    branch = name.split("_")[0]
    if dtype == torch.bfloat16 and branch in BF16_GRAD_GATES:
        g.update(BF16_GRAD_GATES[branch])
    if dtype == torch.bfloat16 and name in BF16_CASE_GATES:
        g.update(BF16_CASE_GATES[name])
    return g


@pytest.fixture(autouse=True)
def _deterministic_mode():
    univl_amd.set_deterministic(True)
    yield
    univl_amd.set_deterministic(True)


class atomic_mode:
    """with atomic_mode(): the library's default mode (fp32 atomics) for the models BUILT inside the block."""

    def __enter__(self):
        univl_amd.set_deterministic(False)

    def __exit__(self, *a):
        univl_amd.set_deterministic(True)


_ERRORS = {}


def _record(name, dtype, **kw):
    import json
    _ERRORS.setdefault(name, {})[str(dtype).replace("torch.", "")] = {k: float(v) for k, v in kw.items()}
    path = os.environ.get("UNIVL_PARITY_OUT")          # bench.py's parity leg runs one case of this module and reads its numbers back
    if not path:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_errors.json")
    with open(path, "w") as f:
        json.dump(_ERRORS, f, indent=1, sort_keys=True)


def task_ns(cfg, dtype):
    return argparse.Namespace(
        max_words=cfg.max_words, max_frames=cfg.max_frames, video_dim=cfg.video_dim, batch_size=cfg.batch_size,
        n_gpu=cfg.n_gpu, n_pair=cfg.n_pair, margin=cfg.margin, negative_weighting=cfg.negative_weighting,
        hard_negative_rate=cfg.hard_negative_rate, use_mil=cfg.use_mil, do_pretrain=cfg.do_pretrain,
        task_type=cfg.task_type, stage_two=cfg.stage_two, train_sim_after_cross=cfg.train_sim_after_cross,
        text_num_hidden_layers=cfg.text_num_hidden_layers, visual_num_hidden_layers=cfg.visual_num_hidden_layers,
        cross_num_hidden_layers=cfg.cross_num_hidden_layers, decoder_num_hidden_layers=cfg.decoder_num_hidden_layers,
        local_rank=0, dropout_prob=0.0, compute_dtype="fp32" if dtype == torch.float32 else "bf16")


def build(cfg, dtype):
    model = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base",
                                  cache_dir=None, state_dict=None, task_config=task_ns(cfg, dtype))
    P = O.procedural_params(cfg, 0)
    sd = dict(P)
    for alias, owner in O.tied_aliases(cfg).items():      # the reference's state_dict lists tied tensors under both keys
        sd[alias] = P[owner]
    model.load_state_dict(sd, strict=True)
    model.to(DEV)
    return model, P


def call(model, batch):
    b = {k: v.to(DEV) for k, v in batch.items()}
    kw = {}
    if model.decoder is not None:
        kw = dict(input_caption_ids=b["input_caption_ids"], decoder_mask=b["decoder_mask"],
                  output_caption_ids=b["output_caption_ids"])
    return model(b["input_ids"], b["token_type_ids"], b["attention_mask"], b["video"], b["video_mask"],
                 pairs_masked_text=b["pairs_masked_text"], pairs_token_labels=b["pairs_token_labels"],
                 masked_video=b["masked_video"], video_labels_index=b["video_labels_index"], **kw)


def _sample(t, n=4096):
    f = t.detach().reshape(-1)
    if f.numel() <= n:
        return f.float().cpu().numpy()
    idx = torch.linspace(0, f.numel() - 1, n).long().to(f.device)
    return f[idx].float().cpu().numpy()


def max_abs(a, b):
    return float((torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max())


def compare_gradients(params, names, ref_norms, ref_samples, top_index, top_samples, G, f32):
    """Gradient statistics of a model against a golden fixture (per-tensor norms, 256-element strided samples of every tensor,
    4096-element samples of the ten largest).  Returns (err dict, list of per-tensor violations of the gates that are not None)."""
    # absolute floor: bf16 operand rounding leaves noise proportional to the LARGEST gradients flowing through the
    # same kernels (a bias whose true gradient is ~0, e.g. key.bias, only sees that noise); fp32 mode: rounding only
    gmax = float(np.max(ref_norms))
    floor = (1e-6 if f32 else 2e-3) * max(gmax, 1.0 if f32 else gmax)
    worst_norm = 0.0
    bad, rels, raw, allg, allr, relnames = [], [], [], [], [], []
    for i, n in enumerate(names):
        gr = params[n].grad
        assert gr is not None, n
        ref = float(ref_norms[i])
        got = float(gr.double().norm())
        # key.bias gradients are mathematically zero (softmax is invariant to a per-query shift): the reference
        # itself only holds rounding noise (~1e-10) there, hence the absolute floor.
        significant = ref > 1e-3 * gmax
        if significant:
            worst_norm = max(worst_norm, abs(got - ref) / ref)
        if not abs(got - ref) < G["gnorm"] * ref + floor:
            bad.append((n, "norm", got, ref))
        rs = ref_samples[i]
        gs = sample_exact(gr.float().cpu(), 256)
        rs = rs[:gs.size]
        allg.append(gs.astype(np.float64)); allr.append(rs.astype(np.float64))
        d = float(np.linalg.norm(gs - rs))
        rn = float(np.linalg.norm(rs))
        # A strided sample of a ROW-SPARSE gradient (the word table: a few hundred token rows carry the norm, the sample mostly
        # hits rows that only see the vocabulary heads' ~1e-5 terms) can be tiny against the tensor: its error is then measured
        # against the sample norm a uniformly spread gradient of the same tensor norm would have (dense tensors: the same number).
        rn_eff = max(rn, ref * (min(256, gr.numel()) / gr.numel()) ** 0.5)
        if significant and rn > 0:
            rels.append(d / rn_eff)
            raw.append(d / rn)                        # the statistic oracle/bf16_noise.py records for the reference's own bf16 run
            relnames.append((d / rn_eff, n, ref / gmax))
        if G["gsample"] is not None and not d < G["gsample"] * rn + floor * (min(256, gr.numel()) / gr.numel()) ** 0.5 + 1e-9:
            bad.append((n, "sample", d, rn))
    worst_top = 0.0
    for j, i in enumerate(top_index):                 # the ten largest gradients: 4096-element samples
        rs = top_samples[j]
        gs = sample_exact(params[names[int(i)]].grad.float().cpu(), 4096)
        if float(np.linalg.norm(rs)) > 0:              # (a one-hot cotangent leaves whole strided samples of a table at zero)
            worst_top = max(worst_top, float(np.linalg.norm(gs - rs[:gs.size])) / float(np.linalg.norm(rs)))
    ag, ar = np.concatenate(allg), np.concatenate(allr)
    print("[worst tensors: sample error, name, norm / largest norm] " + "; ".join("%.2e %s %.1e" % t for t in sorted(relnames, reverse=True)[:4]))
    if relnames:                                      # the worst tensor's largest element differences: (sample position, ours, reference)
        wn = max(relnames)[1]
        wi = names.index(wn)
        gs = sample_exact(params[wn].grad.float().cpu(), 256)
        rs = ref_samples[wi][:gs.size]
        top = np.argsort(-np.abs(gs - rs))[:6]
        print("[worst tensor %s: |ref sample| %.3e, |diff| %.3e] " % (wn, np.linalg.norm(rs), np.linalg.norm(gs - rs)) +
              "; ".join("#%d %.4e vs %.4e" % (int(k), gs[k], rs[k]) for k in top))
    err = dict(gnorm=worst_norm, gsample=float(np.max(rels)), gtop=worst_top, gmedian=float(np.median(rels)),
               gp95=float(np.percentile(rels, 95)), gglobal=float(np.linalg.norm(ag - ar) / np.linalg.norm(ar)),
               gcos=float(1.0 - np.dot(ag, ar) / (np.linalg.norm(ag) * np.linalg.norm(ar))), gmedian_raw=float(np.median(raw)))
    return err, bad


def check_gates(tag, err, G):
    for k, v in err.items():
        if G.get(k) is not None:
            assert v < G[k], (tag, k, v, G[k])


_NOISE = None


def check_against_reference_bf16(golden_dir, name, err):
    """bf16 only.  The yardstick for "how close can a bf16 run be": the REFERENCE model itself under torch.autocast(bfloat16) against
    its own fp32 run on the same inputs (oracle/bf16_noise.py -> tests/golden/bf16_autocast_noise.json: median per-tensor sample error and
    worst per-tensor norm error, the statistics `gmedian_raw` / `gnorm` are computed the same way).  This library's bf16 step must be
    AT LEAST as close to the reference's fp32 gradients as the reference's own bf16 run is: median strictly, the worst norm within 1.25 x
    (a worst-of-300 statistic of two different rounding patterns).  Measured: 0.55 - 0.65 x the reference's median at cfg1-3, 0.9 x at
    cfg4, 0.45 x at cfg5, 0.2 - 0.55 x through the FT-Align hinge."""
    global _NOISE
    if _NOISE is None:
        import json
        with open(os.path.join(golden_dir, "bf16_autocast_noise.json")) as f:
            _NOISE = json.load(f)
    ref = _NOISE.get(name)
    if ref is None:
        return {}
    out = dict(vs_autocast_median=err["gmedian_raw"] / ref["grad_sample_rel_median"], vs_autocast_norm=err["gnorm"] / ref["grad_norm_rel_max"])
    assert err["gmedian_raw"] <= ref["grad_sample_rel_median"], (name, "median", err["gmedian_raw"], ref["grad_sample_rel_median"])
    assert err["gnorm"] <= 1.25 * ref["grad_norm_rel_max"], (name, "norm", err["gnorm"], ref["grad_norm_rel_max"])
    return out


def hinge_pattern(sim, margin):
    """Which terms of MaxMarginRankingLoss (until_module.py:245-251) are active: relu(margin + s_ij - s_ii), relu(margin + s_ij - s_jj)."""
    s = np.asarray(sim, dtype=np.float64)
    d = np.diag(s)
    return np.stack([(margin + s - d[:, None]) > 0, (margin + s - d[None, :]) > 0])


def _golden_params():
    """(case, dtype) grid; the through-the-loss bf16 runs of the ill-conditioned branches are marked `statistical` (conftest.py runs
    those after every deterministic-gate test of the session)."""
    out = []
    for name in ALL_CASES:
        for dtype in (torch.float32, torch.bfloat16):
            ill = dtype == torch.bfloat16 and name.split("_")[0] in ("align", "pretrain")
            out.append(pytest.param(name, dtype, marks=[pytest.mark.statistical] if ill else [],
                                    id="%s-%s" % (name, "bf16" if dtype == torch.bfloat16 else "fp32")))
    return out


def plan_ops(model):
    """names of every enqueue of the model's built step plans (forward + both backward forms)"""
    out = []
    for st in model._steps.values():
        if not hasattr(st, "fwd"):
            continue                   # the evaluation entry points keep other objects in the same cache
        out += [op[3] for op in st.fwd.ops]
        for fresh in (True, False):
            try:
                out += [op[3] for op in st.backward_plan(fresh).ops]
            except Exception:          # eval steps have no backward
                pass
    return out


@pytest.mark.parametrize("name,dtype", _golden_params())
def test_forward_backward_vs_reference_golden(golden_dir, name, dtype):
    _golden_check(golden_dir, name, dtype, default_mode=False)


# The configuration bench.py TIMES is the library's default mode: fp32 atomics, split-K, the LayerNorm folds (univl_gemm_ln /
# univl_gemm_pair_ln), riders.  Round 4 held it against the reference only through the deterministic twin at a 4e-2 aggregate
# (VERDICT r4 weak 1); here the default mode itself goes through the SAME golden gates as the deterministic one -- x 1.1, because its
# sums meet in hardware order: two runs differ by a few 1e-4 in these statistics and the deterministic gates sit at ~1.3 x measured --
# taken as the worst of two independent runs, and the test asserts that the plans it ran really contain the folds / the 256 body.
DEFAULT_MODE_CASES = [("joint_full", ("univl_gemm_ln", "univl_gemm_pair_ln")), ("joint_b16", ("univl_gemm_pair",)),
                      ("joint_b32", ("univl_gemm_group",)), ("joint_b64", ("univl_gemm_group",)), ("joint_b128", ("univl_gemm_group",)),
                      ("align_full", ("univl_gemm_ln", "univl_gemm_pair_ln")), ("caption_small", ("univl_gemm_ln",)),
                      ("caption_full", ()), ("pretrain_small", ("univl_gemm_ln",)), ("pretrain_full", ())]


@pytest.mark.parametrize("name,must_run", [pytest.param(n, ops_, marks=[pytest.mark.statistical] if n.split("_")[0] in ("align", "pretrain") else [],
                                                        id=n) for n, ops_ in DEFAULT_MODE_CASES])
def test_forward_backward_vs_reference_golden_default_mode(golden_dir, name, must_run):
    with atomic_mode():
        for _ in range(2):
            model = _golden_check(golden_dir, name, torch.bfloat16, default_mode=True)
        ran = set(plan_ops(model))
        for op in must_run:
            assert op in ran, (name, op, sorted(ran))


def _golden_check(golden_dir, name, dtype, default_mode):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg, rows, dseed = case_config(name)
    model, P = build(cfg, dtype)
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    f32 = dtype == torch.float32
    G = gates_for(name, dtype)
    if default_mode:
        G = {k: (None if v is None else 1.1 * v) for k, v in G.items()}
    err = {}
    # ---- eval surface: get_sequence_visual_output + get_similarity_logits (main_task_retrieval.py:398, 376)
    model.eval()
    with torch.no_grad():
        b = {k: v.to(DEV) for k, v in batch.items()}
        seq, vis = model.get_sequence_visual_output(b["input_ids"], b["token_type_ids"], b["attention_mask"],
                                                    b["video"], b["video_mask"])
        sim = model.get_similarity_logits(seq, vis, b["attention_mask"], b["video_mask"])
        assert model(b["input_ids"], b["token_type_ids"], b["attention_mask"], b["video"], b["video_mask"]) is None
        if model.decoder is not None:
            logits = model.decoder_caption(seq, vis, b["input_ids"], b["attention_mask"], b["video_mask"],
                                           b["input_caption_ids"], b["decoder_mask"], shaped=False, get_logits=True)
            assert tuple(logits.shape[-1:]) == (cfg.vocab_size,)
            ref_l = g["decoder_logits_sample"]
            err["logits"] = float(np.abs(_sample(logits) - ref_l).max()) / max(1.0, float(np.abs(ref_l).max()))
    if "sequence_output" in g.files:
        err["hidden"] = max(max_abs(seq, g["sequence_output"]), max_abs(vis, g["visual_output"]))
    else:                                     # large cases store the strided 4096-sample only
        err["hidden"] = max(float(np.abs(_sample(seq) - g["sequence_output_sample"]).max()),
                            float(np.abs(_sample(vis) - g["visual_output_sample"]).max()))
    err["sim"] = max_abs(sim, g["sim_matrix"]) / max(1.0, float(np.abs(g["sim_matrix"]).max()))
    if cfg.train_sim_after_cross and not cfg.use_mil:
        # FT-Align: the same hinges are active as in the reference (the loss is not differentiable where one flips)
        assert np.array_equal(hinge_pattern(sim.float().cpu().numpy(), cfg.margin), hinge_pattern(g["sim_matrix"], cfg.margin))
    # ---- training step: loss + every parameter gradient (main_task_retrieval.py:333-342)
    model.train()
    loss = call(model, batch)
    loss.backward()
    err["loss"] = abs(float(loss) - float(g["loss"])) / max(1.0, abs(float(g["loss"])))
    names = [str(s) for s in g["grad_names"]]
    nograd = [str(s) for s in g["nograd_names"]]
    params = dict(model.named_parameters())
    for n in nograd:
        assert params[n].grad is None, n                      # dead poolers stay grad-less, as in the reference
    gerr, bad = compare_gradients(params, names, g["grad_norms"], g["grad_samples"], g["grad_top_index"], g["grad_top_samples"], G, f32)
    err.update(gerr)
    tag = name + ("@default" if default_mode else "")
    _record(tag, dtype, **err)
    print(f"[parity {tag} {dtype}] " + " ".join(f"{k}={v:.2e}" for k, v in sorted(err.items())))
    assert not bad, bad[:5]
    check_gates(tag, err, G)
    if not f32 and not default_mode:
        err.update(check_against_reference_bf16(golden_dir, name, err))
        _record(tag, dtype, **err)
    return model


def _pooler_step(model):
    sts = [st for st in model._steps.values() if getattr(st, "pooler", None) is not None and st.cx.training]
    assert len(sts) == 1
    return sts[0]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["align_small", "align_full", "pretrain_full"])
def test_backward_vs_reference_cotangent_golden(golden_dir, name, dtype):
    """The WELL-POSED form of the FT-Align backward (modeling.py:341-375 on all B^2 pairs; until_module.py:223-251 is what makes the
    through-the-loss form ill-conditioned): gradients of (sim * W).sum() for three seeded non-negative cotangents W against the
    reference's (tests/golden/*_cot.npz, oracle/make_golden.py: `loss_fct` replaced, everything else of UniVL.forward unchanged --
    on the pretrain path the other four losses stay in the gradient).  The cotangent is fed where the loss kernel leaves d loss /
    d sim (steps.PoolerSim.dsim, test-only access); everything downstream is the production backward plan."""
    from make_golden import COT_KINDS
    g = np.load(os.path.join(golden_dir, name + "_cot.npz"))
    cfg, rows, dseed = case_config(name)
    model, P = build(cfg, dtype)
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    f32 = dtype == torch.float32
    G = dict(GATES[dtype])
    if not f32:
        G.update(COT_GATES.get(name, COT_GATES[name.split("_")[0]]))
    names = [str(s) for s in g["grad_names"]]
    model.train()
    for kind in COT_KINDS:
        model.zero_grad(set_to_none=True)
        loss = call(model, batch)
        st = _pooler_step(model)
        Wc = torch.from_numpy(g["W_" + kind]).to(DEV)
        err = dict(sim=max_abs(st.pooler.sim.view(Wc.shape), g["sim_" + kind]) / max(1.0, float(np.abs(g["sim_" + kind]).max())))
        st.pooler.dsim.copy_(Wc.reshape(-1))            # d loss / d sim := W   (the loss kernel had written the hinge's)
        loss.backward()
        params = dict(model.named_parameters())
        gerr, bad = compare_gradients(params, names, g["grad_norms_" + kind], g["grad_samples_" + kind],
                                      g["grad_top_index_" + kind], g["grad_top_samples_" + kind], G, f32)
        err.update(gerr)
        _record(name + "_cot_" + kind, dtype, **err)
        print(f"[cotangent {name} {kind} {dtype}] " + " ".join(f"{k}={v:.2e}" for k, v in sorted(err.items())))
        assert not bad, (kind, bad[:5])
        check_gates((name, kind), err, G)


def _grads_and_loss(name, dtype, steps=0):
    cfg, rows, dseed = case_config(name)
    model, _ = build(cfg, dtype)
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    model.train()
    loss = call(model, batch)
    loss.backward()
    out = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    out["__loss__"] = loss.detach().clone()
    if steps:
        opt = BertAdam(model.parameters(), lr=1e-4, warmup=-1, t_total=-1, weight_decay=0.01, max_grad_norm=1.0)
        for _ in range(steps):
            clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            opt.zero_grad()
            call(model, batch).backward()
        out.update({"param:" + n: p.detach().clone() for n, p in model.named_parameters()})
    return out


@pytest.mark.parametrize("name,dtype,steps", [("align_full", torch.bfloat16, 0), ("pretrain_small", torch.bfloat16, 2),
                                              ("caption_small", torch.bfloat16, 2), ("joint_small", torch.float32, 2)])
def test_deterministic_mode_is_bit_reproducible(name, dtype, steps):
    """univl_set_deterministic(1): two independent runs (fresh model, fresh workspaces) of the forward + backward (+ clip + BertAdam
    steps) give the SAME BITS -- loss, every gradient tensor, every parameter.  This is what makes the measured parity errors of
    this module properties of the code rather than of one run (the default mode orders its fp32 atomics differently every time)."""
    a = _grads_and_loss(name, dtype, steps)
    b = _grads_and_loss(name, dtype, steps)
    assert a.keys() == b.keys()
    diff = [k for k in a if not torch.equal(a[k], b[k])]
    assert not diff, diff[:8]


@pytest.mark.parametrize("name,dtype", [("joint_full", torch.float32), ("align_small", torch.float32), ("pretrain_small", torch.float32),
                                        ("caption_small", torch.float32), ("joint_full", torch.bfloat16), ("caption_small", torch.bfloat16)])
def test_default_atomic_mode_matches_deterministic(name, dtype):
    """The production default (fp32 atomics: split-K, column sums, scatter-adds) computes the same sums as the deterministic mode in
    another order: fp32 compute -> per-tensor differences at rounding level; bf16 compute -> a different summation order flips bf16
    roundings downstream, so the comparison is the aggregate one."""
    det = _grads_and_loss(name, dtype)
    with atomic_mode():
        atom = _grads_and_loss(name, dtype)
    assert det.keys() == atom.keys()
    f32 = dtype == torch.float32
    gmax = max(float(v.double().norm()) for k, v in det.items() if k != "__loss__")
    assert abs(float(det["__loss__"]) - float(atom["__loss__"])) < (1e-5 if f32 else 2e-3) * max(1.0, abs(float(det["__loss__"])))
    num = den = 0.0
    for k, v in det.items():
        if k == "__loss__":
            continue
        d = float((v.double() - atom[k].double()).norm())
        n = float(v.double().norm())
        num += d * d; den += n * n
        if f32:
            assert d < 2e-4 * n + 1e-6 * gmax, (k, d, n)
    rel = (num / den) ** 0.5
    # PER-ROW check of the tensors that keep one row per token position (position-embedding gradients: row s = the sum over the batch
    # of the stack-input gradient of the tokens at position s): one token row mishandled by a fold's last arrivers -- a LayerNorm row not
    # normalised, a LayerNorm-backward row skipped -- changes every gradient flowing through that token, i.e. 1 / batch of one row of
    # these tensors at O(1), while the aggregate above moves by ~1 / tokens.  Rows of the two modes differ by rounding order only.
    rows_checked, row_worst = 0, 0.0
    for k in ("bert.embeddings.position_embeddings.weight", "visual.embeddings.position_embeddings.weight"):
        if k in det:
            a, b = det[k].double(), atom[k].double()
            rn = a.norm(dim=1)
            live = rn > 1e-3 * float(rn.max())
            rd = ((a - b).norm(dim=1) / rn.clamp_min(1e-30))[live]
            rows_checked += int(live.sum())
            row_worst = max(row_worst, float(rd.max()))
    print("[atomic vs deterministic %s %s] global relative difference %.3e; %d position rows, worst row difference %.3e"
          % (name, dtype, rel, rows_checked, row_worst))
    assert rows_checked > 0 and row_worst < (1e-4 if f32 else 6e-2), (name, row_worst)
    # bf16: both modes sit ~1e-2 (global) from the fp32 reference with independent rounding patterns; a missing or doubled
    # contribution in either would show up at O(1)
    assert rel < (1e-4 if f32 else 4e-2), (name, rel)


def test_layernorm_folds_at_576_tokens_match_the_plans_without_them(ab):
    """Round 6 moved the upper end of the LayerNorm folds from 512 to 640 tokens (12 pairs x 48: -2.6 % per step).  No BASELINE golden
    sits in that range, so: 12 rows of the joint configuration in the default (atomic) mode with the folds against the same step with
    ln_fold = 0 -- same sums in another order (aggregate difference at the bf16 reordering level, every position row within it) -- and
    the folded plan really holds the fold launches."""
    import dataclasses
    cfg, _, dseed = case_config("joint_b16")
    cfg = dataclasses.replace(cfg, batch_size=12)
    batch = O.synthetic_batch(cfg, 12, seed=dseed)

    def run():
        model, _ = build(cfg, torch.bfloat16)
        model.train()
        loss = call(model, batch)
        loss.backward()
        g = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}
        return float(loss), g, set(plan_ops(model))

    with atomic_mode():
        l1, g1, ops1 = run()
        ab(ln_fold=0)
        l0, g0, ops0 = run()
        ab(ln_fold=None)
    assert "univl_gemm_ln" in ops1 and "univl_gemm_pair_ln" in ops1 and "univl_gemm_ln" not in ops0 and "univl_gemm_pair_ln" not in ops0
    assert abs(l1 - l0) < 2e-3 * max(1.0, abs(l0))
    num = sum(float((g1[k] - g0[k]).norm()) ** 2 for k in g0)
    den = sum(float(g0[k].norm()) ** 2 for k in g0)
    assert (num / den) ** 0.5 < 4e-2, (num / den) ** 0.5
    for k in ("bert.embeddings.position_embeddings.weight", "visual.embeddings.position_embeddings.weight"):
        rn = g0[k].norm(dim=1)
        live = rn > 1e-3 * float(rn.max())
        rd = ((g1[k] - g0[k]).norm(dim=1) / rn.clamp_min(1e-30))[live]
        assert int(live.sum()) > 0 and float(rd.max()) < 6e-2, (k, float(rd.max()))


def test_large_batch_backward_paths_match_the_default_ones(ab):
    """Two backward forms that only switch on at large per-GPU batch -- position-table gradients by a gather over per-token rows
    (dpos_gather_min, default 32 rows per position) and the grouped weight gradients on the 128 tile with their bias gradients
    taken by column-sum workgroups of the same launch (wgrad_big_min, default 5462 tokens) -- forced on at the size of the
    golden case: every gradient tensor equals the default path's up to fp32 summation order (deterministic mode: the products
    themselves are bit-identical), and the golden gates hold."""
    ref = _grads_and_loss("joint_full", torch.bfloat16)
    ab(dpos_gather_min=1, wgrad_big_min=1)
    new = _grads_and_loss("joint_full", torch.bfloat16)
    cfg, rows, dseed = case_config("joint_full")
    model, _ = build(cfg, torch.bfloat16)
    model.train()
    call(model, O.synthetic_batch(cfg, rows, seed=dseed)).backward()
    st = next(iter(model._steps.values()))
    kinds = [op[3] for op in st.backward_plan(True).ops]
    assert kinds.count("univl_gemm_pair") == 0 and kinds.count("univl_gemm_group") >= cfg.text_num_hidden_layers + cfg.visual_num_hidden_layers
    assert ref.keys() == new.keys()
    assert float(ref["__loss__"]) == float(new["__loss__"])
    for k, v in ref.items():
        d, n = float((v.double() - new[k].double()).norm()), float(v.double().norm())
        assert d <= 2e-5 * n + 1e-9, (k, d, n)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_joint_full_gradients_vs_oracle_elementwise(dtype):
    """Full-tensor comparison of every gradient with the oracle (autograd on CPU) for a 2+1 layer model."""
    cfg, rows, dseed = case_config("joint_small")
    model, P = build(cfg, dtype)
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    model.train()
    loss = call(model, batch)
    loss.backward()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = O.univl_forward(Pr, cfg, batch, training=True)
    ref.backward()
    f32 = dtype == torch.float32
    assert abs(float(loss) - float(ref)) < (1e-3 if f32 else 1e-2)
    for n, p in model.named_parameters():
        if Pr[n].grad is None:
            assert p.grad is None
            continue
        d = (p.grad.double().cpu() - Pr[n].grad.double())
        refn = float(Pr[n].grad.double().norm())
        assert float(d.norm()) < (1e-3 if f32 else 4e-2) * refn + (1e-6 if f32 else 5e-5), (n, float(d.norm()), refn)


def test_gradient_accumulation_and_loss_scaling():
    """Two backward passes accumulate (as p.grad += in the reference under gradient_accumulation_steps > 1,
    main_task_retrieval.py:339-345) and an upstream factor (loss / k) scales the gradients."""
    cfg, rows, dseed = case_config("joint_small")
    model, _ = build(cfg, torch.float32)
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    model.train()
    call(model, batch).backward()
    g1 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    (call(model, batch) / 2).backward()
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        d = float((p.grad - 1.5 * g1[n]).abs().max())
        assert d < 1e-5 + 1e-4 * float(g1[n].abs().max()), n
    model.zero_grad(set_to_none=True)
    call(model, batch).backward()
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert float((p.grad - g1[n]).abs().max()) < 1e-5 + 1e-4 * float(g1[n].abs().max()), n


@pytest.mark.parametrize("deferred", [True, False])
def test_clip_and_bert_adam_vs_reference_golden(golden_dir, deferred):
    """clip_grad_norm_ + BertAdam, two steps, against the reference's own optimizer (optimization.py:103-168)."""
    name = "joint_small"
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg, rows, dseed = case_config(name)
    model, P = build(cfg, torch.float32)
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    model.train()
    call(model, batch).backward()
    names = [n for n, _ in model.named_parameters()]
    groups = O.param_groups(names, lr=3e-5, coef_lr=0.1)
    pg = [{"params": [p], "weight_decay": groups[n]["weight_decay"], "lr": groups[n]["lr"]} for n, p in model.named_parameters()]
    opt = BertAdam(pg, lr=3e-5, warmup=0.1, schedule='warmup_linear', t_total=100, weight_decay=0.01, max_grad_norm=1.0)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    for _ in range(2):
        total = clip_grad_norm_(model.parameters(), 1.0, deferred=deferred)
        opt.step()
    assert abs(float(total) - float(g["clip_total_norm"])) < 1e-3 * float(g["clip_total_norm"])
    gnames = [str(s) for s in g["grad_names"]]
    params = dict(model.named_parameters())
    for i, n in enumerate(gnames):
        d = (params[n].detach() - before[n]).double()
        ref = float(g["adam_delta_norms"][i])
        assert abs(float(d.norm()) - ref) < 5e-3 * ref + 1e-9, (n, float(d.norm()), ref)
        k = min(8, d.numel())
        scale = max(float(np.abs(g["adam_delta_heads"][i][:k]).max()), ref / (d.numel() ** 0.5))
        assert max_abs(d.reshape(-1)[:k], g["adam_delta_heads"][i][:k]) < 2e-2 * scale + 1e-9, n
    for n in [str(s) for s in g["nograd_names"]]:
        assert torch.equal(params[n].detach(), before[n]), n          # grad-less parameters are not touched
    st = opt.state[params[gnames[0]]]
    assert st["step"] == 2 and st["next_m"].shape == params[gnames[0]].shape


def test_training_reduces_loss_fp32():
    cfg, rows, dseed = case_config("joint_small")
    model, P = build(cfg, torch.float32)
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    model.train()
    opt = BertAdam(model.parameters(), lr=1e-6, warmup=-1, t_total=-1, weight_decay=0.01, max_grad_norm=1.0)
    l0 = float(call(model, batch))
    for _ in range(3):
        opt.zero_grad()
        loss = call(model, batch)
        loss.backward()
        clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
    assert float(call(model, batch)) < l0                      # the optimizer reduces the loss


def test_bf16_shadow_follows_optimizer_and_state_dict_roundtrip(tmp_path):
    cfg, rows, dseed = case_config("joint_small")
    model, P = build(cfg, torch.bfloat16)
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    model.train()
    opt = BertAdam(model.parameters(), lr=1e-4, warmup=-1, t_total=-1, weight_decay=0.01, max_grad_norm=1.0)
    for _ in range(3):
        opt.zero_grad()
        loss = call(model, batch)
        loss.backward()
        clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
    fl = model.flat
    used = model.used_parameter_names()
    for n in (used[0], used[5], used[-1]):
        assert torch.equal(fl.wop(n), fl.w32(n).to(torch.bfloat16)), n     # the step rewrote the bf16 shadow
        assert not torch.equal(fl.w32(n).cpu(), P[n]), n
    # checkpoint interchange: same keys as the reference's state_dict, reload gives the same outputs
    sd = model.state_dict()
    assert set(sd.keys()) == set(O.param_shapes(cfg).keys()) | set(O.tied_aliases(cfg))
    path = os.path.join(tmp_path, "pytorch_model.bin.0")
    torch.save(sd, path)
    model2, _ = build(cfg, torch.bfloat16)
    model2.load_state_dict(torch.load(path, map_location="cpu"))
    model.eval(); model2.eval()
    b = {k: v.to(DEV) for k, v in batch.items()}
    a1 = model.get_sequence_visual_output(b["input_ids"], b["token_type_ids"], b["attention_mask"], b["video"], b["video_mask"])
    a2 = model2.get_sequence_visual_output(b["input_ids"], b["token_type_ids"], b["attention_mask"], b["video"], b["video_mask"])
    # not bit-identical run to run: split-K and bias/LayerNorm gradients accumulate with fp32 atomics
    assert max_abs(a1[0], a2[0]) < 2e-2 and max_abs(a1[1], a2[1]) < 2e-2


@pytest.mark.statistical
def test_dropout_training_runs_and_is_seeded():
    cfg, rows, dseed = case_config("joint_small")
    ns = task_ns(cfg, torch.bfloat16)
    ns.dropout_prob = 0.1
    model = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base", task_config=ns)
    model.load_state_dict(O.procedural_params(cfg, 0))
    model.to(DEV).train()
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    l1 = call(model, batch); l1.backward()
    l2 = call(model, batch)
    assert torch.isfinite(l1) and torch.isfinite(l2) and float(l1) != float(l2)     # a fresh mask per step
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n


def _train(kind, case, steps=5, lr=1e-5, dtype=torch.float32, spy=None):
    """`steps` optimizer steps on one fixed batch; returns the per-step losses, sampled final parameters and
    bookkeeping of the gradient exchange.  kind: eager | graph | graph_nopipe | loopback_eager | loopback_graph
    (graph / loopback_graph: BertAdam pipelined with the next forward, univl_amd.graphed)."""
    from univl_amd.graphed import GraphedTrainStep
    cfg, rows, dseed = case_config(case)
    model, P = build(cfg, dtype)
    model.train()
    if kind.startswith("loopback"):
        model.enable_data_parallel(loopback=True)
    opt = BertAdam(model.parameters(), lr=lr, warmup=-1, t_total=-1, weight_decay=0.01, max_grad_norm=1.0)
    b = {k: v.to(DEV) for k, v in O.synthetic_batch(cfg, rows, seed=dseed).items()}
    args = (b["input_ids"], b["token_type_ids"], b["attention_mask"], b["video"], b["video_mask"])
    kw = dict(pairs_masked_text=b["pairs_masked_text"], pairs_token_labels=b["pairs_token_labels"],
              masked_video=b["masked_video"], video_labels_index=b["video_labels_index"])
    if model.decoder is not None:
        kw.update(input_caption_ids=b["input_caption_ids"], decoder_mask=b["decoder_mask"],
                  output_caption_ids=b["output_caption_ids"])
    losses = []
    if "graph" in kind:
        gs = GraphedTrainStep(model, opt, max_grad_norm=1.0, warmup=1, pipeline_optimizer=not kind.endswith("nopipe"))
        for _ in range(steps):
            losses.append(float(gs(*args, **kw)))
        mode = gs.mode
        assert gs.pipeline == (not kind.endswith("nopipe"))
        if gs.pipeline:
            assert opt.has_pending                 # the last update still rides with a forward that never came
            gs.flush()
            assert not opt.has_pending
    else:
        for _ in range(steps):
            loss = model(*args, **kw)
            loss.backward()
            clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            opt.zero_grad()
            losses.append(float(loss))
        mode = "eager"
        opt.flush()          # bf16: step() leaves its update to the next forward; the parameters are read below past the module API (flat.w32)
    if spy is not None:
        spy(model)
    used = model.used_parameter_names()
    final = {n: model.flat.w32(n).detach().float().cpu() for n in (used[0], used[3], used[len(used) // 2], used[-1])}
    red = model._reducer
    st = next(iter(model._steps.values()))
    info = dict(mode=mode, calls=0 if red is None else red.calls, bytes=0 if red is None else red.bytes_reduced,
                points=getattr(st, "exchange_points", None), total=model.flat.total,
                sparse=getattr(st, "sparse_exchange", None),
                nseg=None if not kind == "loopback_graph" else
                [s[0] for s in st.backward_plan(True)._segments])
    return losses, final, info


@pytest.mark.parametrize("case", ["joint_full", "align_small", "caption_small", "pretrain_small"])
def test_graphed_and_data_parallel_schedules_match_eager(case):
    """The hipGraph replays (whole-step graph; captured segments around host-issued gradient exchange points) and the
    data-parallel bucket schedule compute what the eager single-GPU loop computes.  Loopback reducer: identity
    exchanges with the RCCL path's stream/event choreography (univl_amd.parallel.BucketReducer)."""
    ref_l, ref_p, _ = _train("eager", case)
    sparse_word = case in ("joint_full", "align_small")     # the token gather is the word table's only gradient source
    for kind in ("graph", "graph_nopipe", "loopback_eager", "loopback_graph"):
        l, p, info = _train(kind, case)
        # fp32 atomics (split-K, bias / LayerNorm gradients) make runs differ in the last bits only
        np.testing.assert_allclose(l, ref_l, rtol=2e-4, atol=2e-5, err_msg=kind)
        for n in ref_p:
            assert max_abs(p[n], ref_p[n]) < 2e-5, (kind, n)
        if kind in ("graph", "graph_nopipe"):
            assert info["mode"] == "whole"
        else:
            # every used gradient element is exchanged exactly once per step, in few large pieces
            per_step = info["bytes"] / 5
            covered = sum(e - s for cut in info["points"] for s, e in cut)
            # dense slices + the word-embedding gradient as (ids, rows) of the batch's tokens instead of 94 MB
            if sparse_word:
                assert info["sparse"] is not None and info["sparse"]["bytes"] < 4e6
                assert per_step == covered * 4 + info["sparse"]["bytes"]
            else:                                    # tied decoder / MLM head: the table is exchanged densely
                assert info["sparse"] is None and per_step == covered * 4
            flat_ranges = sorted(r for cut in info["points"] for r in cut)
            assert all(a[1] <= b[0] for a, b in zip(flat_ranges, flat_ranges[1:]))          # no overlap
            assert 1 <= len(info["points"]) <= 10
        if kind == "loopback_graph":
            assert info["mode"] == "segmented"
            assert info["nseg"].count("eager") == len(info["points"]) + (2 if sparse_word else 1)   # exchanges (+ token gather) + join
            assert info["nseg"].count("graph") >= len(info["points"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_unchanged_training_loop_switches_to_graph_replay(dtype):
    """The eager loop of main_task_retrieval.py:333-352: after a few iterations the forward / backward plans are replayed
    as hipGraphs (UniVL._run_plan) and keep producing what the plain enqueue produces.  bf16 (round 5): optimizer.step() leaves its
    BertAdam update to the model's NEXT forward, whose products carry it as extra workgroups (the riding update of the captured step,
    here without any GraphedTrainStep) -- losses AND parameters of every iteration equal, bit for bit in this module's deterministic
    mode, those of the loop that applies every update immediately; reading the parameters through the module applies a pending update."""
    def run(auto_graph, ride):
        cfg, rows, dseed = case_config("joint_small")
        model, P = build(cfg, dtype)
        model.auto_graph, model.auto_ride = auto_graph, ride
        model.train()
        opt = BertAdam(model.parameters(), lr=1e-4, warmup=-1, t_total=-1, weight_decay=0.01, max_grad_norm=1.0)
        batch = O.synthetic_batch(cfg, rows, seed=dseed)
        losses = []
        for _ in range(7):
            loss = call(model, batch)
            loss.backward()
            clip_grad_norm_(model.parameters(), 1.0)     # the reference's own call (main_task_retrieval.py:347): in-loop parameters()
            opt.step()
            opt.zero_grad()
            losses.append(float(loss))
        st = next(iter(model._steps.values()))
        pending = opt.has_pending
        final = {n: p.detach().clone() for n, p in model.named_parameters()}       # (applies the pending update)
        assert not opt.has_pending
        return losses, st, final, pending, getattr(model, "auto_ride_count", 0)
    l_graph, st, p_graph, pend, rides = run(True, True)
    l_eager, st_e, p_eager, pend_e, rides_e = run(False, False)
    assert st.fwd._segments is not None and st.backward_plan(True)._segments is not None
    assert [s[0] for s in st.fwd._segments] == ["graph"]
    assert st_e.fwd._segments is None and not pend_e and rides_e == 0
    if dtype == torch.bfloat16:
        assert pend and rides == 6                 # every forward but the first applied the update of the iteration before it
        assert l_graph == l_eager
        assert all(torch.equal(p_graph[n], p_eager[n]) for n in p_eager), [n for n in p_eager if not torch.equal(p_graph[n], p_eager[n])][:5]
    else:
        assert not pend and rides == 0             # fp32 compute: the update is applied by step() itself
        np.testing.assert_allclose(l_graph, l_eager, rtol=2e-4, atol=2e-5)


def _loop_model(dtype=torch.bfloat16, lr=1e-3):
    cfg, rows, dseed = case_config("joint_small")
    model, P = build(cfg, dtype)
    model.train()
    opt = BertAdam(model.parameters(), lr=lr, warmup=-1, t_total=-1, weight_decay=0.01, max_grad_norm=1.0)
    return cfg, model, opt, O.synthetic_batch(cfg, rows, seed=dseed)


def _one_iteration(model, opt, batch):
    loss = call(model, batch)
    loss.backward()
    clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    opt.zero_grad()
    return float(loss)


def test_forward_that_fails_after_adopting_the_pending_update_hands_it_back():
    """optimizer.step() left its update to the next forward; that forward raises (here: a batch the step cannot load) before any
    launch of the update went out.  The update must be pending again -- applied exactly once by the next forward, never twice, never
    dropped -- so a loop that skips the bad batch ends bit-identical to the loop that never saw it."""
    def run(with_bad_batch):
        cfg, model, opt, batch = _loop_model()
        losses = []
        for it in range(6):
            if with_bad_batch and it == 3:
                assert opt.has_pending and opt._auto_deferred
                bad = dict(batch)
                bad["video"] = batch["video"][:, :, :, :17]          # wrong feature width: steps.*.load refuses it
                with pytest.raises(Exception):
                    call(model, bad)
                assert opt.has_pending and opt._auto_deferred and model._rider_update is None
            losses.append(_one_iteration(model, opt, batch))
        final = {n: p.detach().clone() for n, p in model.named_parameters()}
        return losses, final, getattr(model, "auto_ride_count", 0)
    l_bad, p_bad, rides_bad = run(True)
    l_ok, p_ok, rides_ok = run(False)
    assert rides_bad == rides_ok == 5
    assert l_bad == l_ok
    assert all(torch.equal(p_bad[n], p_ok[n]) for n in p_ok), [n for n in p_ok if not torch.equal(p_bad[n], p_ok[n])][:5]


def test_load_state_dict_and_dirty_marks_with_an_update_pending():
    """with optimizer.step()'s update still pending, (1) model.load_state_dict() must end with exactly the loaded weights in the
    fp32 master AND in the bf16 shadow the next forward reads (the pending update belongs to the old weights: it is applied first and then
    overwritten), (2) optimizer.load_state_dict() must not replace the moments under a pending update, (3) a shadow marked dirty stays
    dirty through a flush / an adoption: tensors outside the update are refreshed from the fp32 master."""
    cfg, model, opt, batch = _loop_model()
    for _ in range(4):
        _one_iteration(model, opt, batch)
    assert opt.has_pending
    P = O.procedural_params(cfg, 3)
    sd = dict(P)
    for alias, owner in O.tied_aliases(cfg).items():
        sd[alias] = P[owner]
    model.load_state_dict(sd, strict=True)
    assert not opt.has_pending and model.flat.shadow_valid is False
    loss = call(model, batch)
    fl = model.flat
    assert torch.equal(fl.p16.float(), fl.p32.to(torch.bfloat16).float())                  # shadow == rounded master, every tensor
    for n, p in model.named_parameters():
        assert torch.equal(p.detach().cpu(), P[n]), n
    # the same model loaded into a FRESH instance gives the same loss, bit for bit (deterministic mode)
    cfg2, model2, opt2, _ = _loop_model()
    model2.load_state_dict(sd, strict=True)
    assert float(call(model2, batch)) == float(loss)
    # (2): the optimizer's own load with an update pending -- the update lands first (with the moments it was prepared for), then the
    # loaded moments replace them
    loss.backward()
    clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    assert opt.has_pending
    before = {n: p.detach().clone() for n, p in torch.nn.Module.named_parameters(model)}      # base-class walk: flushes nothing
    _one_iteration(model2, opt2, batch)
    sd2 = opt2.state_dict()
    opt.load_state_dict(sd2)
    assert not opt.has_pending
    after = {n: p.detach().clone() for n, p in torch.nn.Module.named_parameters(model)}
    assert any(not torch.equal(before[n], after[n]) for n in after)                            # the pending update was applied, once
    m2 = {i: st["next_m"] for i, st in sd2["state"].items()}
    for i, st in opt.state_dict()["state"].items():
        assert torch.equal(st["next_m"].cpu(), m2[i].cpu())
    # (3): dirty shadow + pending update -> the next forward refreshes every tensor, also those the update does not touch
    loss = call(model, batch)
    loss.backward()
    opt.step()
    with torch.no_grad():
        fl.p32[fl.index["bert.pooler.dense.weight"][0]] += 1.0          # a tensor without gradient on this path: outside the update
    model._flat.shadow_valid = False                                      # what mark_params_dirty records (without its flush)
    call(model, batch)
    assert torch.equal(fl.p16.float(), fl.p32.to(torch.bfloat16).float())


def test_gradient_accumulation_replays_both_forward_graphs_without_recapture():
    """with gradient_accumulation_steps > 1 on one GPU (README pretrain commands: 16 / 60) the forward after step() carries the
    riding update and the following ones do not.  Both variants of the forward plan are captured ONCE and replayed: no capture after
    warm-up, and the result equals the loop that applies every update immediately."""
    def run(ride):
        cfg, model, opt, batch = _loop_model()
        model.auto_ride = ride
        caps, losses = [], []
        for it in range(12):
            loss = call(model, batch)
            (loss / 2).backward()
            if it % 2 == 1:
                clip_grad_norm_(model.parameters(), 1.0)
                opt.step()
                opt.zero_grad()
            losses.append(float(loss))
            caps.append(getattr(model, "graph_captures", 0))
        return losses, caps, {n: p.detach().clone() for n, p in model.named_parameters()}, model
    l_r, caps, p_r, model = run(True)
    l_e, _, p_e, _ = run(False)
    st = next(iter(model._steps.values()))
    assert st.fwd._segments is not None and len(st.fwd.__dict__.get("_seg_cache", {})) >= 1     # both signatures alive
    assert caps[-1] == caps[7], caps              # warm-up: 3 eager calls + the first capture of either variant; nothing after it
    assert model.auto_ride_count == 5
    assert l_r == l_e
    assert all(torch.equal(p_r[n], p_e[n]) for n in p_e), [n for n in p_e if not torch.equal(p_r[n], p_e[n])][:5]


def test_bert_adam_checkpoint_resume_and_state_dict_layout(tmp_path):
    """optimizer.state_dict() -> torch.save -> load_state_dict on a fresh optimizer (main_pretrain.py:266-273, 389): the
    resumed run continues exactly like the uninterrupted one (moments, per-parameter step counters, warmup schedule)."""
    cfg, rows, dseed = case_config("joint_small")
    batch = O.synthetic_batch(cfg, rows, seed=dseed)

    def make():
        model, _ = build(cfg, torch.float32)
        model.train()
        opt = BertAdam(model.parameters(), lr=1e-3, warmup=0.5, t_total=8, schedule="warmup_linear", weight_decay=0.01, max_grad_norm=1.0)
        return model, opt

    def one(model, opt):
        loss = call(model, batch)
        loss.backward()
        clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        return float(loss)

    m1, o1 = make()
    cont = [one(m1, o1) for _ in range(5)]
    m2, o2 = make()
    first = [one(m2, o2) for _ in range(3)]
    assert first == pytest.approx(cont[:3], rel=1e-5, abs=1e-6)
    osd, msd = o2.state_dict(), m2.state_dict()
    st0 = osd["state"][0]
    assert set(st0) == {"step", "next_m", "next_v"} and st0["step"] == 3          # the reference's keys (optimization.py:124-129)
    p0 = next(iter(m2.parameters()))
    assert tuple(st0["next_m"].shape) == tuple(p0.shape)
    torch.save(dict(opt=osd, model=msd), os.path.join(tmp_path, "ck.bin"))
    ck = torch.load(os.path.join(tmp_path, "ck.bin"), map_location="cpu")
    m3, o3 = make()
    one(m3, o3)                                   # an optimizer that already owns flat moment buffers
    m3.load_state_dict(ck["model"])
    o3.load_state_dict(ck["opt"])
    resumed = [one(m3, o3) for _ in range(2)]
    assert resumed == pytest.approx(cont[3:], rel=2e-4, abs=2e-5), (resumed, cont)
    assert o3.state_dict()["state"][0]["step"] == 5 and sorted(set(o3.get_lr())) == sorted(set(o1.get_lr()))
    # loading before the first step (no flat buffers yet): the loaded tensors are migrated at the first step
    m4, o4 = make()
    m4.load_state_dict(ck["model"])
    o4.load_state_dict(ck["opt"])
    assert one(m4, o4) == pytest.approx(cont[3], rel=2e-4, abs=2e-5)


@pytest.mark.parametrize("schedule", ["warmup_cosine", "warmup_constant", "warmup_linear"])
def test_warmup_schedules_on_device(schedule):
    """optimization.py:26-50: the per-step learning rate factor the update kernel computes equals the reference's
    schedule function (its cosine form written with math.cos; the reference's torch.cos(float) cannot run)."""
    from univl_amd.optimization import SCHEDULES
    cfg, rows, dseed = case_config("joint_small")
    model, _ = build(cfg, torch.float32)
    model.train()
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    lr, T, wu = 1e-2, 6, 0.34
    opt = BertAdam(model.parameters(), lr=lr, warmup=wu, t_total=T, schedule=schedule, weight_decay=0.0, max_grad_norm=-1)
    n = "visual.encoder.layer.0.intermediate.dense.weight"
    p = dict(model.named_parameters())[n]
    for k in range(5):
        call(model, batch).backward()
        before = p.detach().clone()
        g = p.grad.detach().clone()
        st = opt.state.get(p, {})
        m = st["next_m"].clone() if st else torch.zeros_like(p)
        v = st["next_v"].clone() if st else torch.zeros_like(p)
        opt.step()
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        want = before - lr * SCHEDULES[schedule](k / T, wu) * (m / (v.sqrt() + 1e-6))
        assert max_abs(p, want) < 1e-6 + 1e-4 * float((want - before).abs().max()), (schedule, k)
        # get_lr() reports the rate of the NEXT step for parameters that hold a gradient (optimization.py:86-101)
        assert sorted(set(opt.get_lr()))[0] == pytest.approx(lr * SCHEDULES[schedule]((k + 1) / T, wu))
        opt.zero_grad()


def test_shaped_true_and_pretrain_without_captions():
    """get_sequence_visual_output(shaped=True) (modeling.py:299-313: inputs already flattened, video already normalised)
    and forward() on the pretrain path without captions (modeling.py:238: the decoder loss is skipped)."""
    cfg, rows, dseed = case_config("pretrain_small")
    model, P = build(cfg, torch.float32)
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    b = {k: v.to(DEV) for k, v in batch.items()}
    model.eval()
    with torch.no_grad():
        seq, vis = model.get_sequence_visual_output(b["input_ids"], b["token_type_ids"], b["attention_mask"], b["video"], b["video_mask"])
        flat = lambda t: t.view(-1, t.shape[-1])
        vnorm = O.normalize_video(batch["video"], P).to(DEV)           # modeling.py:88-92 on the CPU oracle
        seq2, vis2 = model.get_sequence_visual_output(flat(b["input_ids"]), flat(b["token_type_ids"]), flat(b["attention_mask"]),
                                                      vnorm.view(-1, cfg.max_frames, cfg.video_dim), flat(b["video_mask"]), shaped=True)
    assert max_abs(seq, seq2) < 1e-4 and max_abs(vis, vis2) < 1e-4
    model.train()
    full = call(model, batch)
    st = model._steps[("pretrain", rows * cfg.n_pair, cfg.max_words, cfg.max_frames, True)]
    dec = float(st.decoder.loss)
    full = float(full)
    model.zero_grad(set_to_none=True)
    nocap = model(b["input_ids"], b["token_type_ids"], b["attention_mask"], b["video"], b["video_mask"],
                  pairs_masked_text=b["pairs_masked_text"], pairs_token_labels=b["pairs_token_labels"],
                  masked_video=b["masked_video"], video_labels_index=b["video_labels_index"])
    nocap.backward()
    assert abs(float(nocap) - (full - dec)) < 1e-3 * max(1.0, abs(full))
    params = dict(model.named_parameters())
    assert all(p.grad is None for n, p in params.items() if n.startswith("decoder."))
    assert params["cls.predictions.transform.dense.weight"].grad is not None


def test_sparse_word_table_bookkeeping_matches_dense_clear(ab):
    """Retrieval configurations clear only the word-table rows the previous backward wrote and take the table's gradient
    norm from those rows (engine.FlatParams.word_rows): same gradients, clip norm and parameters as the dense 94 MB clear +
    streaming norm (sparse_rows=0), over changing batches, gradient accumulation and a foreign dense write."""
    cfg, rows, dseed = case_config("joint_small")
    batches = [O.synthetic_batch(cfg, rows, seed=dseed + k) for k in range(4)]
    wname = "bert.embeddings.word_embeddings.weight"

    def run(sparse):
        ab(sparse_rows=1 if sparse else 0)
        model, _ = build(cfg, torch.float32)
        model.train()
        opt = BertAdam(model.parameters(), lr=1e-4, warmup=-1, t_total=-1, weight_decay=0.01, max_grad_norm=1.0)
        trace = []
        for k, b in enumerate(batches):
            call(model, b).backward()
            if k == 1:                                   # gradient accumulation: a second backward before the step
                call(model, batches[0]).backward()
            if k == 2:                                   # somebody edits the table gradient through torch
                dict(model.named_parameters())[wname].grad[123].add_(1.0)
            gw = dict(model.named_parameters())[wname].grad
            total = clip_grad_norm_(model.parameters(), 1.0)
            trace.append((float(total), float(gw.double().norm()), int((gw.abs().sum(1) > 0).sum())))
            opt.step()
            opt.zero_grad()
        fl = model.flat
        assert (getattr(fl, "_word_rows", None) is not None) == sparse
        return trace, fl.w32(wname).detach().cpu().clone()

    t1, w1 = run(True)
    t0, w0 = run(False)
    for a, b in zip(t1, t0):
        assert a[0] == pytest.approx(b[0], rel=1e-5) and a[1] == pytest.approx(b[1], rel=1e-5) and a[2] == b[2], (t1, t0)
    # Not bit-equal: the two runs order their fp32 atomics (embedding scatter-add, split-K, LayerNorm column sums) differently,
    # and BertAdam's m / sqrt(v) turns a 1e-7 relative gradient difference into ~1e-7 absolute on a parameter (measured over
    # the round-2 sessions: 0.4e-7 ... 1.5e-7).  One BertAdam step moves a touched element by lr * |m| / sqrt(v) ~ 3e-4 here, so
    # a row the sparse bookkeeping forgot to clear or to count would show up ~1000x above this gate.
    assert max_abs(w1, w0) < 5e-7


@pytest.mark.parametrize("ride", ["1", "0"])
@pytest.mark.parametrize("name", ["joint_full", "caption_small", "pretrain_small"])
def test_weight_gradient_packaging_matches_golden(golden_dir, name, ride, ab):
    """wgrad_ride=1 (default): every encoder weight-gradient GEMM goes out in the launch of the dgrad GEMM that consumes the
    same upstream gradient (univl_gemm_pair); =0: the layer's grouped launch at the end of its chain.  Both give the loss and
    the gradients of the reference's golden vectors inside the ordinary bf16 gates, and the plan really is what the switch says."""
    ab(wgrad_ride=int(ride))
    if ride == "0":                     # the default form is what every other golden test of this file runs
        test_forward_backward_vs_reference_golden(golden_dir, name, torch.bfloat16)
    cfg, rows, dseed = case_config(name)
    model, _ = build(cfg, torch.bfloat16)
    model.train()
    call(model, O.synthetic_batch(cfg, rows, seed=dseed)).backward()
    st = next(iter(model._steps.values()))
    kinds = [op[3] for op in st.backward_plan(True).ops]
    if ride == "1":
        # (round 5: the attention-output projection's pair became the fused attention backward, which carries that weight gradient)
        layers = cfg.text_num_hidden_layers + cfg.visual_num_hidden_layers
        assert kinds.count("univl_gemm_pair") + kinds.count("univl_gemm_pair_ln") + kinds.count("univl_attention_bwd_fused") >= 4 * layers
        assert kinds.count("univl_attention_bwd_fused") >= layers      # (decoder / cross stacks keep univl_attention_bwd where Sq != Sk or > 64)
    else:
        assert kinds.count("univl_gemm_pair") == 0 and kinds.count("univl_gemm_group") >= cfg.text_num_hidden_layers + cfg.visual_num_hidden_layers


@pytest.mark.parametrize("case", ["joint_full", "pretrain_small", "caption_small", "align_small"])
def test_adam_update_riding_with_the_next_forward_matches_eager(case):
    """GraphedTrainStep(pipeline_optimizer=True): the BertAdam update of iteration t is applied by the forward
    of iteration t + 1 -- embedding tables, vectors and each stack's first layer as launches in front of it, the other layers'
    chunks as extra workgroups of the forward products of the layer before (univl_gemm_rider).  In deterministic mode the losses
    and the parameters are BIT-IDENTICAL to the eager loop's, iteration by iteration (same arithmetic per element, same gradients);
    the last update stays pending until flush()."""
    ref_l, ref_p, _ = _train("eager", case, dtype=torch.bfloat16)
    l, p, info = _train("graph", case, dtype=torch.bfloat16)
    assert info["mode"] == "whole"
    assert l == ref_l, (l, ref_l)
    for n in ref_p:
        assert torch.equal(p[n], ref_p[n]), (n, max_abs(p[n], ref_p[n]))


def test_adam_update_riding_in_rectangular_tile_products_matches_the_plain_update(ab):
    """From 1536 tokens on the forward products run on the 64 x 128 tile; round 5 gave that tile a rider kernel
    (gemm_adam_rect_kernel, univl_gemm_rider) and lets the host spread a layer's chunks over the products the library says carry
    (univl_gemm_rider_fits) -- before, product and update were enqueued one after the other.  joint_b32 (32 pairs, 1536 tokens per
    stack), deterministic mode: losses and parameters BIT-IDENTICAL to the loop whose BertAdam.step() applies the update itself."""
    import ctypes as C
    from univl_amd import _lib
    ab(adam_ride="0")
    ref_l, ref_p, _ = _train("eager", "joint_b32", steps=3, dtype=torch.bfloat16)
    ab(adam_ride=None)
    _probe = {}

    def spy(model):
        st = next(iter(model._steps.values()))
        riders = [op[2] for op in st.fwd.ops if op[0] == "rider"]
        _probe["riders"] = [(key, slot, n) for _, key, slot, n in riders]
        _probe["fits"] = [_lib.lib().univl_gemm_rider_fits(C.byref(d)) for d, _, _, _ in riders]
        # at 1536 tokens the fused attention forward still runs and carries its share of the layer's chunks (slot 0)
        _probe["riders"] += [(op[2][2], op[2][3], op[2][4]) for op in st.fwd.ops if op[0] == "attn_fwd_fused" and op[2][2] is not None]
    l, p, info = _train("graph", "joint_b32", steps=3, dtype=torch.bfloat16, spy=spy)
    assert info["mode"] == "whole"
    assert _probe["riders"] and all(f == 1 for f in _probe["fits"]), _probe
    by_key = {}
    for key, slot, n in _probe["riders"]:
        by_key.setdefault(key, []).append((slot, n))
    cfg = case_config("joint_b32")[0]
    assert len(by_key) == (cfg.text_num_hidden_layers - 1) + (cfg.visual_num_hidden_layers - 1)
    for key, sl in by_key.items():                      # every layer's chunk range is covered exactly once by the launches that carry
        assert sorted(s_ for s_, _ in sl) == list(range(sl[0][1])) and 1 <= sl[0][1] <= 4, (key, sl)
    assert l == ref_l, (l, ref_l)
    for n in ref_p:
        assert torch.equal(p[n], ref_p[n]), (n, max_abs(p[n], ref_p[n]))


def _grad_errors(model, batch, g):
    model.train()
    model.zero_grad()
    loss = call(model, batch)
    loss.backward()
    names = [str(s) for s in g["grad_names"]]
    G = {k: None for k in GATES[torch.bfloat16]}
    G["gnorm"] = 1.0
    err, _ = compare_gradients(dict(model.named_parameters()), names, g["grad_norms"], g["grad_samples"], g["grad_top_index"], g["grad_top_samples"], G, False)
    err["loss"] = abs(float(loss) - float(g["loss"])) / max(1.0, abs(float(g["loss"])))
    return err


@pytest.mark.parametrize("name", ["joint_full", "joint_b16", "pretrain_full"])
def test_operand_pairs_tighten_the_bf16_step(golden_dir, name):
    """model.operand_pairs = 'xw' (round 6): the forward products of stacks up to 768 tokens take both operands as bf16 PAIRS
    (UnivlGemm.A_lo / B_lo; every producer of such an activation writes the lo half, the optimizer keeps the lo half of the weight shadow).
    Against the real reference's fp32 gradients the median per-tensor error drops by a third or more, the global error is inside
    north_star's 1e-2 (the plain step sits AT it) and every statistic stays inside the unchanged gates; the plans really carry the lo halves; setting the attribute back restores the plain plans bit for bit.
    (Measured: joint_full gmedian 8.9e-3 -> 5.7e-3, gglobal 9.8e-3 -> 9.0e-3 -- that statistic is 72 % one tensor, the token-type table;
    pretrain_full gglobal 1.19e-2 -> 7.3e-3, gmedian 1.07e-2 -> 5.1e-3; +22 % step time at 4 pairs: profiles/r06p_*.)"""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg, rows, dseed = case_config(name)
    model, P = build(cfg, torch.bfloat16)
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    assert model.operand_pairs == ""
    e0 = _grad_errors(model, batch, g)
    g0 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    assert model.flat.p16lo is None
    model.operand_pairs = "xw"
    e1 = _grad_errors(model, batch, g)
    fl = model.flat
    assert fl.p16lo is not None and torch.equal(fl.p16lo, (fl.p32 - fl.p16.float()).to(torch.bfloat16))
    st = [s_ for s_ in model._steps.values() if hasattr(s_, "fwd")][0]
    descs = [d for ds in st.fwd.descs.values() for d in (ds if isinstance(ds, (list, tuple)) else [ds]) if hasattr(d, "B_lo")]
    assert sum(1 for d in descs if d.B_lo) >= 4 * (cfg.text_num_hidden_layers + cfg.visual_num_hidden_layers)
    assert sum(1 for d in descs if d.A_lo) >= 4 * (cfg.text_num_hidden_layers + cfg.visual_num_hidden_layers) - 2
    print("[operand pairs %s] plain %s | pairs %s" % (name, " ".join("%s=%.2e" % kv for kv in sorted(e0.items())), " ".join("%s=%.2e" % kv for kv in sorted(e1.items()))))
    _record(name + "@pairs_xw", torch.bfloat16, **e1)
    assert e1["gmedian"] < 0.72 * e0["gmedian"], (e0, e1)
    assert e1["gglobal"] < e0["gglobal"] and e1["loss"] < 1e-3
    # north_star's bf16 tolerance on the relative gradient error, met with margin in this mode at every BASELINE configuration up to 16 pairs
    # (cfg1/2 0.90e-2, cfg3's share 0.71e-2, cfg5 0.73e-2; cfg4 is at 0.43e-2 without pairs) -- the default mode trades it for 22 % step time
    assert e1["gglobal"] <= 1.0e-2, e1["gglobal"]
    check_gates(name + "@pairs_xw", e1, gates_for(name, torch.bfloat16))
    with pytest.raises(ValueError):
        model.operand_pairs = "y"
    model.operand_pairs = ""
    _grad_errors(model, batch, g)
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, g0[n]), n               # deterministic mode: the plain plans again, bit for bit


def test_operand_pairs_follow_the_optimizer(ab):
    """The lo half of the weight shadow is kept by the fused BertAdam update -- eager, and riding in the next forward's products -- so that
    hi + lo tracks the fp32 master after every step; losses with pairs stay within bf16 noise of the plain loop's."""
    ab(pairs="xw")

    def spy(model):
        fl = model.flat
        assert fl.p16lo is not None and model.operand_pairs == "xw"
        assert torch.equal(fl.p16, fl.p32.to(torch.bfloat16)) and torch.equal(fl.p16lo, (fl.p32 - fl.p16.float()).to(torch.bfloat16))
    l1, p1, _ = _train("eager", "joint_small", steps=3, dtype=torch.bfloat16, spy=spy)
    l2, p2, info = _train("graph", "joint_small", steps=3, dtype=torch.bfloat16, spy=spy)
    assert l1 == l2, (l1, l2)                                  # riding update == eager update, bit for bit, with pairs on
    for n in p1:
        assert torch.equal(p1[n], p2[n]), n
    ab(pairs=None)
    l0, _, _ = _train("eager", "joint_small", steps=3, dtype=torch.bfloat16)
    assert l0 != l1 and all(abs(a - b) < 2e-3 * max(1.0, abs(a)) for a, b in zip(l0, l1)), (l0, l1)


def test_lo_shadow_is_kept_where_the_rider_kernel_does_not_carry_it():
    """The rider kernel of the 64 x 128 tile (1536+ tokens) is built without the lo-shadow write (adam_chunk<.., LO = false>: two spilled
    registers in a <= 80-VGPR kernel); univl_gemm_rider therefore runs product and update as two launches when the update carries
    UnivlAdam.p16_lo.  A model whose lo shadow exists (operand pairs were used at a smaller batch) and which trains at 32 pairs: losses and
    parameters of the riding step are bit-identical to the eager loop, and hi + lo still tracks the fp32 master."""
    from univl_amd.graphed import GraphedTrainStep
    cfg, rows, dseed = case_config("joint_b32")
    b = {k: v.to(DEV) for k, v in O.synthetic_batch(cfg, rows, seed=dseed).items()}
    args = (b["input_ids"], b["token_type_ids"], b["attention_mask"], b["video"], b["video_mask"])
    kw = dict(pairs_masked_text=b["pairs_masked_text"], pairs_token_labels=b["pairs_token_labels"], masked_video=b["masked_video"],
              video_labels_index=b["video_labels_index"])

    def run(graph):
        model, _ = build(cfg, torch.bfloat16)
        model.train()
        assert model.flat.ensure_lo() and model.operand_pairs == ""
        opt = BertAdam(model.parameters(), lr=1e-5, warmup=-1, t_total=-1, weight_decay=0.01, max_grad_norm=1.0)
        losses = []
        if graph:
            gs = GraphedTrainStep(model, opt, max_grad_norm=1.0, warmup=1, pipeline_optimizer=True)
            for _ in range(3):
                losses.append(float(gs(*args, **kw)))
            gs.flush()
        else:
            for _ in range(3):
                loss = model(*args, **kw)
                loss.backward()
                clip_grad_norm_(model.parameters(), 1.0)
                opt.step()
                opt.zero_grad()
                losses.append(float(loss))
            opt.flush()
        fl = model.flat
        assert torch.equal(fl.p16, fl.p32.to(torch.bfloat16)) and torch.equal(fl.p16lo, (fl.p32 - fl.p16.float()).to(torch.bfloat16))
        return losses, fl.p32.detach().clone()

    l0, p0 = run(False)
    l1, p1 = run(True)
    assert l0 == l1, (l0, l1)
    assert torch.equal(p0, p1)


def test_lazy_word_rows_update_is_bit_identical(ab):
    """UnivlAdam.row_flags (adam_lazy_rows, default on): chunks of word-table rows that never held a gradient take the
    weight-decay-only form of the BertAdam update (10 instead of 30 bytes per parameter).  Same bits as the full update, over
    changing batches (the set of touched rows grows), gradient accumulation and a foreign write to the table's gradient."""
    cfg, rows, dseed = case_config("joint_small")
    batches = [O.synthetic_batch(cfg, rows, seed=dseed + k) for k in range(4)]
    wname = "bert.embeddings.word_embeddings.weight"

    def run(lazy):
        ab(adam_lazy_rows=1 if lazy else 0)
        model, _ = build(cfg, torch.bfloat16)
        model.train()
        opt = BertAdam(model.parameters(), lr=1e-3, warmup=-1, t_total=-1, weight_decay=0.01, max_grad_norm=1.0)
        flagged = []
        for k, b in enumerate(batches):
            call(model, b).backward()
            if k == 1:
                call(model, batches[0]).backward()
            if k == 3:
                dict(model.named_parameters())[wname].grad[123].add_(1.0)
            clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            opt.zero_grad()
            fl = model.flat
            flagged.append(None if fl.word_ever is None else int(fl.word_ever.sum()))
        fl = model.flat
        return fl.w32(wname).detach().clone(), fl.wop(wname).detach().clone(), opt.state[dict(model.named_parameters())[wname]]["next_v"].clone(), flagged

    w1, s1, v1, f1 = run(True)
    w0, s0, v0, f0 = run(False)
    assert f0 == [None] * 4
    tokens = [int(torch.unique(b["input_ids"]).numel()) for b in batches]
    assert f1[0] == tokens[0] and f1[0] <= f1[1] <= f1[2] < cfg.vocab_size      # only the touched rows are flagged ...
    assert f1[3] == cfg.vocab_size                                               # ... until somebody edits the gradient behind our back
    assert torch.equal(w1, w0) and torch.equal(s1, s0) and torch.equal(v1, v0)
    untouched = (v1.abs().sum(1) == 0)
    assert int(untouched.sum()) > cfg.vocab_size // 2                            # most rows took the shortcut at least until step 3
