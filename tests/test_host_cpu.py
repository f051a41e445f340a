"""CPU-side checks (no GPU, no compute calls into the HIP library): the C-ABI library loads and exports every symbol
include/univl_hip.h declares, the ctypes struct mirrors match the library's sizeof(), the static plans build, the
flat parameter layout fuses q/k/v and covers every parameter once, and the product path refuses to run on CPU."""
import argparse
import os
import re

import pytest
import torch

import univl_oracle as O
from univl_amd import UniVL, BertAdam, _lib
from univl_amd.engine import FlatParams
from univl_amd.parallel import layer_buckets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(dt="bf16", **kw):
    cfg = O.OracleConfig(**dict(dict(batch_size=2, text_num_hidden_layers=2, visual_num_hidden_layers=1, max_words=16, max_frames=16), **kw))
    ns = argparse.Namespace(**cfg.to_dict(), local_rank=0, compute_dtype=dt)
    m = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base", task_config=ns)
    return m, cfg


def test_library_exports_every_declared_symbol():
    L = _lib.lib()          # also verifies sizeof() of the 7 ABI structs against the ctypes mirrors
    header = open(os.path.join(ROOT, "include", "univl_hip.h")).read()
    declared = set(re.findall(r"\b(univl_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), "libunivl_hip.so does not export " + name
    assert declared == set(_lib.EXPORTED)
    assert L.univl_version() >= 100


def test_state_dict_keys_match_reference_inventory():
    m, cfg = _model()
    assert list(dict(m.named_parameters())) == list(O.param_shapes(cfg))
    for n, p in m.named_parameters():
        assert tuple(p.shape) == O.param_shapes(cfg)[n], n


def test_flat_layout_and_plans_build_on_cpu():
    m, cfg = _model("fp32")
    m.load_state_dict(O.procedural_params(cfg, 0))
    fl = FlatParams(list(m.named_parameters()), "cpu", torch.float32)
    m._flat, m._seed_dev = fl, torch.zeros(1, dtype=torch.int64)
    # parameters are views of the flat buffer and kept their values
    P = O.procedural_params(cfg, 0)
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), P[n]), n
        lo = fl.p32.data_ptr()
        assert lo <= p.data_ptr() < lo + fl.total * 4
    # q/k/v are contiguous -> one fused [2304,768] operand
    a = "bert.encoder.layer.1.attention.self."
    w = fl.w32_fused([a + "query.weight", a + "key.weight", a + "value.weight"])
    assert w.shape == (2304, 768) and torch.equal(w[768:1536], P[a + "key.weight"])
    m.train()
    from univl_amd.steps import build_step
    st = build_step(m, "joint", 2, 16, 16, True)
    # (the similarity head's backward is ONE launch since round 3, so the backward plan is no longer the longer one)
    assert len(st.fwd) >= 3 * 7 + 4 + 4 and len(st.backward_plan(True)) >= 3 * 7 + 4
    # data-parallel buckets: disjoint, cover every parameter that gets a gradient
    b = layer_buckets(fl, m.used_parameter_names())
    sl = sorted(list(b["layers"].values()) + b["tail"])
    for (s0, e0), (s1, e1) in zip(sl, sl[1:]):
        assert e0 <= s1
    for n in m.used_parameter_names():
        o, k, _ = fl.index[n]
        assert any(s <= o and o + k <= e for s, e in sl), n


def test_product_path_fails_loudly_without_gpu():
    m, cfg = _model()
    batch = O.synthetic_batch(cfg, 2, seed=1)
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="HIP device"):
        m.train()
        m(batch["input_ids"], batch["token_type_ids"], batch["attention_mask"], batch["video"], batch["video_mask"])
    with pytest.raises(RuntimeError, match="HIP device"):
        m.get_similarity_logits(torch.zeros(2, 16, 768), torch.zeros(2, 16, 768), batch["attention_mask"], batch["video_mask"])
    opt = BertAdam(m.parameters(), lr=1e-3)
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    with pytest.raises(RuntimeError, match="HIP device"):
        opt.step()


@pytest.mark.parametrize("kw,kind", [(dict(), "joint"), (dict(train_sim_after_cross=True), "align"),
                                     (dict(stage_two=True, task_type="caption", decoder_num_hidden_layers=1), "caption"),
                                     (dict(stage_two=True, do_pretrain=True, use_mil=True, decoder_num_hidden_layers=1), "pretrain")])
def test_every_forward_branch_builds_its_plans(kw, kind):
    """The four branches of UniVL.forward (modeling.py:204-267): inventory equals the reference's, plans build."""
    from univl_amd.steps import build_step
    m, cfg = _model("fp32", **kw)
    assert list(dict(m.named_parameters())) == list(O.param_shapes(cfg))
    assert set(m.state_dict()) == set(O.param_shapes(cfg)) | set(O.tied_aliases(cfg))
    m._flat, m._seed_dev = FlatParams(list(m.named_parameters()), "cpu", torch.float32), torch.zeros(1, dtype=torch.int64)
    m.train()
    assert m.step_kind(True) == kind
    st = build_step(m, kind, 2, 16, 16, True)
    assert len(st.fwd) > 20 and len(st.backward_plan(True)) > 20      # (FT-Joint: the similarity head's backward is one launch)
    assert len(st.backward_plan(False)) > 0


def test_bert_adam_argument_validation():
    m, _ = _model()
    with pytest.raises(ValueError):
        BertAdam(m.parameters(), lr=-1.0)
    with pytest.raises(ValueError):
        BertAdam(m.parameters(), lr=1e-3, warmup=1.5)
    with pytest.raises(ValueError):
        BertAdam(m.parameters(), lr=1e-3, b1=1.0)


def _tile_map(what, nx, ny, nz, gm, a, b=0, c=0):
    import ctypes as C
    out = (C.c_int32 * 3)(a, b, c)
    assert _lib.lib().univl_gemm_tile_map(what, nx, ny, nz, gm, out) == 0
    return out[0], out[1], out[2]


@pytest.mark.parametrize("gm", [0, 4, 8, 16])
def test_gemm_workgroup_to_tile_maps_are_bijections(gm):
    """The XCD-aware / L2-blocked tile order of gemm.hip only re-orders the tiles: every tile of the grid is computed by
    exactly one workgroup, for every grid shape the plans produce (row tiles 1..96, split-K 1..8) and for ragged ones."""
    shapes = [(36, 3, 1), (12, 3, 2), (12, 3, 8), (48, 3, 1), (6, 1, 1), (5, 3, 1), (1, 3, 8), (18, 12, 1), (6, 48, 1),
              (24, 96, 1), (239, 4, 1), (7, 13, 3), (3, 9, 2), (477, 1, 1), (2, 2, 2), (16, 1, 1), (1, 17, 1), (6, 47, 2),
              (6, 96, 1), (12, 48, 1), (6, 192, 1)]      # the last three: half-width tiles (64 x 128, 128 x 64) at 6144 / 12288 rows
    for nx, ny, nz in shapes:
        total = nx * ny * nz
        # plain grid (gemm_kernel): hardware (x, y, z) -> tile
        seen = set()
        for z in range(nz):
            for y in range(ny):
                for x in range(nx):
                    t = _tile_map(0, nx, ny, nz, gm, x, y, z)
                    assert 0 <= t[0] < nx and 0 <= t[1] < ny and 0 <= t[2] < nz, (nx, ny, nz, gm, (x, y, z), t)
                    seen.add(t)
        assert len(seen) == total, (nx, ny, nz, gm)
        # grouped launch (gemm_group_kernel): list position, then the member's local tile
        pos = sorted(_tile_map(1, total, 1, 1, gm, w)[0] for w in range(total))
        assert pos == list(range(total)), (total,)
        local = {_tile_map(2, nx, ny, nz, gm, t) for t in range(total)}
        assert len(local) == total and all(0 <= a < nx and 0 <= b < ny and 0 <= c < nz for a, b, c in local), (nx, ny, nz, gm)
        # either half of a pair / rider launch (gemm_pair_kernel, gemm_adam_kernel): local workgroup -> tile
        half = {_tile_map(3, nx, ny, nz, gm, t) for t in range(total)}
        assert len(half) == total and all(0 <= a < nx and 0 <= b < ny and 0 <= c < nz for a, b, c in half), (nx, ny, nz, gm)
    # workgroups dealt to one XCD (linear id % 8) take a CONTIGUOUS run of the list: that is what keeps the row tiles
    # that share a weight tile behind one L2
    total = 108
    for c in range(8):
        run = [_tile_map(1, total, 1, 1, gm, w)[0] for w in range(c, total, 8)]
        assert run == list(range(run[0], run[0] + len(run)))


@pytest.mark.parametrize("kw,kind", [(dict(), "joint"), (dict(train_sim_after_cross=True), "align"),
                                     (dict(stage_two=True, task_type="caption", decoder_num_hidden_layers=1), "caption"),
                                     (dict(stage_two=True, do_pretrain=True, use_mil=True, decoder_num_hidden_layers=1), "pretrain")])
@pytest.mark.parametrize("fresh", [True, False])
def test_riding_weight_gradients_only_repackage_the_backward_plan(kw, kind, fresh, ab):
    """wgrad_ride (univl_amd/_ab.py): a weight-gradient GEMM goes out in the launch of the dgrad GEMM fed by the same upstream gradient
    (univl_gemm_pair) instead of the layer's grouped launch.  Built on the CPU for every branch of UniVL.forward, the two
    backward plans must contain exactly the same GEMM descriptors (same shapes, flags, parameter / gradient / norm
    pointers), every other launch in the same order, and every pair must join the two halves of ONE nn.Linear: the dgrad
    reads the weight at the flat offset the weight gradient is written to."""
    import ctypes as C
    from univl_amd.steps import build_step
    m, cfg = _model("bf16", **kw)
    fl = FlatParams(list(m.named_parameters()), "cpu", torch.bfloat16)
    m._flat, m._seed_dev = fl, torch.zeros(1, dtype=torch.int64)
    m.train()

    def flat_off(ptr):
        """(buffer, element offset) for pointers into the flat parameter storage, None for workspaces"""
        if not ptr:
            return None
        for tag, t in (("p16", fl.p16), ("p32", fl.p32), ("g32", fl.g32), ("partials", fl.partials)):
            lo = t.data_ptr()
            if lo <= ptr < lo + t.numel() * t.element_size():
                return tag, (ptr - lo) // t.element_size()
        return "ws"

    def rec(d):
        return (d.M, d.N, d.K, d.trans_a, d.trans_b, d.flags, d.ksplit, d.lda, d.ldb, d.ldc, flat_off(d.B), flat_off(d.C32),
                flat_off(d.bias), flat_off(d.dbias), flat_off(d.sumsq), d.sumsq_rows, d.sumsq_stride, bool(d.aux), bool(d.R))

    def canon(plan):
        chain, wgrads, pairs = [], [], []
        for i, op in enumerate(plan.ops):
            kind_, name = op[0], op[3]
            if kind_ == "call" and name == "univl_gemm":
                d = plan.descs[i][0]
                (wgrads if (d.trans_a and d.trans_b and flat_off(d.C32) and flat_off(d.C32)[0] == "g32") else chain).append(("gemm",) + rec(d))
            elif kind_ == "group":
                wgrads += [("gemm",) + rec(d) for d in plan.descs[i]]
            elif kind_ in ("pair", "pair_ln"):
                dg, wg = plan.descs[i]
                chain.append(("gemm",) + rec(dg))
                wgrads.append(("gemm",) + rec(wg))
                pairs.append((dg, wg))
                if kind_ == "pair_ln":           # the LayerNorm backward fed by this dgrad, finished inside the launch (round 4)
                    chain.append(("univl_layernorm_bwd",))
            elif kind_ == "attn_fused":          # round 5: the attention-output dgrad inside the attention backward, its weight gradient riding
                ds = plan.descs[i]
                chain.append(("gemm",) + rec(ds[0]))
                chain.append(("univl_attention_bwd",))
                if len(ds) > 1:
                    wgrads.append(("gemm",) + rec(ds[1]))
                    pairs.append((ds[0], ds[1]))
            elif kind_ in ("call", "py", "eager"):
                chain.append((name,))
        return chain, sorted(wgrads, key=repr), pairs

    ab(wgrad_ride=0)
    base = build_step(m, kind, 2, 16, 16, True).backward_plan(fresh)
    ab(wgrad_ride=None)
    ride = build_step(m, kind, 2, 16, 16, True).backward_plan(fresh)
    c0, w0, p0 = canon(base)
    c1, w1, p1 = canon(ride)
    assert not p0 and len(p1) >= 4 * (cfg.text_num_hidden_layers + cfg.visual_num_hidden_layers)
    assert c0 == c1
    assert w0 == w1 and len(w0) >= len(p1)
    for dg, wg in p1:
        assert not dg.trans_a and dg.trans_b and wg.trans_a and wg.trans_b
        assert flat_off(dg.B)[0] == "p16" and flat_off(wg.C32)[0] == "g32"
        assert flat_off(dg.B)[1] == flat_off(wg.C32)[1], "a pair must be the two halves of one nn.Linear backward"
        assert dg.A == wg.A and dg.K == wg.M and dg.N == wg.N and dg.M == wg.K        # same upstream gradient dY [tokens, out]


def test_adam_rider_plan_builds_on_cpu(ab):
    """FlatParams.adam_ride (set by graphed.GraphedTrainStep): in the forward plan the four products of every text / video layer but the last become rider
    launches keyed by the NEXT layer of the same stack; nothing else changes, and the backward plan is untouched."""
    from univl_amd.steps import build_step
    m, cfg = _model("bf16")
    m._flat, m._seed_dev = FlatParams(list(m.named_parameters()), "cpu", torch.bfloat16), torch.zeros(1, dtype=torch.int64)
    m.train()
    m._flat.adam_ride = False
    base = build_step(m, "joint", 2, 16, 16, True)
    m._flat.adam_ride = True
    ride = build_step(m, "joint", 2, 16, 16, True)
    names0 = [op[3] for op in base.fwd.ops]
    names1 = [op[3] for op in ride.fwd.ops]
    L0_fused = cfg.text_num_hidden_layers + cfg.visual_num_hidden_layers
    assert [n.replace("univl_gemm_rider", "univl_gemm") for n in names1] == names0
    # (desc, key, slot, nslots) of every launch that can carry chunks: plain rider launches and the (product + LayerNorm) launches
    riders = [op[2] for op in ride.fwd.ops if op[0] == "rider"] + [(op[2][0],) + op[2][3:] for op in ride.fwd.ops if op[0] == "gemm_ln" and op[2][3] is not None]
    # round 5: the q | k | v projection of a layer is computed inside the attention forward launch, which carries its quarter of the chunks
    riders += [(op[2][1],) + op[2][2:] for op in ride.fwd.ops if op[0] == "attn_fwd_fused" and op[2][2] is not None]
    assert sum(1 for n in names1 if n == "univl_attention_fwd_fused") == L0_fused and "univl_attention_fwd" not in names1
    L_t, L_v = cfg.text_num_hidden_layers, cfg.visual_num_hidden_layers
    assert len(riders) == 4 * ((L_t - 1) + (L_v - 1))
    # K8 / K10: the attention-output and FFN2 products of every layer carry their LayerNorm (round 4); no separate launch is left for those
    folds = [op for op in ride.fwd.ops if op[0] == "gemm_ln"]
    assert len(folds) == 2 * (L_t + L_v) and len([op for op in base.fwd.ops if op[0] == "gemm_ln"]) == len(folds)
    assert sum(1 for n in names1 if n == "univl_layernorm_fwd") == 2          # NormalizeVideo + the video embedding LayerNorm
    ab(ln_fold=0)
    unfolded = build_step(m, "joint", 2, 16, 16, True)
    ab(ln_fold=None)
    assert sum(1 for op in unfolded.fwd.ops if op[3] == "univl_layernorm_fwd") == 2 + len(folds) and not [op for op in unfolded.fwd.ops if op[0] == "gemm_ln"]
    carried = {("layer", "bert", l) for l in range(1, L_t)} | {("layer", "visual", l) for l in range(1, L_v)}
    assert ride.fwd.rider_keys == carried
    for key in carried:
        assert sorted(r[2] for r in riders if r[1] == key) == [0, 1, 2, 3]
    assert not base.fwd.rider_keys and len(ride.fwd.launches("univl_gemm")) == len(base.fwd.launches("univl_gemm"))
    assert [op[3] for op in ride.backward_plan(True).ops] == [op[3] for op in base.backward_plan(True).ops]
    # ... and the backward twin: every LayerNorm backward of a stack but its topmost one rides in the pair launch of the dgrad that feeds it
    bw = [op[3] for op in ride.backward_plan(True).ops]
    assert bw.count("univl_gemm_pair_ln") == (2 * L_t - 1) + (2 * L_v - 1)
    ab(ln_fold_bwd=0)
    bw0 = [op[3] for op in build_step(m, "joint", 2, 16, 16, True).backward_plan(True).ops]
    ab(ln_fold_bwd=None)
    assert bw0.count("univl_gemm_pair_ln") == 0 and bw0.count("univl_layernorm_bwd") == bw.count("univl_layernorm_bwd") + bw.count("univl_gemm_pair_ln")


def test_rect_rider_workgroup_roles_are_a_bijection():
    """gemm_adam_rect_kernel spreads its update workgroups through the grid (groups of 8 ids, one per XCD): every tile slot and every
    update index is taken exactly once, and a tile slot keeps its workgroup id modulo 8 (what the XCD-aware tile order is built on)."""
    import ctypes as C
    L = _lib.lib()
    for nd_pad, nb in ((2304, 288), (576, 288), (8, 8), (16, 296), (1152, 8), (24, 1000), (4608, 224)):
        tiles, upd = [], []
        for w0 in range(nd_pad + nb):
            out = (C.c_int32 * 3)(w0, 0, 0)
            assert L.univl_gemm_tile_map(4, nd_pad, nb, 1, 0, out) == 0
            (upd if out[1] else tiles).append(out[0])
            assert out[1] or out[0] % 8 == w0 % 8
        assert sorted(tiles) == list(range(nd_pad)) and sorted(upd) == list(range(nb)), (nd_pad, nb)
        if nd_pad >= 8 * nb:           # spread: no stretch of tile slots longer than two periods without an update group
            pos = [w0 for w0 in range(nd_pad + nb) if (lambda o: (L.univl_gemm_tile_map(4, nd_pad, nb, 1, 0, o), o[1])[1])((C.c_int32 * 3)(w0, 0, 0))]
            gaps = [b - a for a, b in zip(pos[:-1], pos[1:])]
            assert max(gaps) <= 2 * (nd_pad + nb) // (nb // 8) // 1, (nd_pad, nb, max(gaps))


def test_adam_rider_slots_from_1536_tokens_follow_the_library(ab):
    """From 1536 tokens on a layer's optimizer chunks are spread over the forward products univl_gemm_rider_fits says carry them
    (round 5: the 64 x 128 tile has a rider kernel, the 128 x 128 and 256 x 256 tiles have none); every layer's range is covered once."""
    import ctypes as C
    from univl_amd.steps import build_step
    m, cfg = _model("bf16", max_words=48, max_frames=48)
    m._flat, m._seed_dev = FlatParams(list(m.named_parameters()), "cpu", torch.bfloat16), torch.zeros(1, dtype=torch.int64)
    m.train()
    m._flat.adam_ride = True
    for B in (32, 128):
        st = build_step(m, "joint", B, 48, 48, True)
        riders = [op[2] for op in st.fwd.ops if op[0] == "rider"]
        assert riders and all(_lib.lib().univl_gemm_rider_fits(C.byref(d)) == 1 for d, _, _, _ in riders)
        plain = [st.fwd.descs[i][0] for i, op in enumerate(st.fwd.ops) if op[0] == "call" and op[3] == "univl_gemm"]
        by_key = {}
        for _, key, slot, n in riders:
            by_key.setdefault(key, []).append((slot, n))
        for op in st.fwd.ops:        # at 1536 tokens the fused attention forward still runs and carries its share (slot 0)
            if op[0] == "attn_fwd_fused" and op[2][2] is not None:
                by_key.setdefault(op[2][2], []).append((op[2][3], op[2][4]))
        assert set(by_key) == {("layer", "bert", l) for l in range(1, cfg.text_num_hidden_layers)}      # (one video layer: nothing to carry)
        for key, sl in by_key.items():
            assert sorted(s_ for s_, _ in sl) == list(range(sl[0][1])), (B, key, sl)
        # a product of a carrying layer that is NOT a rider launch is one the library does not carry
        layer0 = [d for d in plain if d.M == B * 48 and d.K in (768, 3072) and d.N in (768, 2304, 3072)]
        n_carry = len(by_key[("layer", "bert", 1)])
        assert n_carry + sum(1 for d in layer0 if _lib.lib().univl_gemm_rider_fits(C.byref(d)) == 0) >= 4 or n_carry == 4, (B, n_carry)


def test_stage_inputs_host_logic_on_cpu():
    """steps.stage_inputs: anything that is not a same-device, same-dtype, contiguous CUDA source takes the plain staging copy
    (dtype conversion, reshape, strided views, numpy arrays) -- the copy kernel is only ever handed pairs it can move bytewise."""
    import numpy as np
    from univl_amd.steps import stage_inputs
    d_ids, d_vid, d_mask = torch.zeros(2, 4, dtype=torch.int64), torch.zeros(8, 3, dtype=torch.float64), torch.zeros(2, 4, dtype=torch.int64)
    ids = torch.arange(8, dtype=torch.int32).reshape(2, 1, 4)                  # other dtype, extra pair dimension
    vid = torch.arange(48, dtype=torch.float64).reshape(2, 4, 6)[:, :, ::2]    # strided view, 24 elements
    mask = np.ones((2, 4), dtype=np.int64)                                     # numpy array
    stage_inputs([(d_ids, ids), (d_vid, vid), (d_mask, mask)])
    assert torch.equal(d_ids, ids.reshape(2, 4).to(torch.int64))
    assert torch.equal(d_vid, vid.reshape(8, 3))
    assert torch.equal(d_mask, torch.ones(2, 4, dtype=torch.int64))


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 2` without a launcher starts the ranks itself -- and fails loudly (exit code 2, the stated message)
    when fewer GPUs are visible than asked for; it never runs N ranks on fewer devices."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-400:])
    assert "--gpus 2 requested but only" in r.stderr


def test_word_table_chunks_start_on_row_boundaries_and_plan_ops_are_capturable():
    """Host logic of round 3: (1) the optimizer's chunk table cuts the word-embedding table into whole rows (UnivlAdam.row_flags
    tests rows, so a chunk must not straddle one), covers it exactly once, and leaves every other tensor on 8192-element chunks;
    (2) a plan whose exchange points are `with_streams` callables (the captured RCCL form) has NO eager segment -- the whole
    backward is one capturable run -- while the process-group form cuts it."""
    from univl_amd.optimization import _Tables, CHUNK
    from univl_amd.engine import Plan
    m, cfg = _model("fp32")
    fl = FlatParams(list(m.named_parameters()), "cpu", torch.float32)
    assert fl.word_ever is not None and fl.word_ever.numel() == cfg.vocab_size and int(fl.word_ever.sum()) == 0
    names = m.used_parameter_names()
    tb = _Tables(fl, {n: (1e-4, 0.01, 1.0, 1) for n in names})
    seg = fl.seg_of[fl.WORD]
    off, numel, shp = fl.index[fl.WORD]
    row = shp[1]
    cs, co, cl = tb.chunk_seg.tolist(), tb.chunk_off.tolist(), tb.chunk_len.tolist()
    mine = [(o, l) for s_, o, l in zip(cs, co, cl) if s_ == seg]
    assert mine[0][0] == off and sum(l for _, l in mine) == numel
    assert all((o - off) % row == 0 for o, _ in mine) and all(l % row == 0 for _, l in mine)
    assert all(a[0] + a[1] == b[0] for a, b in zip(mine, mine[1:]))                     # contiguous, in order, no overlap
    assert all(l <= CHUNK for l in cl) and max(l for s_, l in zip(cs, cl) if s_ != seg) == CHUNK
    fl.mark_all_word_rows()
    assert int(fl.word_ever.sum()) == cfg.vocab_size and fl.word_ever_all
    p1, p2 = Plan(), Plan()
    for p, captured in ((p1, True), (p2, False)):
        p.add_callable(lambda: None)
        if captured:
            p.add_callable(lambda streams: None, with_streams=True)
        else:
            p.add_callable(lambda: None, eager=True)
        p.add_callable(lambda: None)
    assert [k for k, _ in p1.segments()] == ["graph"]
    assert [k for k, _ in p2.segments()] == ["graph", "eager", "graph"]


def test_deep_weight_gradients_take_the_big_tile_without_fused_bias_gradients(ab):
    """engine.EncoderStack where every dgrad product of a layer is on the 128 tile (>= 256 tiles of 128 x 128 for an H-wide output:
    5462 tokens at H = 768 -- the weight gradients cannot ride with their dgrad there; wgrad_big_min = tokens overrides), bf16:
    the layer's grouped weight gradients ask for the 128 tile on two stages and 4 waves; the two bias gradients (FFN1, QKV) stay in
    the descriptors and the C side takes them with column-sum workgroups of the same launch (round 4; round 3: two extra launches
    per layer).  Below, the plan is the former one (pairs).  Built on the CPU: the same weight-gradient outputs either way."""
    from univl_amd.steps import build_step
    m, cfg = _model("bf16")
    fl = FlatParams(list(m.named_parameters()), "cpu", torch.bfloat16)
    m._flat, m._seed_dev = fl, torch.zeros(1, dtype=torch.int64)
    m.train()
    B, W = 352, 16                                       # 5632 tokens in the text and in the video stack
    layers = cfg.text_num_hidden_layers + cfg.visual_num_hidden_layers

    def groups(plan):
        out, lambdas, pairs = [], 0, 0
        for i, op in enumerate(plan.ops):
            if op[0] == "group":
                out.append(plan.descs[i])
            elif op[0] == "pair":
                pairs += 1
            elif op[0] == "py" and op[3] == "<lambda>":
                lambdas += 1
        return out, lambdas, pairs

    ab(wgrad_big_min=1000000000)
    old, old_py, old_pairs = groups(build_step(m, "joint", B, W, W, True).backward_plan(True))
    ab(wgrad_big_min=None, g256=0)
    new, new_py, new_pairs = groups(build_step(m, "joint", B, W, W, True).backward_plan(True))
    # round 5 default (token count a multiple of 256): no tile request -- the C side runs the group on the 256 x 256 8-phase body
    # (csrc/gemm256.h: auto256), bias gradients as column-sum roles as on the 128 tile
    ab(g256=None)
    g256, _, g256_pairs = groups(build_step(m, "joint", B, W, W, True).backward_plan(True))
    assert len(g256) == layers and g256_pairs == 0
    assert all((d.tile, d.stages, d.waves) == (0, 0, 0) and d.M % 256 == 0 and d.N % 256 == 0 and d.K % 256 == 0 for g in g256 for d in g)
    assert len(old) == len(new) == layers and old_pairs == new_pairs == 0      # no pair launches at this size either way
    assert new_py == old_py                              # no separate column-sum launches
    for go, gn in zip(old, new):
        assert len(go) == len(gn) == 4
        assert sorted(d.C32 for d in go) == sorted(d.C32 for d in gn)           # the same four matrices
        assert sum(1 for d in go if d.dbias) == 2 and sorted(d.dbias or 0 for d in go) == sorted(d.dbias or 0 for d in gn)
        for d in go:
            assert (d.tile, d.stages, d.waves) == (0, 0, 0)
        for d in gn:
            assert (d.tile, d.stages, d.waves) == (128, 2, 4) and d.trans_a and d.trans_b and d.K == B * W
            assert d.sumsq_rows % 128 == 0                # the fused gradient norms stay legal on the 128 tile
    # 2048 tokens (a multiple of 256 from 1536 on): round 5 -- no weight gradient rides any more, all four per layer in the grouped launch
    mid, mid_py, mid_pairs = groups(build_step(m, "joint", 128, 16, 16, True).backward_plan(True))
    assert mid_pairs == 0 and len(mid) == layers and all(len(g) == 4 and all((d.tile, d.stages, d.waves) == (0, 0, 0) for d in g) for g in mid)
    # ... the former plan (g256=0): the weight gradients of the H-wide dgrad products ride with them
    ab(g256=0)
    mid, mid_py, mid_pairs = groups(build_step(m, "joint", 128, 16, 16, True).backward_plan(True))
    assert mid_pairs > 0 and all((d.tile, d.stages, d.waves) == (0, 0, 0) for g in mid for d in g)
    # ... and 1200 tokens (not a multiple of 256): pair launches as before
    ab(g256=None)
    small, _, small_pairs = groups(build_step(m, "joint", 75, 16, 16, True).backward_plan(True))
    assert small_pairs > 0


def test_layernorm_fold_entry_points_validate_on_the_host():
    """univl_gemm_ln / univl_gemm_pair_ln with dry_run: which (product, LayerNorm) pairs the launches carry is decided on the host from the
    two descriptors alone (no device work) -- the plans ask exactly this question at build time.  Refused: a LayerNorm that does not read
    the product's output, other widths, a product with a bf16 output, too few tiles per row block, more than 16 row blocks.  (Deterministic mode refuses everything: covered on the GPU, test_gemm_ln_fold_* -- switching the mode
    allocates device scratch.)"""
    import ctypes as C
    import univl_amd
    from univl_amd import ops
    from univl_amd.engine import _gemm_desc
    L = _lib.lib()
    bf, dt = torch.bfloat16, _lib.DT_BF16
    if univl_amd.deterministic():
        pytest.skip("UNIVL_DETERMINISTIC is set")
    if True:
        T, H, I = 192, 768, 3072
        a, w = torch.zeros(T, I, dtype=bf), torch.zeros(H, I, dtype=bf)
        x, x2 = torch.zeros(T, H), torch.zeros(T, H)
        gm, bt, st = torch.ones(H), torch.zeros(H), torch.zeros(T, 2)
        o16 = torch.zeros(T, H, dtype=bf)
        ctr = torch.zeros(2 * 3, dtype=torch.int32)
        cp = C.c_void_p(ctr.data_ptr())

        def fwd(g, ln):
            return L.univl_gemm_ln(C.byref(g), C.byref(ln), cp, None, 0, 0, 0, 1, None)

        g = _gemm_desc(dt, a, I, w, I, T, H, I, out32=x, ldc=H, ksplit=8)
        ln = ops.layernorm_desc(dt, T, H, x=x, residual=x2, gamma=gm, beta=bt, y=x, stats=st, out16=o16)
        assert fwd(g, ln) == 0
        assert fwd(g, ops.layernorm_desc(dt, T, H, x=x2, gamma=gm, beta=bt, y=x2, stats=st, out16=o16)) == _lib.EUNSUPPORTED      # reads something else
        assert fwd(_gemm_desc(dt, a, I, w, I, T, H, I, out16=o16, ldc=H), ln) == _lib.EUNSUPPORTED                              # bf16 product output
        w2 = torch.zeros(2 * H, I, dtype=bf)
        assert fwd(_gemm_desc(dt, a, I, w2, I, T, 2 * H, I, out32=torch.zeros(T, 2 * H), ldc=2 * H), ln) == _lib.EUNSUPPORTED    # N != 768
        assert L.univl_gemm_ln(C.byref(g), C.byref(ln), None, None, 0, 0, 0, 1, None) != 0                                      # no counters
        # backward twin
        dY, W1, X = torch.zeros(T, I, dtype=bf), torch.zeros(I, H, dtype=bf), torch.zeros(T, H, dtype=bf)
        da, dW = torch.zeros(T, H), torch.zeros(I, H)

        def bwd(dg, wg, lb):
            return L.univl_gemm_pair_ln(C.byref(dg), C.byref(wg), C.byref(lb), cp, 1, None)

        dg = _gemm_desc(dt, dY, I, W1, H, T, H, I, trans_b=1, out32=da, ldc=H, ksplit=8)
        wg = _gemm_desc(dt, dY, I, X, H, I, H, T, trans_a=1, trans_b=1, out32=dW, ldc=H)
        lb = ops.layernorm_desc(dt, T, H, gamma=gm, y=x, stats=st, dout=da, dx32=x2, dxd16=o16, dgamma=torch.zeros(H), dbeta=torch.zeros(H))
        assert bwd(dg, wg, lb) == 0
        lb_other = ops.layernorm_desc(dt, T, H, gamma=gm, y=x, stats=st, dout=x2, dx32=x2, dxd16=o16)
        assert bwd(dg, wg, lb_other) == _lib.EUNSUPPORTED                                                                       # other upstream gradient
        T2 = 384                                                                                                                 # rectangular dgrad body
        dY2, X2, da2 = torch.zeros(T2, I, dtype=bf), torch.zeros(T2, H, dtype=bf), torch.zeros(T2, H)
        dg2 = _gemm_desc(dt, dY2, I, W1, H, T2, H, I, trans_b=1, out32=da2, ldc=H, ksplit=2)
        wg2 = _gemm_desc(dt, dY2, I, X2, H, I, H, T2, trans_a=1, trans_b=1, out32=dW, ldc=H)
        lb2 = ops.layernorm_desc(dt, T2, H, gamma=gm, y=torch.zeros(T2, H), stats=torch.zeros(T2, 2), dout=da2, dx32=torch.zeros(T2, H),
                                 dxd16=torch.zeros(T2, H, dtype=bf))
        # round 5: carried on the rectangular form too (beside the 128 x 64 weight-gradient body), up to 1024 rows
        assert L.univl_gemm_pair(C.byref(dg2), C.byref(wg2), 1, None) == 0 and bwd(dg2, wg2, lb2) == 0
        T3 = 1088
        dY3, X3, da3 = torch.zeros(T3, I, dtype=bf), torch.zeros(T3, H, dtype=bf), torch.zeros(T3, H)
        dg3 = _gemm_desc(dt, dY3, I, W1, H, T3, H, I, trans_b=1, out32=da3, ldc=H, ksplit=2)
        wg3 = _gemm_desc(dt, dY3, I, X3, H, I, H, T3, trans_a=1, trans_b=1, out32=dW, ldc=H)
        lb3 = ops.layernorm_desc(dt, T3, H, gamma=gm, y=torch.zeros(T3, H), stats=torch.zeros(T3, 2), dout=da3, dx32=torch.zeros(T3, H),
                                 dxd16=torch.zeros(T3, H, dtype=bf))
        assert L.univl_gemm_pair(C.byref(dg3), C.byref(wg3), 1, None) == 0 and bwd(dg3, wg3, lb3) == _lib.EUNSUPPORTED      # 17 row blocks


@pytest.mark.parametrize("trans", [0, 1])
def test_gemm256_lds_image_and_fragment_reads_agree_and_are_conflict_free(trans):
    """The 256 x 256 product body (csrc/gemm256.h) fills 16-KB half-tile images by LDS-DMA (lane-linear destination, XOR-swizzled
    SOURCE piece) and reads MFMA fragments back with the same XOR.  This rebuilds an image from the library's DMA map, emulates
    ds_read_b128 / ds_read_b64_tr_b16 on the library's per-lane read offsets, and checks (a) lane l of a 32x32x16 fragment gets row /
    column (l & 31), contraction indices 16 ks + 8 (l >> 5) + 0..7 in order -- for both operand layouts, so that A and B always agree
    on the k slots; (b) the image is a bijection; (c) no bank conflicts within the LDS's per-instruction lane groups
    (MI355X_MICROARCH.md, LDS table)."""
    import ctypes as C
    L = _lib.lib()
    out = (C.c_int32 * 8)()
    # image[byte // 2] = (row, k) of the bf16 element stored there
    image = {}
    for piece in range(1024):
        assert L.univl_gemm256_layout(0, trans, piece, 0, out) == 0
        row, k = out[0], out[1]
        for e in range(8):
            image[piece * 8 + e] = (row + e, k) if trans else (row, k + e)
    assert sorted(image.values()) == [(r, k) for r in range(128) for k in range(64)]          # every element exactly once
    b128_groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    b128_groups += [[l + 32 for l in g] for g in b128_groups]
    for r32 in (0, 32, 64, 96):
        offs = []
        for lane in range(64):
            assert L.univl_gemm256_layout(1, trans, lane, r32, out) == 0
            offs.append(list(out))
        for ks in range(4):
            if not trans:
                for lane in range(64):
                    o = offs[lane][ks]
                    assert o % 16 == 0 and 0 <= o < 16384
                    got = [image[o // 2 + e] for e in range(8)]
                    assert got == [(r32 + (lane & 31), 16 * ks + 8 * (lane >> 5) + e) for e in range(8)]
                for g in b128_groups:                       # 16 lanes -> the 16 distinct 16-byte slots of the 256-byte bank row
                    assert len({(offs[lane][ks] >> 4) & 15 for lane in g}) == 16
            else:
                for rd in range(2):
                    addr = [offs[lane][2 * ks + rd] for lane in range(64)]
                    assert all(a % 8 == 0 and 0 <= a < 16384 for a in addr)
                    for lane in range(64):
                        base, i = lane & ~15, lane & 15
                        # ds_read_b64_tr_b16: lane i of a 16-lane group receives column i of the 4 x 16 block whose row j is supplied,
                        # 4 columns each, by lanes 4 j .. 4 j + 3 of the group
                        got = [image[addr[base + 4 * j + (i >> 2)] // 2 + (i & 3)] for j in range(4)]
                        assert got == [(r32 + (lane & 31), 16 * ks + 8 * (lane >> 5) + 4 * rd + j) for j in range(4)]
                    for half in (range(0, 32), range(32, 64)):      # 32 lanes x 8 bytes = every one of the 64 banks once
                        banks = [((addr[lane] >> 2) + d) & 63 for lane in half for d in (0, 1)]
                        assert len(set(banks)) == 64
