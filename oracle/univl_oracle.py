"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A CPU (torch, fp32 or fp64, autograd for the backward) restatement of the algorithm on the UniVL hot path
(SURVEY.md section 8a).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; nothing under `univl_amd/` does.  The product path is the HIP library and fails loudly
without it.

Every function cites the reference file:line (relative to microsoft/UniVL) whose arithmetic it restates.
The arithmetic itself lives in PyTorch ATen in the reference (torch==1.7.0 pinned at requirements.txt:1); the
reference has no tests and no golden vectors of its own, so this restatement is pinned
instead against outputs of the reference's own classes executed in the build container:
`oracle/make_golden.py` imports `/root/reference`, loads the SAME procedural parameters and inputs defined in
this file into `modules.modeling.UniVL`, and writes the fixtures under `tests/golden/`;
`tests/test_oracle_golden.py` then checks this restatement against those fixtures (and, when the reference
is mounted, against the live reference).

Functional style: parameters are a flat dict {reference state_dict key -> tensor}.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field, asdict
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

# ------------------------------------------------------------------------------------------------ config


@dataclass
class OracleConfig:
    """Union of the module JSON configs and the task_config attributes the reference reads
    (modules/*-base/*.json; bert_config values of module_bert.py:61-72)."""
    vocab_size: int = 30522
    hidden_size: int = 768
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512      # bert + visual position tables
    cross_max_position_embeddings: int = 1024
    max_target_embeddings: int = 512        # decoder_config.json
    type_vocab_size: int = 2
    video_dim: int = 1024
    text_num_hidden_layers: int = 12
    visual_num_hidden_layers: int = 6
    cross_num_hidden_layers: int = 2
    decoder_num_hidden_layers: int = 3
    # task_config
    max_words: int = 48
    max_frames: int = 48
    batch_size: int = 4
    n_gpu: int = 1
    n_pair: int = 1
    margin: float = 0.1
    negative_weighting: int = 1
    hard_negative_rate: float = 0.5
    use_mil: bool = False
    do_pretrain: bool = False
    task_type: str = "retrieval"
    stage_two: bool = False
    train_sim_after_cross: bool = False
    dropout_prob: float = 0.0              # 0 for parity; 0.1 in the reference's JSON configs

    # derived exactly as modeling.py:120-131 does
    @property
    def stage_one(self) -> bool:
        return not bool(self.stage_two)

    @property
    def sim_after_cross(self) -> bool:
        return self.stage_one and bool(self.train_sim_after_cross)

    @property
    def has_cross(self) -> bool:            # modeling.py:147
        return (not self.stage_one) or self.sim_after_cross

    @property
    def has_decoder(self) -> bool:          # modeling.py:154
        return self.has_cross and not self.sim_after_cross

    @property
    def has_pretrain_heads(self) -> bool:   # modeling.py:161
        return self.has_cross and bool(self.do_pretrain)

    def to_dict(self):
        return asdict(self)


# ------------------------------------------------------------------------------ parameter inventory


def _encoder_layer_shapes(prefix: str, H: int, I: int):
    s = {}
    for nm in ("query", "key", "value"):
        s[f"{prefix}.attention.self.{nm}.weight"] = (H, H)
        s[f"{prefix}.attention.self.{nm}.bias"] = (H,)
    s[f"{prefix}.attention.output.dense.weight"] = (H, H)
    s[f"{prefix}.attention.output.dense.bias"] = (H,)
    s[f"{prefix}.attention.output.LayerNorm.weight"] = (H,)
    s[f"{prefix}.attention.output.LayerNorm.bias"] = (H,)
    s[f"{prefix}.intermediate.dense.weight"] = (I, H)
    s[f"{prefix}.intermediate.dense.bias"] = (I,)
    s[f"{prefix}.output.dense.weight"] = (H, I)
    s[f"{prefix}.output.dense.bias"] = (H,)
    s[f"{prefix}.output.LayerNorm.weight"] = (H,)
    s[f"{prefix}.output.LayerNorm.bias"] = (H,)
    return s


def param_shapes(cfg: OracleConfig) -> Dict[str, tuple]:
    """Names and shapes of `UniVL.named_parameters()` in registration order (tied duplicates removed, as
    nn.Module.named_parameters does).  Follows modeling.py:110-186 and the four module constructors."""
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    s: Dict[str, tuple] = {}
    # BertModel (module_bert.py:118-130, 267-296, 364-414)
    s["bert.embeddings.word_embeddings.weight"] = (V, H)
    s["bert.embeddings.position_embeddings.weight"] = (cfg.max_position_embeddings, H)
    s["bert.embeddings.token_type_embeddings.weight"] = (cfg.type_vocab_size, H)
    s["bert.embeddings.LayerNorm.weight"] = (H,)
    s["bert.embeddings.LayerNorm.bias"] = (H,)
    for i in range(cfg.text_num_hidden_layers):
        s.update(_encoder_layer_shapes(f"bert.encoder.layer.{i}", H, I))
    s["bert.pooler.dense.weight"] = (H, H)
    s["bert.pooler.dense.bias"] = (H,)
    # VisualModel (module_visual.py:104-131, 251-280, 346-395)
    s["visual.embeddings.word_embeddings.weight"] = (H, cfg.video_dim)
    s["visual.embeddings.word_embeddings.bias"] = (H,)
    s["visual.embeddings.position_embeddings.weight"] = (cfg.max_position_embeddings, H)
    s["visual.embeddings.LayerNorm.weight"] = (H,)
    s["visual.embeddings.LayerNorm.bias"] = (H,)
    for i in range(cfg.visual_num_hidden_layers):
        s.update(_encoder_layer_shapes(f"visual.encoder.layer.{i}", H, I))
    s["visual.pooler.dense.weight"] = (H, H)
    s["visual.pooler.dense.bias"] = (H,)
    if cfg.has_cross:
        # CrossModel (module_cross.py:109-138, 258-287, 356-362)
        s["cross.embeddings.position_embeddings.weight"] = (cfg.cross_max_position_embeddings, H)
        s["cross.embeddings.token_type_embeddings.weight"] = (cfg.type_vocab_size, H)
        s["cross.embeddings.LayerNorm.weight"] = (H,)
        s["cross.embeddings.LayerNorm.bias"] = (H,)
        for i in range(cfg.cross_num_hidden_layers):
            s.update(_encoder_layer_shapes(f"cross.encoder.layer.{i}", H, I))
        s["cross.pooler.dense.weight"] = (H, H)
        s["cross.pooler.dense.bias"] = (H,)
        if cfg.has_decoder:
            # DecoderModel (module_decoder.py:279-349); word/position embeddings and the classifier's
            # decoder.weight are TIED to BERT's tables (modeling.py:137-138,159; module_decoder.py:301-302,177)
            s["decoder.embeddings.LayerNorm.weight"] = (H,)
            s["decoder.embeddings.LayerNorm.bias"] = (H,)
            for i in range(cfg.decoder_num_hidden_layers):
                p = f"decoder.decoder.layer.{i}"
                for blk in ("slf_attn", "enc_attn"):
                    for nm in ("query", "key", "value"):
                        s[f"{p}.{blk}.att.{nm}.weight"] = (H, H)
                        s[f"{p}.{blk}.att.{nm}.bias"] = (H,)
                    s[f"{p}.{blk}.output.dense.weight"] = (H, H)
                    s[f"{p}.{blk}.output.dense.bias"] = (H,)
                    s[f"{p}.{blk}.output.LayerNorm.weight"] = (H,)
                    s[f"{p}.{blk}.output.LayerNorm.bias"] = (H,)
                s[f"{p}.intermediate.dense.weight"] = (I, H)
                s[f"{p}.intermediate.dense.bias"] = (I,)
                s[f"{p}.output.dense.weight"] = (H, I)
                s[f"{p}.output.dense.bias"] = (H,)
                s[f"{p}.output.LayerNorm.weight"] = (H,)
                s[f"{p}.output.LayerNorm.bias"] = (H,)
            s["decoder.classifier.cls.predictions.bias"] = (V,)
            s["decoder.classifier.cls.predictions.transform.dense.weight"] = (H, H)
            s["decoder.classifier.cls.predictions.transform.dense.bias"] = (H,)
            s["decoder.classifier.cls.predictions.transform.LayerNorm.weight"] = (H,)
            s["decoder.classifier.cls.predictions.transform.LayerNorm.bias"] = (H,)
        if cfg.has_pretrain_heads:
            # BertOnlyMLMHead tied to BERT word table; VisualOnlyMLMHead tied to visual input projection
            # (modeling.py:161-164; module_bert.py:314-330; module_visual.py:298-311)
            s["cls.predictions.bias"] = (V,)
            s["cls.predictions.transform.dense.weight"] = (H, H)
            s["cls.predictions.transform.dense.bias"] = (H,)
            s["cls.predictions.transform.LayerNorm.weight"] = (H,)
            s["cls.predictions.transform.LayerNorm.bias"] = (H,)
            s["cls_visual.predictions.bias"] = (cfg.video_dim,)
            s["cls_visual.predictions.transform.dense.weight"] = (H, H)
            s["cls_visual.predictions.transform.dense.bias"] = (H,)
            s["cls_visual.predictions.transform.LayerNorm.weight"] = (H,)
            s["cls_visual.predictions.transform.LayerNorm.bias"] = (H,)
        s["similarity_dense.weight"] = (1, H)           # modeling.py:167
        s["similarity_dense.bias"] = (1,)
    s["normalize_video.visual_norm2d.weight"] = (cfg.video_dim,)   # modeling.py:83-86
    s["normalize_video.visual_norm2d.bias"] = (cfg.video_dim,)
    return s


#: state_dict aliases of tied parameters: alias key -> owning key (modeling.py:137-138,159,163-164)
def tied_aliases(cfg: OracleConfig) -> Dict[str, str]:
    t = {}
    if cfg.has_decoder:
        t["decoder.embeddings.word_embeddings.weight"] = "bert.embeddings.word_embeddings.weight"
        t["decoder.embeddings.position_embeddings.weight"] = "bert.embeddings.position_embeddings.weight"
        t["decoder.classifier.cls.predictions.decoder.weight"] = "bert.embeddings.word_embeddings.weight"
    if cfg.has_pretrain_heads:
        t["cls.predictions.decoder.weight"] = "bert.embeddings.word_embeddings.weight"
        t["cls_visual.predictions.weight"] = "visual.embeddings.word_embeddings.weight"
    return t


def procedural_params(cfg: OracleConfig, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Deterministic parameters shared by reference, oracle and HIP path WITHOUT shipping a 615 MB state_dict.

    Each tensor is drawn from its own torch CPU generator seeded by crc32(name) ^ seed, so the value of one
    parameter does not depend on which other parameters exist.  Magnitudes follow the reference init
    (N(0, 0.02) matrices/embeddings, until_module.py:70-85) but LayerNorm gains and all biases are perturbed
    away from 1/0 so that every bias/gain path is exercised by the parity tests.
    """
    out = {}
    for name, shape in param_shapes(cfg).items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
        if name.endswith("LayerNorm.weight") or name.endswith("visual_norm2d.weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            t = 0.02 * torch.randn(shape, generator=g)
        out[name] = t.to(dtype)
    return out


def synthetic_batch(cfg: OracleConfig, rows: int, seed: int = 1234, all_ones_mask: bool = False):
    """Synthetic inputs of SURVEY.md section 8d.  `rows` = batch (each item one pair, pair dim = n_pair)."""
    g = torch.Generator().manual_seed(seed)
    W, Fm, D = cfg.max_words, cfg.max_frames, cfg.video_dim
    npair = cfg.n_pair
    B = rows
    ids = torch.randint(1000, cfg.vocab_size, (B, npair, W), generator=g, dtype=torch.int64)
    ids[..., 0] = 101
    lt = torch.randint(4, W + 1, (B, npair), generator=g)
    lv = torch.randint(1, Fm + 1, (B, npair), generator=g)
    if all_ones_mask:
        lt[:] = W
        lv[:] = Fm
    amask = (torch.arange(W)[None, None, :] < lt[..., None]).to(torch.int64)
    vmask = (torch.arange(Fm)[None, None, :] < lv[..., None]).to(torch.int64)
    ids = ids * amask
    video = torch.randn((B, npair, Fm, D), generator=g, dtype=torch.float64)
    video = video * vmask[..., None].to(torch.float64)
    batch = dict(input_ids=ids, token_type_ids=torch.zeros_like(ids), attention_mask=amask,
                 video=video, video_mask=vmask)
    # pretrain extras (15 % token / frame masking, labels -1 elsewhere) -- dataloader_howto100m semantics
    pm = (torch.rand((B, npair, W), generator=g) < 0.15) & (amask > 0)
    pm[..., 0] = False
    masked_text = torch.where(pm, torch.full_like(ids, 103), ids)
    labels = torch.where(pm, ids, torch.full_like(ids, -1))
    fm_ = (torch.rand((B, npair, Fm), generator=g) < 0.15) & (vmask > 0)
    masked_video = video * (~fm_)[..., None].to(torch.float64)
    vlabels = torch.where(fm_, torch.arange(Fm)[None, None, :].expand(B, npair, Fm), torch.full((B, npair, Fm), -1))
    batch.update(pairs_masked_text=masked_text, pairs_token_labels=labels,
                 masked_video=masked_video, video_labels_index=vlabels.to(torch.int64))
    # caption extras
    lc = torch.randint(2, W + 1, (B, npair), generator=g)
    cap = torch.randint(1000, cfg.vocab_size, (B, npair, W), generator=g, dtype=torch.int64)
    cmask = (torch.arange(W)[None, None, :] < lc[..., None]).to(torch.int64)
    cap_in = cap * cmask
    cap_out = torch.roll(cap, -1, dims=-1) * cmask        # padding label 0 is NOT ignored (modeling.py:168)
    batch.update(input_caption_ids=cap_in, decoder_mask=cmask, output_caption_ids=cap_out)
    return batch


# ---------------------------------------------------------------------------------------- primitives


def gelu(x):
    """until_module.py:28-33 (erf form)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, weight, bias, eps=1e-12):
    """until_module.py:49-53 -- TF style: biased variance, eps inside the sqrt."""
    u = x.mean(-1, keepdim=True)
    s = (x - u).pow(2).mean(-1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return weight * x + bias


def linear(x, P, prefix):
    return F.linear(x, P[prefix + ".weight"], P[prefix + ".bias"])


def _drop(x, p, training):
    return F.dropout(x, p, training) if (p > 0 and training) else x


def extended_mask(mask, dtype):
    """module_bert.py:429-437: (B,S) {0,1} -> (B,1,1,S) additive {0,-10000}."""
    m = mask.unsqueeze(1).unsqueeze(2).to(dtype)
    return (1.0 - m) * -10000.0


def attention_core(q, k, v, add_mask, nh, p_drop=0.0, training=False):
    """module_bert.py:166-197 (copies module_visual.py:165-181, module_cross.py:172-188,
    module_decoder.py:230-247): scores/sqrt(d) THEN + mask, softmax, dropout on probs, .V, merge heads."""
    B, Sq, H = q.shape
    Sk = k.shape[1]
    d = H // nh
    ql = q.view(B, Sq, nh, d).permute(0, 2, 1, 3)
    kl = k.view(B, Sk, nh, d).permute(0, 2, 1, 3)
    vl = v.view(B, Sk, nh, d).permute(0, 2, 1, 3)
    scores = torch.matmul(ql, kl.transpose(-1, -2))
    scores = scores / math.sqrt(d)
    scores = scores + add_mask
    probs = torch.softmax(scores, dim=-1)
    probs = _drop(probs, p_drop, training)
    ctx = torch.matmul(probs, vl)
    return ctx.permute(0, 2, 1, 3).contiguous().view(B, Sq, H)


def self_output(x, residual, P, prefix, p_drop, training):
    """module_bert.py:207-211: LN(dropout(dense(x)) + residual)."""
    h = linear(x, P, prefix + ".dense")
    h = _drop(h, p_drop, training)
    return layer_norm(h + residual, P[prefix + ".LayerNorm.weight"], P[prefix + ".LayerNorm.bias"])


def encoder_layer(x, add_mask, P, prefix, cfg, training):
    """module_bert.py:253-264 (BertLayer) == VisualLayer == CrossLayer."""
    a = prefix + ".attention.self"
    q, k, v = linear(x, P, a + ".query"), linear(x, P, a + ".key"), linear(x, P, a + ".value")
    ctx = attention_core(q, k, v, add_mask, cfg.num_attention_heads, cfg.dropout_prob, training)
    att = self_output(ctx, x, P, prefix + ".attention.output", cfg.dropout_prob, training)
    inter = gelu(linear(att, P, prefix + ".intermediate.dense"))            # module_bert.py:233-236
    return self_output(inter, att, P, prefix + ".output", cfg.dropout_prob, training)   # :246-250


def encoder(x, add_mask, P, prefix, n_layers, cfg, training):
    for i in range(n_layers):
        x = encoder_layer(x, add_mask, P, f"{prefix}.encoder.layer.{i}", cfg, training)
    return x


# ------------------------------------------------------------------------------------------- modules


def normalize_video(video, P):
    """modeling.py:88-92: float64 -> float32, view(-1,F,D), LayerNorm(video_dim)."""
    dt = P["normalize_video.visual_norm2d.weight"].dtype
    video = torch.as_tensor(video).to(torch.float32).to(dt)
    video = video.view(-1, video.shape[-2], video.shape[-1])
    return layer_norm(video, P["normalize_video.visual_norm2d.weight"], P["normalize_video.visual_norm2d.bias"])


def bert_model(input_ids, token_type_ids, attention_mask, P, cfg, training):
    """module_bert.py:417-447 + BertEmbeddings :132-146.  Pooler output is discarded by every caller
    (modeling.py:307) and is not computed here."""
    S = input_ids.size(1)
    pos = torch.arange(S, dtype=torch.long)
    e = (P["bert.embeddings.word_embeddings.weight"][input_ids]
         + P["bert.embeddings.position_embeddings.weight"][pos].unsqueeze(0)
         + P["bert.embeddings.token_type_embeddings.weight"][token_type_ids])
    e = layer_norm(e, P["bert.embeddings.LayerNorm.weight"], P["bert.embeddings.LayerNorm.bias"])
    e = _drop(e, cfg.dropout_prob, training)
    return encoder(e, extended_mask(attention_mask, e.dtype), P, "bert", cfg.text_num_hidden_layers, cfg, training)


def visual_model(video, video_mask, P, cfg, training):
    """module_visual.py:397-425 + VisualEmbeddings :118-131."""
    S = video.size(1)
    e = linear(video, P, "visual.embeddings.word_embeddings")
    e = e + P["visual.embeddings.position_embeddings.weight"][torch.arange(S)].unsqueeze(0)
    e = layer_norm(e, P["visual.embeddings.LayerNorm.weight"], P["visual.embeddings.LayerNorm.bias"])
    e = _drop(e, cfg.dropout_prob, training)
    return encoder(e, extended_mask(video_mask, e.dtype), P, "visual", cfg.visual_num_hidden_layers, cfg, training)


def cross_model(concat, concat_type, concat_mask, P, cfg, training):
    """module_cross.py:364-394 + CrossEmbeddings :123-138 + CrossPooler :281-287 -> (last layer, pooled)."""
    S = concat.size(1)
    e = (concat + P["cross.embeddings.position_embeddings.weight"][torch.arange(S)].unsqueeze(0)
         + P["cross.embeddings.token_type_embeddings.weight"][concat_type])
    e = layer_norm(e, P["cross.embeddings.LayerNorm.weight"], P["cross.embeddings.LayerNorm.bias"])
    e = _drop(e, cfg.dropout_prob, training)
    out = encoder(e, extended_mask(concat_mask, e.dtype), P, "cross", cfg.cross_num_hidden_layers, cfg, training)
    pooled = torch.tanh(linear(out[:, 0], P, "cross.pooler.dense"))
    return out, pooled


def get_cross_output(seq_out, vis_out, attention_mask, video_mask, P, cfg, training):
    """modeling.py:315-325."""
    concat = torch.cat((seq_out, vis_out), dim=1)
    cmask = torch.cat((attention_mask, video_mask), dim=1)
    ctype = torch.cat((torch.zeros_like(attention_mask), torch.ones_like(video_mask)), dim=1)
    out, pooled = cross_model(concat, ctype, cmask, P, cfg, training)
    return out, pooled, cmask


def lm_head(x, P, prefix, tied_weight, transposed=False):
    """module_bert.py:302-330 / module_decoder.py:156-183: LN(gelu(dense(x))) . E^T + bias;
    module_visual.py:291-311 uses x.matmul(W) (W is (768,1024)) -> transposed=True."""
    h = gelu(linear(x, P, prefix + ".transform.dense"))
    h = layer_norm(h, P[prefix + ".transform.LayerNorm.weight"], P[prefix + ".transform.LayerNorm.bias"])
    if transposed:
        return h.matmul(tied_weight) + P[prefix + ".bias"]
    return F.linear(h, tied_weight) + P[prefix + ".bias"]


def decoder_model(input_ids, encoder_outs, answer_mask, encoder_mask, P, cfg, training):
    """module_decoder.py:372-406 (+ embeddings :309-320, layer :287-292, attention :220-247, :274-277)."""
    S = input_ids.size(1)
    e = (P["bert.embeddings.word_embeddings.weight"][input_ids]
         + P["bert.embeddings.position_embeddings.weight"][torch.arange(S)].unsqueeze(0))
    e = layer_norm(e, P["decoder.embeddings.LayerNorm.weight"], P["decoder.embeddings.LayerNorm.bias"])
    e = _drop(e, cfg.dropout_prob, training)
    dt = e.dtype
    enc_add = extended_mask(encoder_mask, dt)
    ext_ans = answer_mask.unsqueeze(1).unsqueeze(2).to(dt)
    sub = torch.triu(torch.ones((S, S), dtype=dt), diagonal=1)
    slf = ((1.0 - ext_ans) + sub.unsqueeze(0).unsqueeze(1)).gt(0).to(dt) * -10000.0       # :389-396
    nh = cfg.num_attention_heads
    x = e
    for i in range(cfg.decoder_num_hidden_layers):
        p = f"decoder.decoder.layer.{i}"
        a = p + ".slf_attn.att"
        ctx = attention_core(linear(x, P, a + ".query"), linear(x, P, a + ".key"), linear(x, P, a + ".value"),
                             slf, nh, cfg.dropout_prob, training)
        s_out = self_output(ctx, x, P, p + ".slf_attn.output", cfg.dropout_prob, training)
        a = p + ".enc_attn.att"
        ctx = attention_core(linear(s_out, P, a + ".query"), linear(encoder_outs, P, a + ".key"),
                             linear(encoder_outs, P, a + ".value"), enc_add, nh, cfg.dropout_prob, training)
        d_out = self_output(ctx, s_out, P, p + ".enc_attn.output", cfg.dropout_prob, training)
        inter = gelu(linear(d_out, P, p + ".intermediate.dense"))
        x = self_output(inter, d_out, P, p + ".output", cfg.dropout_prob, training)
    return lm_head(x, P, "decoder.classifier.cls.predictions", P["bert.embeddings.word_embeddings.weight"])


# --------------------------------------------------------------------------------- similarity + losses


def mean_pooling_for_similarity(seq_out, vis_out, attention_mask, video_mask):
    """modeling.py:327-339: text mean excludes position 0; video zero-count -> 1."""
    am = attention_mask.to(seq_out.dtype).unsqueeze(-1).clone()
    am[:, 0, :] = 0.
    text_out = torch.sum(seq_out * am, dim=1) / torch.sum(am, dim=1)
    vm = video_mask.to(vis_out.dtype).unsqueeze(-1)
    vsum = torch.sum(vm, dim=1)
    vsum = torch.where(vsum == 0., torch.ones_like(vsum), vsum)
    video_out = torch.sum(vis_out * vm, dim=1) / vsum
    return text_out, video_out


def cross_similarity(seq_out, vis_out, attention_mask, video_mask, P, cfg, training):
    """modeling.py:341-375: every (text, video) pair through the cross encoder, 5 text rows per chunk."""
    bt, st, h = seq_out.shape
    bv, sv, _ = vis_out.shape
    rows = []
    for lo in range(0, bt, 5):
        srow = seq_out[lo:lo + 5]
        mrow = attention_mask[lo:lo + 5]
        n = srow.size(0)
        sl = srow.unsqueeze(1).repeat(1, bv, 1, 1).view(-1, st, h)
        ml = mrow.unsqueeze(1).repeat(1, bv, 1).view(-1, st)
        vr = vis_out.unsqueeze(0).repeat(n, 1, 1, 1).view(-1, sv, h)
        vmr = video_mask.unsqueeze(0).repeat(n, 1, 1).view(-1, sv)
        _, pooled, _ = get_cross_output(sl, vr, ml, vmr, P, cfg, training)
        rows.append(linear(pooled, P, "similarity_dense").squeeze(-1).view(n, bv))
    return torch.cat(rows, dim=0)


def similarity_logits(seq_out, vis_out, attention_mask, video_mask, P, cfg, training, pretrain_joint=False):
    """modeling.py:377-391."""
    if (cfg.stage_two and not pretrain_joint) or cfg.sim_after_cross:
        return cross_similarity(seq_out, vis_out, attention_mask, video_mask, P, cfg, training)
    t, v = mean_pooling_for_similarity(seq_out, vis_out, attention_mask, video_mask)
    if not cfg.use_mil:
        t = F.normalize(t, dim=-1)
        v = F.normalize(v, dim=-1)
    return torch.matmul(t, v.t())


def cross_en(sim):
    """until_module.py:186-191."""
    return (-torch.diag(F.log_softmax(sim, dim=-1))).mean()


def mil_nce_loss(sim, batch_size, n_pair):
    """until_module.py:201-221."""
    mm = torch.tensor(np.kron(np.eye(batch_size), np.ones((n_pair, n_pair)))).to(sim.dtype)
    from_text = sim + mm * -1e12
    from_video = sim.transpose(1, 0)
    new = torch.cat([from_video, from_text], dim=-1)
    logpt = F.log_softmax(new, dim=-1)
    mm2 = torch.cat([mm, torch.zeros_like(mm)], dim=-1)
    masked = logpt + (torch.ones_like(mm2) - mm2) * -1e12
    new_logpt = -torch.logsumexp(masked, dim=-1)
    sel = torch.arange(batch_size) * n_pair + (n_pair // 2)
    return new_logpt[sel].mean()


def max_margin_ranking_loss(x, margin, negative_weighting, batch_size, n_pair, hard_negative_rate):
    """until_module.py:223-251."""
    d = torch.diag(x)
    mmg = F.relu(margin + x - d.view(-1, 1)) + F.relu(margin + x - d.view(1, -1))
    if negative_weighting and n_pair > 1 and batch_size > 1:
        easy = 1 - hard_negative_rate
        alpha = easy / ((batch_size - 1) * (1 - easy))
        mm = (1 - alpha) * np.eye(batch_size) + alpha
        mm = np.kron(mm, np.ones((n_pair, n_pair)))
        mm = (torch.tensor(mm) * (batch_size * (1 - easy))).float().to(x.dtype)
        mmg = mmg * mm
    return mmg.mean()


def _loss_fcts(cfg: OracleConfig):
    """modeling.py:172-184."""
    bs = cfg.batch_size // cfg.n_gpu
    mil = lambda s: mil_nce_loss(s, bs, cfg.n_pair)
    mmr = lambda s: max_margin_ranking_loss(s, cfg.margin, cfg.negative_weighting, bs, cfg.n_pair,
                                            cfg.hard_negative_rate)
    if cfg.use_mil:
        return (cross_en if cfg.stage_two else mil), mil
    return (cross_en if cfg.stage_two else mmr), mmr


def cross_entropy_ignore(logits, labels, ignore_index=-1):
    return F.cross_entropy(logits, labels, ignore_index=ignore_index)


def mfm_loss(vis_cross_out, video, video_mask, video_labels_index, P):
    """modeling.py:278-297."""
    scores = lm_head(vis_cross_out, P, "cls_visual.predictions",
                     P["visual.embeddings.word_embeddings.weight"], transposed=True)
    scores_tr = scores.view(-1, scores.shape[-1])
    video_tr = video.permute(2, 0, 1)
    video_tr = video_tr.reshape(video_tr.shape[0], -1)
    logits = torch.mm(scores_tr, video_tr)
    vm = video_mask.to(logits.dtype)
    mask_matrix = torch.mm(vm.view(-1, 1), vm.view(1, -1))
    masked_logits = logits + (1. - mask_matrix) * -1e8
    logpt = torch.diag(F.log_softmax(masked_logits, dim=-1))
    sel = (video_labels_index != -1).view(-1)
    return (-logpt)[sel].mean()


# ------------------------------------------------------------------------------------- full forward


def get_sequence_visual_output(P, cfg, input_ids, token_type_ids, attention_mask, video, video_mask,
                               training=False, shaped=False):
    """modeling.py:299-313."""
    if not shaped:
        input_ids = input_ids.view(-1, input_ids.shape[-1])
        token_type_ids = token_type_ids.view(-1, token_type_ids.shape[-1])
        attention_mask = attention_mask.view(-1, attention_mask.shape[-1])
        video_mask = video_mask.view(-1, video_mask.shape[-1])
        video = normalize_video(video, P)
    seq = bert_model(input_ids, token_type_ids, attention_mask, P, cfg, training)
    vis = visual_model(video, video_mask, P, cfg, training)
    return seq, vis


def univl_forward(P, cfg: OracleConfig, batch, training=True, return_parts=False, sim_loss_fct=None):
    """modeling.py:188-271 -- returns the scalar training loss (all stages / losses).  sim_loss_fct: stands in for the reference's
    `self.loss_fct` (the loss on the similarity matrix, modeling.py:209 / :265) -- the cotangent fixtures of oracle/make_golden.py
    replace it by sim -> (sim * W).sum()."""
    v = lambda t: t.view(-1, t.shape[-1])
    input_ids, token_type_ids = v(batch["input_ids"]), v(batch["token_type_ids"])
    attention_mask, video_mask = v(batch["attention_mask"]), v(batch["video_mask"])
    video = normalize_video(batch["video"], P)
    seq, vis = get_sequence_visual_output(P, cfg, input_ids, token_type_ids, attention_mask, video, video_mask,
                                          training, shaped=True)
    parts = dict(sequence_output=seq, visual_output=vis)
    loss_fct, pretrain_sim_fct = _loss_fcts(cfg)
    if sim_loss_fct is not None:
        loss_fct = sim_loss_fct
    loss = 0.
    if cfg.stage_one:
        sim = similarity_logits(seq, vis, attention_mask, video_mask, P, cfg, training)
        parts["sim_matrix"] = sim
        loss = loss + loss_fct(sim)
    if cfg.stage_two:
        seq_alm = vis_alm = None
        if cfg.do_pretrain:
            masked_text = v(batch["pairs_masked_text"])
            token_labels = v(batch["pairs_token_labels"])
            masked_video = normalize_video(batch["masked_video"], P)
            vlabels = v(batch["video_labels_index"])
            seq_alm, vis_alm = get_sequence_visual_output(P, cfg, masked_text, token_type_ids, attention_mask,
                                                          masked_video, video_mask, training, shaped=True)
            cross_out, _, _ = get_cross_output(seq_alm, vis_alm, attention_mask, video_mask, P, cfg, training)
            seq_cross, vis_cross = torch.split(cross_out, [attention_mask.size(-1), video_mask.size(-1)], dim=1)
            alm_scores = lm_head(seq_cross, P, "cls.predictions", P["bert.embeddings.word_embeddings.weight"])
            parts["alm_loss"] = cross_entropy_ignore(alm_scores.view(-1, cfg.vocab_size), token_labels.view(-1))
            loss = loss + parts["alm_loss"]
            parts["nce_loss"] = mfm_loss(vis_cross, video, video_mask, vlabels, P)
            loss = loss + parts["nce_loss"]
            sim_j = similarity_logits(seq, vis, attention_mask, video_mask, P, cfg, training, pretrain_joint=True)
            parts["sim_joint"] = sim_j
            loss = loss + pretrain_sim_fct(sim_j)
        if batch.get("input_caption_ids") is not None and (cfg.do_pretrain or cfg.task_type == "caption"):
            cap_in, dmask = v(batch["input_caption_ids"]), v(batch["decoder_mask"])
            s_in, v_in = (seq_alm, vis_alm) if cfg.do_pretrain else (seq, vis)
            cross_out, _, cmask = get_cross_output(s_in, v_in, attention_mask, video_mask, P, cfg, training)
            scores = decoder_model(cap_in, cross_out, dmask, cmask, P, cfg, training)
            parts["decoder_scores"] = scores
            cap_out = v(batch["output_caption_ids"])
            parts["decoder_loss"] = cross_entropy_ignore(scores.view(-1, cfg.vocab_size), cap_out.view(-1))
            loss = loss + parts["decoder_loss"]
        if cfg.do_pretrain or cfg.task_type == "retrieval":
            s_in, v_in = (seq_alm, vis_alm) if cfg.do_pretrain else (seq, vis)
            sim_tv = similarity_logits(s_in, v_in, attention_mask, video_mask, P, cfg, training)
            parts["sim_matrix"] = sim_tv
            loss = loss + loss_fct(sim_tv)
    if return_parts:
        return loss, parts
    return loss


def decoder_caption(P, cfg, seq_out, vis_out, attention_mask, video_mask, input_caption_ids, decoder_mask):
    """modeling.py:409-428 with get_logits=True."""
    v = lambda t: t.view(-1, t.shape[-1])
    attention_mask, video_mask = v(attention_mask), v(video_mask)
    cross_out, _, cmask = get_cross_output(seq_out, vis_out, attention_mask, video_mask, P, cfg, False)
    return decoder_model(v(input_caption_ids), cross_out, v(decoder_mask), cmask, P, cfg, False)


# --------------------------------------------------------------------------------------- optimizer


def warmup_linear(x, warmup=0.002):
    """optimization.py:38-43."""
    if x < warmup:
        return x / warmup
    return max((x - 1.) / (warmup - 1.), 0)


def clip_grad_norm_(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (called at main_task_retrieval.py:347 and optimization.py:135-136):
    total L2 norm, coef = max_norm / (norm + 1e-6) clamped to 1, grads scaled in place."""
    grads = [g for g in grads if g is not None]
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g.to(torch.float32)) for g in grads]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef.to(g.dtype))
    return total


def bert_adam_step(p, g, m, v, step, lr, warmup, t_total, weight_decay, b1=0.9, b2=0.999, e=1e-6,
                   max_grad_norm=1.0):
    """optimization.py:103-168 for one parameter tensor (in place): per-param clip, Adam moments WITHOUT bias
    correction, decoupled weight decay, warmup_linear LR.  Returns the new step count."""
    if max_grad_norm > 0:
        clip_grad_norm_([g], max_grad_norm)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    update = m / (v.sqrt() + e)
    if weight_decay > 0.0:
        update = update + weight_decay * p
    lr_s = lr * warmup_linear(step / t_total, warmup) if t_total != -1 else lr
    p.add_(-lr_s * update)
    return step + 1


def param_groups(names, lr, coef_lr=1.0):
    """main_task_retrieval.py:173-190: ('bert.' -> lr*coef_lr) x (bias/LayerNorm -> weight_decay 0)."""
    no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']
    out = {}
    for n in names:
        wd = 0.0 if any(nd in n for nd in no_decay) else 0.01
        out[n] = dict(weight_decay=wd, lr=lr * coef_lr if "bert." in n else lr)
    return out


def compute_metrics(x):
    """metrics.py:8-20 restated (numpy): positions of the diagonal score in each descending-sorted row -- ties with the
    diagonal contribute one entry each, exactly like the reference's `np.where(sx - d == 0)`."""
    import numpy as np
    x = np.asarray(x)
    sx = np.sort(-x, axis=1)
    d = np.diag(-x)[:, np.newaxis]
    ind = np.where(sx - d == 0)[1]
    return dict(R1=float(np.sum(ind == 0)) / len(ind), R5=float(np.sum(ind < 5)) / len(ind),
                R10=float(np.sum(ind < 10)) / len(ind), MR=np.median(ind) + 1)


def beam_search_caption(P, cfg, seq_out, vis_out, attention_mask, video_mask, n_bm, max_words, bos, eos):
    """The caption decoding procedure of main_task_caption.py:434-618 + modules/beam.py restated: every step re-runs
    decoder_caption on the complete prefixes of all beams of the still-active instances and keeps the last position's
    log-softmax; per instance: first step top-k of beam 0's distribution (beam.py:69), later steps top-k over the
    flattened (beam x vocab) sums (beam.py:67,71-72), back-pointer = id // vocab, token = id % vocab (beam.py:77-79),
    done when the top beam emits EOS (beam.py:84); result = best-scored beam walked back (beam.py:108-116,
    collect_hypothesis_and_scores n_best=1).  Returns (list of token lists, list of best scores)."""
    n = seq_out.shape[0]
    v = lambda t: t.view(-1, t.shape[-1])
    attention_mask, video_mask = v(attention_mask), v(video_mask)
    scores = [torch.zeros(n_bm) for _ in range(n)]
    prev_ks = [[] for _ in range(n)]
    next_ys = [[torch.full((n_bm,), bos, dtype=torch.long)] for _ in range(n)]
    done = [False] * n

    def hypothesis(i, k):
        hyp = []
        for j in range(len(prev_ks[i]) - 1, -1, -1):
            hyp.append(int(next_ys[i][j + 1][k]))
            k = int(prev_ks[i][j][k])
        return hyp[::-1]

    for length in range(1, max_words + 1):
        active = [i for i in range(n) if not done[i]]
        if not active:
            break
        rows, owner = [], []
        for i in active:
            if len(next_ys[i]) == 1:
                seqs = next_ys[i][0].unsqueeze(1)
            else:
                keys = torch.sort(scores[i], 0, True)[1]
                seqs = torch.tensor([[bos] + hypothesis(i, int(k)) for k in keys], dtype=torch.long)
            rows.append(seqs)
            owner += [i] * n_bm
        ids = torch.cat(rows, 0)
        idx = torch.tensor(owner)
        logits = decoder_caption(P, cfg, seq_out[idx], vis_out[idx], attention_mask[idx], video_mask[idx], ids,
                                 torch.ones_like(ids))
        word_prob = F.log_softmax(logits[:, -1, :], dim=1).view(len(active), n_bm, -1)
        V = word_prob.shape[-1]
        for pos, i in enumerate(active):
            wp = word_prob[pos]
            beam_lk = wp + scores[i].unsqueeze(1) if prev_ks[i] else wp[0]
            best, best_id = beam_lk.reshape(-1).topk(n_bm, 0, True, True)
            scores[i] = best
            pk = best_id // V
            prev_ks[i].append(pk)
            next_ys[i].append(best_id - pk * V)
            if int(next_ys[i][-1][0]) == eos:
                done[i] = True
    hyps, best_scores = [], []
    for i in range(n):
        sc, order = torch.sort(scores[i], 0, True)
        hyps.append(hypothesis(i, int(order[0])))
        best_scores.append(float(sc[0]))
    return hyps, best_scores
