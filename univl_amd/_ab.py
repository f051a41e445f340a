"""The ONE place measurement / A-B overrides of the plan builder come from.

Every plan choice in `univl_amd` is a function of the step descriptor (shapes, dtype, world size); its default is the measured-best
setting and is what production runs.  For A/B measurements and for tests that must reach a non-default path, the defaults can be
overridden through a single environment variable

    UNIVL_AB="key=value,key=value"          e.g.  UNIVL_AB="ln_fold=0,splitk_tiles=0"

parsed here, once per process.  Unknown keys are an error (a typo must not silently measure the default twice), and the variable is
REFUSED unless the process has declared itself a measurement / test harness with `_ab.allow()` -- `bench.py`, `tests/conftest.py` and
the scripts under `scripts/` do; a training script that inherits a stray UNIVL_AB from its shell fails at model construction instead
of training a non-default plan.

`KEYS` documents every override with its default; `get(key)` returns the override (converted to the default's type) or the default.
"""
import os

# key -> (default, what it selects)
KEYS = {
    # ---- GEMM plan (engine.EncoderStack / DecoderStack, steps.py)
    "wgrad_ride": (1, "weight gradients ride in their dgrad's launch (univl_gemm_pair); 0: the layer's grouped launch"),
    "group_wgrad": (1, "a layer's non-riding weight gradients as ONE grouped launch; 0: one launch per member"),
    "pair_form": ("", "'square': round 2's 64 x 64 form of the pair launch"),
    "decoder_pair": (1, "pair launches in the decoder stack"),
    "splitk_tiles": (128, "split the contraction of the N = 768 products while the 64 x 64 output grid has fewer tiles than this"),
    "splitk_len": (384, "target slice depth of that split"),
    "splitk_maxwg": (512, "cap on tiles x slices of that split"),
    "splitk_dgrad_maxwg": (0, "cap on tiles x slices of the dgrad half of a pair launch (0: splitk_maxwg) -- so that dgrad + weight-gradient tiles fit one round of resident slots"),
    "splitk_target_wg": (180, "up to 48 output tiles (256 tokens): split the N = 768 products into round(target / tiles) slices of at least 256 (0: the splitk_len / splitk_maxwg policy of rounds 1-5)"),
    "splitk_target_wg_big": (264, "the same from 49 tiles up to splitk_tiles"),
    "ks_o_fwd": (0, "A/B: slices of the attention-output product (0: policy)"), "ks_ffn2_fwd": (0, "A/B: slices of the FFN2 forward product"),
    "ks_ffn1_dgrad": (0, "A/B: slices of the FFN1 dgrad"), "ks_qkv_dgrad": (0, "A/B: slices of the QKV dgrad"),
    "splitk_mid": (1, "three-slice split of the deep (K >= 2304) N = 768 products at 128 .. 255 tiles"),
    "splitk_mid_ks": (3, "slices of the deep (K >= 2304) N = 768 products in the splitk_mid regime up to 160 tiles"),
    "splitk_mid_ks_big": (2, "... from 161 tiles on"),
    "splitk_mid_ks_768": (1, "slices of the K = 768 N = 768 products in the splitk_mid regime"),
    "splitk_mid_tiles": (256, "upper tile bound of that regime"),
    "ln_fold": (1, "LayerNorm forward finished inside its product's launch (univl_gemm_ln) up to ln_fold_max_rows tokens"),
    "ln_fold_max_rows": (640, "... up to this many tokens (the library carries up to 1024).  Round 6, alternating on one box: 576 tokens 3.00 -> 2.92 ms (-2.6 %, both folds), 624 tokens -1.7 %, 672 +-0, 768 +0.5 % (profiles/r06u_ab_fold_768*.txt); rounds 4-5: 512"),
    "ln_fold_bwd": (1, "LayerNorm backward finished inside the dgrad pair launch (univl_gemm_pair_ln)"),
    "ln_fold_bwd_max": (640, "... up to this many tokens (383: round 4's range, the square pair form only)"),
    "wgrad_big_min": (0, "token count from which a layer's grouped weight gradients take the big tile (0: where every dgrad does)"),
    "attn_fuse_fwd": (1, "the q | k | v projection computed inside the attention forward launch (univl_attention_fwd_fused); 0: two launches"),
    "attn_fuse_bwd": (1, "the attention-output dgrad computed inside the attention backward launch (univl_attention_bwd_fused); 0: two launches"),
    "attn_fuse_fwd_max_seq": (64, "longest sequence the fused attention FORWARD launch is used for.  128: the round-6 form for 65 .. 128 positions (two workgroups per (batch row, head), bit-identical) -- measured NEUTRAL: caption 5.150 vs 5.150 ms, pretrain 9.09 vs 9.09, FT-Align 2.90 vs 2.90 with 18 launches fewer (profiles/r06t_ab_attn_big.txt), so the plans keep the two launches"),
    "attn_fuse_fwd_max_rows": (1536, "token count up to which the fused attention FORWARD launch is used (1536: -1.9 %, 3072: +0.7 %, profiles/r05ae)"),
    "attn_fuse_bwd_max_rows": (1535, "token count up to which the fused attention BACKWARD launch is used (1536: +1 %, 6144: +4.9 %)"),
    "g256": (1, "256 x 256 8-phase body (csrc/gemm256.h) for the grouped weight gradients / single products it is picked for; 0: the older tiles"),
    "g256_min_rows": (1536, "token count (a multiple of 256) from which the plans drop the pair launches for separate dgrads + the grouped 256-body launch"),
    "pairs": ("", "default of model.operand_pairs -- operand PAIRS (bf16 hi + lo, include/univl_hip.h: UnivlGemm.A_lo) in the encoder stacks' forward products: 'x' the activation operand, 'w' the weight operand, 'xw' both, '': plain bf16 operands"),
    "pairs_max_rows": (768, "... in stacks of at most this many tokens (16 pairs x 48): beyond, the matrix pipe is the bound and the extra terms cost step time"),
    "pairs_ks": ("same", "split of a paired N = 768 product: 'same' workgroup count as the plain product (slices nterm x as deep) | 'terms': every term cut like the plain product"),
    "gelu_pre_f32": (0, "A/B measurement: the FFN1 pre-activation saved for GELU' in fp32 instead of the compute type (encoder stacks)"),
    "vocab_ce": (1, "K16: the vocabulary classifier with the online log-softmax CE in its epilogue (univl_vocab_ce_fwd / _bwd: no [tokens, 30522] logits in training); 0: product -> logits -> univl_ce_loss"),
    "vocab_dgrad_split": (1, "split-K of the vocabulary dgrad (caption / pretrain heads)"),
    "fused_sim": (1, "pooling + similarity + loss heads as fused launches up to 256 rows"),
    "dpos_gather_min": (32, "rows per position from which position-table gradients are gathered instead of scatter-added"),
    # ---- optimizer / step structure
    "adam_ride": ("1", "BertAdam chunks ride with the next forward's products; '0': side-stream form; 'force': one graph even with a captured exchange"),
    "tail_ride": (1, "chunks of cross layer 0 / decoder layer 0 ride in the last text / video layer's products; 0: launched in front of the forward"),
    "adam_lazy_rows": (1, "weight-decay-only shortcut for word-table rows that never had a gradient"),
    "adam_chunk": (0, "A/B: elements per workgroup of the BertAdam update (0: optimization.CHUNK = 8192)"),
    "adam_blocks": (0, "grid cap of the overlapped (non-riding) update"),
    "pipeline_opt": (0, "experimental pipelined optimizer"),
    "async_loss": (0, "experimental asynchronous loss read-back"),
    "fused_norms": (1, "gradient norms from the weight-gradient epilogues"),
    "sparse_rows": (1, "sparse bookkeeping of the word-table gradient"),
    "sparse_emb": (1, "sparse exchange of the word-table gradient under data parallelism"),
    "copy_kernel": (1, "input staging / loss hand-off as one copy kernel"),
    "auto_graph": (1, "the unchanged training loop switches to graph replay on its own"),
    "auto_dp": (1, "wrap in the bucketed reducer on its own when torch.distributed is initialised"),
    "dp_capture": (1, "capture the RCCL exchange into the step graph"),
    "dp_dryrun": (0, "measurement: the data-parallel schedule with the collectives not enqueued"),
    # ---- measurement probes (results are meaningless)
    "stamps": (0, "device timestamps between the nodes of the step"),
    "probe_skip": ("", "a stack (bert | visual | cross) without its layer kernels"),
    "probe_no_ln": ("", "fwd | bwd | both: the encoder LayerNorm launches left out"),
    "poison": (0, "workspaces filled with NaN patterns"),
    "guard": (0, "guard words around workspaces"),
}

_allowed = False
_parsed = None
_legacy_checked = False

# Operational switches (not measurements): a training process sets them on the model -- `model.auto_graph`, `model.auto_dp`,
# `model.dp_capture`, `model.auto_ride` (univl_amd.UniVL attributes, all default True) -- UNIVL_AB stays the harness-only override.
OPERATIONAL = {"auto_graph": "model.auto_graph = False", "auto_dp": "model.auto_dp = False", "dp_capture": "model.dp_capture = False",
               "adam_ride": "model.auto_ride = False"}


def _check_legacy():
    """Rounds 1-4 read one environment variable per switch (UNIVL_AUTO_DP, UNIVL_DP_CAPTURE, UNIVL_ADAM_RIDE, UNIVL_LN_FOLD, ...).  They
    are gone; a process that still sets one would silently run the default -- the failure this module exists to refuse -- so it is an
    error that names the replacement."""
    global _legacy_checked
    if _legacy_checked:
        return
    _legacy_checked = True
    for k in KEYS:
        name = "UNIVL_" + k.upper()
        if name in os.environ:
            how = ("set `%s` on the model (or UNIVL_AB=%s=... in a measurement harness)" % (OPERATIONAL[k], k)) if k in OPERATIONAL \
                else ("use UNIVL_AB=%s=... (measurement / test harnesses only)" % k)
            raise RuntimeError("%s is set but no longer read (it would be silently ignored): %s" % (name, how))


def allow():
    """Declare this process a measurement / test harness: UNIVL_AB is honoured from here on."""
    global _allowed
    _allowed = True


def _parse():
    global _parsed
    _check_legacy()
    raw = os.environ.get("UNIVL_AB", "")
    if _parsed is not None and _parsed[0] == raw:
        return _parsed[1]
    out = {}
    for item in [x for x in raw.split(",") if x.strip()]:
        if "=" not in item:
            raise RuntimeError("UNIVL_AB: expected key=value, got %r" % item)
        k, v = (s.strip() for s in item.split("=", 1))
        if k not in KEYS:
            raise RuntimeError("UNIVL_AB: unknown key %r (known: %s)" % (k, ", ".join(sorted(KEYS))))
        d = KEYS[k][0]
        out[k] = type(d)(v) if not isinstance(d, int) else int(v)
    if out and not _allowed:
        raise RuntimeError("UNIVL_AB=%r is set, but this process is not a measurement or test harness (univl_amd._ab.allow()): "
                           "A/B overrides of the plan builder are refused in training / evaluation runs" % raw)
    _parsed = (raw, out)
    return out


def get(key):
    if key not in KEYS:
        raise KeyError("univl_amd._ab: unknown key %r" % key)
    return _parse().get(key, KEYS[key][0])


def overrides():
    """The overrides in effect (for a bench line's `config`)."""
    return dict(_parse())
