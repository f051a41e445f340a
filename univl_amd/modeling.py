"""MI355X-native mirror of the reference's model facade `modules/modeling.py::UniVL`.

Same public surface: `UniVL.from_pretrained()`, `forward()` -> scalar loss,
`get_sequence_visual_output()`, `get_similarity_logits()`, `decoder_caption()`, the attributes the scripts read
(`_stage_one`, `_stage_two`, `train_sim_after_cross`, `task_config`), and the reference's parameter names and
shapes so that `univl.pretrained.bin` / saved checkpoints interchange.  The sub-modules below are PARAMETER
CONTAINERS (their names reproduce the reference's state_dict keys); all arithmetic runs in libunivl_hip.so through
the static plans of `univl_amd.engine` -- there is no PyTorch compute path and no CPU fallback.
"""
import json
import logging
import os

import numpy as np
import torch
from torch import nn

from . import _lib, ops
from .engine import EncoderStack, FlatParams, Plan, _SiteCounter, _gemm_desc
from .parallel import BucketReducer, broadcast_parameters, layer_buckets

logger = logging.getLogger(__name__)

# defaults of modules/{visual,cross,decoder}-base/*.json and of BERT-base (module_bert.py:61-72)
BERT_BASE = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                 hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                 max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02)
VISUAL_BASE = dict(BERT_BASE, vocab_size=1024, num_hidden_layers=1)
CROSS_BASE = dict(BERT_BASE, vocab_size=768, num_hidden_layers=2, max_position_embeddings=1024)
DECODER_BASE = dict(BERT_BASE, num_decoder_layers=1, max_target_embeddings=512)


class _Config(object):
    def __init__(self, d):
        self.__dict__.update(d)

    @classmethod
    def resolve(cls, name, defaults, config_file):
        """until_config.py:41-99: a directory holding <config_file> overrides the built-in defaults."""
        d = dict(defaults)
        for cand in (name, os.path.join(os.path.dirname(os.path.abspath(__file__)), str(name))):
            f = os.path.join(str(cand), config_file)
            if os.path.isfile(f):
                with open(f, "r", encoding="utf-8") as r:
                    d.update(json.load(r))
                break
        return cls(d)


def _check_attr(name, cfg):
    return hasattr(cfg, name) and cfg.__dict__[name]


class LayerNorm(nn.Module):
    """Parameter container for until_module.py:40-53 (weight, bias; eps 1e-12)."""

    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps


class _SelfAttentionParams(nn.Module):
    def __init__(self, H):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)


class _SelfOutputParams(nn.Module):
    def __init__(self, H_in, H):
        super().__init__()
        self.dense = nn.Linear(H_in, H)
        self.LayerNorm = LayerNorm(H)


class _AttentionParams(nn.Module):
    def __init__(self, H):
        super().__init__()
        self.self = _SelfAttentionParams(H)
        self.output = _SelfOutputParams(H, H)


class _IntermediateParams(nn.Module):
    def __init__(self, H, I):
        super().__init__()
        self.dense = nn.Linear(H, I)


class _LayerParams(nn.Module):
    """BertLayer / VisualLayer / CrossLayer (module_bert.py:253-264)."""

    def __init__(self, H, I):
        super().__init__()
        self.attention = _AttentionParams(H)
        self.intermediate = _IntermediateParams(H, I)
        self.output = _SelfOutputParams(I, H)


class _EncoderParams(nn.Module):
    def __init__(self, H, I, L):
        super().__init__()
        self.layer = nn.ModuleList([_LayerParams(H, I) for _ in range(L)])


class _PoolerParams(nn.Module):
    def __init__(self, H):
        super().__init__()
        self.dense = nn.Linear(H, H)


class _BertEmbeddingsParams(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.word_embeddings = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)
        self.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, cfg.hidden_size)
        self.LayerNorm = LayerNorm(cfg.hidden_size)


class BertModel(nn.Module):
    """Parameters of module_bert.py:364-414."""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.embeddings = _BertEmbeddingsParams(cfg)
        self.encoder = _EncoderParams(cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers)
        self.pooler = _PoolerParams(cfg.hidden_size)


class _VisualEmbeddingsParams(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.word_embeddings = nn.Linear(cfg.vocab_size, cfg.hidden_size)
        self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)
        self.LayerNorm = LayerNorm(cfg.hidden_size)


class VisualModel(nn.Module):
    """Parameters of module_visual.py:346-395."""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.embeddings = _VisualEmbeddingsParams(cfg)
        self.encoder = _EncoderParams(cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers)
        self.pooler = _PoolerParams(cfg.hidden_size)


class NormalizeVideo(nn.Module):
    """modeling.py:83-92."""

    def __init__(self, task_config):
        super().__init__()
        self.visual_norm2d = LayerNorm(task_config.video_dim)


class UniVLPreTrainedModel(nn.Module):
    def __init__(self):
        super().__init__()

    @classmethod
    def from_pretrained(cls, pretrained_bert_name, visual_model_name, cross_model_name, decoder_model_name,
                        state_dict=None, cache_dir=None, type_vocab_size=2, *inputs, **kwargs):
        """modeling.py:56-81."""
        task_config = kwargs.get("task_config", None)
        if task_config is not None:
            if not hasattr(task_config, "local_rank"):
                task_config.__dict__["local_rank"] = 0
            elif task_config.local_rank == -1:
                task_config.local_rank = 0
        bert_config = _Config.resolve(pretrained_bert_name, BERT_BASE, "bert_config.json")
        bert_config.type_vocab_size = type_vocab_size
        if state_dict is None and os.path.isfile(os.path.join(str(pretrained_bert_name), "pytorch_model.bin")):
            state_dict = torch.load(os.path.join(str(pretrained_bert_name), "pytorch_model.bin"), map_location="cpu")
        visual_config = _Config.resolve(visual_model_name, VISUAL_BASE, "visual_config.json")
        cross_config = _Config.resolve(cross_model_name, CROSS_BASE, "cross_config.json")
        decoder_config = _Config.resolve(decoder_model_name, DECODER_BASE, "decoder_config.json")
        for c in (visual_config, cross_config, decoder_config):
            c.type_vocab_size = type_vocab_size
        model = cls(bert_config, visual_config, cross_config, decoder_config, *inputs, **kwargs)
        assert model.bert is not None and model.visual is not None
        if state_dict is not None:
            model = cls.init_preweight(model, state_dict, task_config=task_config)
        return model

    @classmethod
    def init_preweight(cls, model, state_dict, prefix=None, task_config=None):
        """until_module.py:91-146: gamma/beta -> weight/bias renames, optional prefix, report missing/unexpected."""
        sd = {}
        for k, v in state_dict.items():
            nk = k.replace("gamma", "weight") if "gamma" in k else k
            nk = nk.replace("beta", "bias") if "beta" in nk else nk
            sd[(prefix + nk) if prefix is not None else nk] = v
        res = model.load_state_dict(sd, strict=False)
        if prefix is None and (task_config is None or task_config.local_rank == 0):
            if res.missing_keys:
                logger.info("Weights of %s not initialized from pretrained model: %s", model.__class__.__name__, res.missing_keys)
            if res.unexpected_keys:
                logger.info("Weights from pretrained model not used in %s: %s", model.__class__.__name__, res.unexpected_keys)
        model.mark_params_dirty()
        return model


class _JointStep:
    """Static plans + workspace of the stage-one (FT-Joint) training step for fixed (rows, max_words, max_frames)."""
    pass


class _JointLossFn(torch.autograd.Function):
    """loss = UniVL.forward(...) for the stage-one FT-Joint path: the whole forward is one plan, the whole backward
    another; parameter gradients are written straight into the flat gradient buffer (p.grad are views of it)."""

    @staticmethod
    def forward(ctx, anchor, model, step):
        step.fwd.run()
        ctx.model, ctx.step = model, step
        return step.loss[0].clone()

    @staticmethod
    def backward(ctx, gout):
        ctx.model._run_backward(ctx.step, gout)
        return None, None, None


class _LossTensor(torch.Tensor):
    """The scalar returned by UniVL.forward.  It is an ordinary autograd tensor (grad_fn = _JointLossFn), but a plain
    `loss.backward()` -- what main_task_retrieval.py:342 does -- runs the backward plan directly in the calling
    thread instead of going through the autograd engine's device thread (less overhead, and a multi-stream plan can
    then be captured into a hipGraph together with the forward).  Anything else (`(loss / k).backward()`,
    `loss.mean()`, explicit gradients) takes the normal autograd route and ends in _JointLossFn.backward."""

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        ms = self.__dict__.get("_univl", None)
        if ms is not None and gradient is None and inputs is None and not create_graph:
            model, step = ms
            model._run_backward(step, None)
            return None
        return super().backward(gradient=gradient, retain_graph=retain_graph, create_graph=create_graph, inputs=inputs)


class UniVL(UniVLPreTrainedModel):
    def __init__(self, bert_config, visual_config, cross_config, decoder_config, task_config):
        super().__init__()
        self.bert_config, self.visual_config = bert_config, visual_config
        self.cross_config, self.decoder_config = cross_config, decoder_config
        self.task_config = task_config
        self.ignore_video_index = -1
        tc = task_config
        assert tc.max_words <= bert_config.max_position_embeddings
        assert tc.max_words <= decoder_config.max_target_embeddings
        assert tc.max_frames <= visual_config.max_position_embeddings
        assert tc.max_words + tc.max_frames <= cross_config.max_position_embeddings
        assert bert_config.hidden_size == 768 and bert_config.num_attention_heads == 12 and \
            bert_config.intermediate_size == 3072, "kernels are specialised for H=768, 12 heads, I=3072"

        self._stage_one, self._stage_two = True, False
        if _check_attr("stage_two", tc):
            self._stage_one, self._stage_two = False, tc.stage_two
        self.train_sim_after_cross = bool(self._stage_one and _check_attr("train_sim_after_cross", tc))

        if hasattr(tc, "text_num_hidden_layers"):
            bert_config.num_hidden_layers = tc.text_num_hidden_layers
        self.bert = BertModel(bert_config)
        if hasattr(tc, "visual_num_hidden_layers"):
            visual_config.num_hidden_layers = tc.visual_num_hidden_layers
        self.visual = VisualModel(visual_config)
        self.cross, self.decoder = None, None
        if self._stage_one is False or self.train_sim_after_cross:
            raise NotImplementedError(
                "univl_amd round 1 implements the stage-one FT-Joint retrieval path (SURVEY.md section 8); the cross "
                "encoder / decoder paths (--train_sim_after_cross, --stage_two) are the next rows of section 8(a)")
        self.normalize_video = NormalizeVideo(tc)
        self.apply(self.init_weights)

        self._flat = None
        self._steps = {}
        self._reducer = None
        self._seed_dev = None
        self._seed = int(getattr(tc, "seed", 42))
        dt = getattr(tc, "compute_dtype", None) or os.environ.get("UNIVL_COMPUTE_DTYPE", "bf16")
        self.compute_dtype = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32,
                              "float32": torch.float32}[str(dt).replace("torch.", "")]
        self.dropout_prob = float(getattr(tc, "dropout_prob", bert_config.hidden_dropout_prob))

    # ------------------------------------------------------------------------------------- housekeeping
    def init_weights(self, module):
        """until_module.py:70-85."""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.bert_config.initializer_range)
        elif isinstance(module, LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._flat, self._steps = None, {}      # parameters were re-allocated (e.g. .to(device)): re-flatten lazily
        return r

    def mark_params_dirty(self):
        """Tell the model its fp32 parameters were modified outside univl_amd.optimization.BertAdam (the bf16
        shadow the GEMMs read is refreshed before the next forward)."""
        if self._flat is not None:
            self._flat.shadow_valid = False

    def load_state_dict(self, *a, **kw):
        r = super().load_state_dict(*a, **kw)
        self.mark_params_dirty()
        return r

    @property
    def flat(self):
        if self._flat is None:
            p0 = next(self.parameters())
            if not p0.is_cuda:
                raise RuntimeError("univl_amd.UniVL runs only on a HIP device (call model.to('cuda')); no CPU fallback")
            _lib.lib()
            self._flat = FlatParams(list(self.named_parameters()), p0.device, self.compute_dtype)
            self._seed_dev = torch.zeros(1, device=p0.device, dtype=torch.int64)
            self._steps = {}
        return self._flat

    def enable_data_parallel(self, process_group=None, broadcast=True):
        """Re-homes the reference's DDP wrap (main_task_retrieval.py:197-198) onto per-layer RCCL all-reduces of
        the flat gradient buffer, overlapped with backward (univl_amd.parallel).  Call after model.to(device) and
        torch.distributed.init_process_group; with world_size 1 it is a no-op."""
        fl = self.flat
        if broadcast:
            broadcast_parameters(fl.p32, 0, process_group)
            fl.shadow_valid = False
        self._reducer = BucketReducer(fl.g32, process_group)
        if self._reducer.world == 1:
            self._reducer = None
        self._steps = {}
        return self

    def used_parameter_names(self):
        """Parameters that receive a gradient on the stage-one path: everything except the two dead poolers
        (modeling.py:307,310 discard the pooled outputs)."""
        return [n for n, _ in self.named_parameters() if ".pooler." not in n]

    # ------------------------------------------------------------------------------------------ plans
    def _get_step(self, B, W, F):
        key = (B, W, F, self.training)
        st = self._steps.get(key)
        if st is None:
            st = self._build_joint_step(B, W, F, self.training)
            self._steps[key] = st
        return st

    def _build_joint_step(self, B, W, F, training):
        fl = self.flat
        dev, ct, dt = fl.device, fl.compute_dtype, fl.dt
        bf = ct == torch.bfloat16
        tc = self.task_config
        D, H = tc.video_dim, 768
        f32, i64 = torch.float32, torch.int64
        e = lambda *s, dtype=f32: torch.zeros(*s, device=dev, dtype=dtype)
        st = _JointStep()
        st.B, st.W, st.F = B, W, F
        p = self.dropout_prob if training else 0.0
        sites = _SiteCounter()
        # static inputs
        st.ids, st.type_ids, st.amask = e(B, W, dtype=i64), e(B, W, dtype=i64), e(B, W, dtype=i64)
        st.video, st.vmask = e(B * F, D, dtype=torch.float64), e(B, F, dtype=i64)
        Tt, Tv = B * W, B * F
        # workspaces outside the encoder stacks
        st.vy, st.vst = e(Tv, D), e(Tv, 2)
        st.vn32 = e(Tv, D)
        st.vn_op = e(Tv, D, dtype=ct) if bf else st.vn32
        st.ve, st.vest = e(Tv, H), e(Tv, 2)
        st.v0_32 = e(Tv, H)
        st.v0_16 = e(Tv, H, dtype=ct) if bf else st.v0_32
        st.te, st.test = e(Tt, H), e(Tt, 2)
        st.t0_32 = e(Tt, H)
        st.t0_16 = e(Tt, H, dtype=ct) if bf else st.t0_32
        ldp = (B + 3) // 4 * 4
        st.tmean, st.tn, st.vmean, st.vn = e(B, H), e(ldp, H), e(B, H), e(ldp, H)
        st.sim, st.dsim = e(ldp, ldp), e(ldp, ldp)
        st.loss, st.gout = e(1), e(1)
        st.dtn, st.dvn = e(B, H), e(B, H)
        st.dseq, st.dvis = e(Tt, H), e(Tv, H)
        st.de_op = e(Tv, H, dtype=ct)
        st.dvnorm = e(Tv, D)
        # streams: 0 = text chain (+ everything serial), 1 = text weight gradients, 2 = video chain, 3 = video wgrads
        dual = os.environ.get("UNIVL_DUAL_ENCODER", "1") != "0"       # text || video encoder
        # weight gradients on a side stream: measured SLOWER on MI355X/ROCm 7.2 (4.83 vs 4.16 ms/step at bs 4: every
        # cross-stream edge of the captured hipGraph costs more than the overlap buys) and, combined with the dual
        # encoder streams, crashes hipStreamEndCapture -- kept as an opt-in experiment only.
        sidew = os.environ.get("UNIVL_SIDE_WGRAD", "0") == "1"
        ST, SV = 0, (2 if dual else 0)
        st.text = EncoderStack(fl, "bert", self.bert_config.num_hidden_layers, B, W, st.amask, p, self._seed_dev, sites,
                               s_main=ST, s_side=(1 if sidew else ST))
        st.vis = EncoderStack(fl, "visual", self.visual_config.num_hidden_layers, B, F, st.vmask, p, self._seed_dev, sites,
                              s_main=SV, s_side=(((1 if os.environ.get('UNIVL_SIDE_SHARED', '0') == '1' else 3) if dual else 1) if sidew else SV))
        off_t, off_v = sites.next(), sites.next()
        use_mil = bool(tc.use_mil)
        W32, G = fl.w32, fl.g
        nv_g, nv_b = "normalize_video.visual_norm2d.weight", "normalize_video.visual_norm2d.bias"
        vw, vb = "visual.embeddings.word_embeddings.weight", "visual.embeddings.word_embeddings.bias"
        vpos = "visual.embeddings.position_embeddings.weight"
        vlg, vlb = "visual.embeddings.LayerNorm.weight", "visual.embeddings.LayerNorm.bias"
        bw, bp, bt = ("bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight",
                      "bert.embeddings.token_type_embeddings.weight")
        blg, blb = "bert.embeddings.LayerNorm.weight", "bert.embeddings.LayerNorm.bias"

        # ------------------------------------------------------------------------------------ forward plan
        fwd = Plan()
        if p > 0:
            fwd.add_callable(lambda: ops.bump_counter(self._seed_dev))
        fwd.fork(ST, SV)           # the video encoder runs concurrently with the text encoder
        fwd.add("univl_layernorm_fwd", ops.layernorm_desc(
            dt, Tv, D, x=st.video, x_f64=True, gamma=W32(nv_g), beta=W32(nv_b), y=st.vy, stats=st.vst,
            out32=st.vn32, out16=st.vn_op if bf else None), SV)
        fwd.add("univl_gemm", _gemm_desc(dt, st.vn_op, D, fl.wop(vw), D, Tv, H, D, out32=st.ve, ldc=H, bias=W32(vb)), SV)
        fwd.add("univl_layernorm_fwd", ops.layernorm_desc(
            dt, Tv, H, x=st.ve, pos=W32(vpos), pos_period=F, gamma=W32(vlg), beta=W32(vlb), y=st.ve, stats=st.vest,
            out32=st.v0_32, out16=st.v0_16 if bf else None, p_post=p, seed=self._seed, off_post=off_v,
            seed_dev=self._seed_dev), SV)
        fwd.add("univl_embed_text_fwd", ops.embed_text_desc(
            dt, B, W, st.ids, W32(bw), W32(bp), W32(blg), W32(blb), type_ids=st.type_ids, type_emb=W32(bt), y=st.te,
            stats=st.test, out32=st.t0_32, out16=st.t0_16 if bf else None, p_post=p, seed=self._seed, off_post=off_t,
            seed_dev=self._seed_dev), ST)
        st.vis.build_forward(fwd, st.v0_32, st.v0_16, training)
        st.text.build_forward(fwd, st.t0_32, st.t0_16, training)
        fwd.join(SV, ST)
        st.seq_out, st.vis_out = st.text.output()[0], st.vis.output()[0]
        st.fwd_encoders_len = len(fwd)
        fwd.add("univl_pool_fwd", ops.pool_desc(B, W, st.seq_out, st.amask, skip_first=True, normalize=not use_mil,
                                                mean=st.tmean, out=st.tn))
        fwd.add("univl_pool_fwd", ops.pool_desc(B, F, st.vis_out, st.vmask, skip_first=False, normalize=not use_mil,
                                                mean=st.vmean, out=st.vn))
        fwd.add("univl_gemm", _gemm_desc(_lib.DT_F32, st.tn, H, st.vn, H, B, B, H, out32=st.sim, ldc=ldp))
        L = _lib.lib()
        sim_v, dsim_v = st.sim, st.dsim
        if use_mil:
            bs, npair = B // tc.n_pair, tc.n_pair
            fwd.add_callable(lambda: ops.milnce_loss(sim_v[:B], bs, npair, st.loss, dsim_v[:B]))
        else:
            wts = None
            bsz = tc.batch_size // tc.n_gpu
            if tc.negative_weighting and tc.n_pair > 1 and bsz > 1:   # until_module.py:238-243
                easy = 1 - tc.hard_negative_rate
                alpha = easy / ((bsz - 1) * (1 - easy))
                mm = np.kron((1 - alpha) * np.eye(bsz) + alpha, np.ones((tc.n_pair, tc.n_pair))) * (bsz * (1 - easy))
                wts = torch.tensor(mm, dtype=f32, device=dev).contiguous()
            st.loss_weight = wts
            margin = float(tc.margin)
            fwd.add_callable(lambda: ops.maxmargin_loss(sim_v[:B], margin, wts, st.loss, dsim_v[:B]))
        st.fwd = fwd

        # ----------------------------------------------------------------------------------- backward plans
        def build_bwd(fresh):
            bwd = Plan()
            hook = None
            red = self._reducer
            if red is not None:
                buckets = layer_buckets(fl, self.used_parameter_names())

                def hook(plan, prefix, l, stream):
                    s0, e0 = buckets["layers"][(prefix, l)]
                    plan.add_callable(lambda: red.reduce_slice(s0, e0), stream=stream)
            if fresh:
                bwd.add_callable(fl.g32[:fl.v_end].zero_)
            bwd.add_callable(lambda: ops.scale_by_device_scalar(st.dsim, st.gout))
            # d tn = dsim . vn ;  d vn = dsim^T . tn      (modeling.py:389)
            bwd.add("univl_gemm", _gemm_desc(_lib.DT_F32, st.dsim, ldp, st.vn, H, B, H, B, trans_b=1, out32=st.dtn, ldc=H))
            bwd.add("univl_gemm", _gemm_desc(_lib.DT_F32, st.dsim, ldp, st.tn, H, B, H, B, trans_a=1, trans_b=1, out32=st.dvn, ldc=H))
            bwd.add("univl_pool_bwd", ops.pool_desc(B, W, st.seq_out, st.amask, skip_first=True, normalize=not use_mil,
                                                    mean=st.tmean, out=st.tn, dout=st.dtn, dx=st.dseq))
            bwd.add("univl_pool_bwd", ops.pool_desc(B, F, st.vis_out, st.vmask, skip_first=False, normalize=not use_mil,
                                                    mean=st.vmean, out=st.vn, dout=st.dvn, dx=st.dvis))
            bwd.fork(ST, SV)       # video-encoder backward runs concurrently with the text-encoder backward
            dxv = st.vis.build_backward(bwd, st.dvis, st.v0_32, st.v0_16, fresh, training, layer_hook=hook)
            bwd.add("univl_layernorm_bwd", ops.layernorm_desc(
                dt, Tv, H, gamma=W32(vlg), y=st.ve, stats=st.vest, dout=dxv, dxd16=st.de_op, dgamma=G(vlg), dbeta=G(vlb),
                dbias=G(vb), dpos=G(vpos), pos_period=F, p_post=p, seed=self._seed, off_post=off_v, seed_dev=self._seed_dev), SV)
            bwd.add("univl_gemm", _gemm_desc(dt, st.de_op, H, st.vn_op, D, H, D, Tv, trans_a=1, trans_b=1, out32=G(vw),
                                             ldc=D, accumulate=not fresh), SV)
            bwd.add("univl_gemm", _gemm_desc(dt, st.de_op, H, fl.wop(vw), D, Tv, D, H, trans_b=1, out32=st.dvnorm, ldc=D), SV)
            bwd.add("univl_layernorm_bwd", ops.layernorm_desc(
                dt, Tv, D, gamma=W32(nv_g), y=st.vy, stats=st.vst, dout=st.dvnorm, dgamma=G(nv_g), dbeta=G(nv_b)), SV)
            dxt = st.text.build_backward(bwd, st.dseq, st.t0_32, st.t0_16, fresh, training, layer_hook=hook)
            bwd.add("univl_embed_text_bwd", ops.embed_text_desc(
                dt, B, W, st.ids, W32(bw), W32(bp), W32(blg), W32(blb), type_ids=st.type_ids, type_emb=W32(bt), y=st.te,
                stats=st.test, p_post=p, seed=self._seed, off_post=off_t, seed_dev=self._seed_dev, dout=dxt,
                dword=G(bw), dpos=G(bp), dtype_emb=G(bt), dgamma=G(blg), dbeta=G(blb)), ST)
            bwd.join(SV, ST)
            if red is not None:
                for (s0, e0) in buckets["tail"]:
                    bwd.add_callable(lambda s0=s0, e0=e0: red.reduce_slice(s0, e0))
                bwd.add_callable(red.join)
            return bwd

        st.bwd_fresh = build_bwd(True)
        st.bwd_acc = None
        st._build_bwd = build_bwd
        return st

    # ---------------------------------------------------------------------------------------- execution
    def _load_inputs(self, st, input_ids, token_type_ids, attention_mask, video, video_mask):
        st.ids.copy_(input_ids.reshape(st.B, st.W), non_blocking=True)
        st.type_ids.copy_(token_type_ids.reshape(st.B, st.W), non_blocking=True)
        st.amask.copy_(attention_mask.reshape(st.B, st.W), non_blocking=True)
        st.video.copy_(torch.as_tensor(video).reshape(st.B * st.F, -1), non_blocking=True)
        st.vmask.copy_(video_mask.reshape(st.B, st.F), non_blocking=True)

    def _run_backward(self, st, gout):
        fl = self.flat
        used = self.used_parameter_names()
        fresh = all(fl.params[n].grad is None for n in (used[0], used[-1]))
        if gout is None:
            st.gout.fill_(1.0)
        else:
            st.gout.copy_(gout.reshape(1).to(torch.float32))
        fl.grad_version += 1
        if fresh:
            st.bwd_fresh.run()
        else:
            if st.bwd_acc is None:
                st.bwd_acc = st._build_bwd(False)
            st.bwd_acc.run()
        fl.attach_grads(used)

    def forward(self, input_ids, token_type_ids, attention_mask, video, video_mask=None,
                pairs_masked_text=None, pairs_token_labels=None, masked_video=None, video_labels_index=None,
                input_caption_ids=None, decoder_mask=None, output_caption_ids=None):
        """modeling.py:188-271 (stage-one branch).  Returns the scalar loss in training mode, None otherwise."""
        W, F = input_ids.shape[-1], video_mask.shape[-1]
        B = input_ids.numel() // W
        fl = self.flat
        fl.refresh_shadow()
        st = self._get_step(B, W, F)
        self._load_inputs(st, input_ids, token_type_ids, attention_mask, video, video_mask)
        if not self.training:
            return None
        anchor = fl.params["normalize_video.visual_norm2d.bias"]
        if torch.is_grad_enabled() and anchor.requires_grad:
            out = _JointLossFn.apply(anchor, self, st).as_subclass(_LossTensor)
            out._univl = (self, st)
            return out
        st.fwd.run()
        return st.loss[0].clone()

    def get_sequence_visual_output(self, input_ids, token_type_ids, attention_mask, video, video_mask, shaped=False):
        """modeling.py:299-313.  `shaped=True` means the caller already flattened the pair dim AND normalised the video
        (only UniVL.forward does that in the reference); external callers use shaped=False."""
        if shaped:
            raise NotImplementedError("shaped=True is internal to the reference's forward(); pass raw inputs")
        W, F = input_ids.shape[-1], video_mask.shape[-1]
        B = input_ids.numel() // W
        fl = self.flat
        fl.refresh_shadow()
        st = self._get_step(B, W, F)
        self._load_inputs(st, input_ids, token_type_ids, attention_mask, video, video_mask)
        st.fwd.run(upto=st.fwd_encoders_len)
        return st.seq_out.view(B, W, -1).clone(), st.vis_out.view(B, F, -1).clone()

    def get_similarity_logits(self, sequence_output, visual_output, attention_mask, video_mask, shaped=False,
                              _pretrain_joint=False):
        """modeling.py:377-391 (mean-pooling branch): masked means, L2 normalisation unless use_mil, text . video^T."""
        attention_mask = attention_mask.reshape(-1, attention_mask.shape[-1])
        video_mask = video_mask.reshape(-1, video_mask.shape[-1])
        _lib.lib()
        dev = sequence_output.device
        if dev.type != "cuda":
            raise RuntimeError("univl_amd.UniVL.get_similarity_logits needs HIP device tensors; no CPU fallback")
        Bt, W, H = sequence_output.shape
        Bv, F, _ = visual_output.shape
        norm = not bool(self.task_config.use_mil)
        seq = sequence_output.to(torch.float32).contiguous()
        vis = visual_output.to(torch.float32).contiguous()
        am = attention_mask.to(dev, torch.int64).contiguous()
        vm = video_mask.to(dev, torch.int64).contiguous()
        ldt, ldv = (Bt + 3) // 4 * 4, (Bv + 3) // 4 * 4
        tn = torch.zeros(ldt, H, device=dev)
        vn = torch.zeros(ldv, H, device=dev)
        ops.pool_fwd(Bt, W, seq, am, skip_first=True, normalize=norm, out=tn)
        ops.pool_fwd(Bv, F, vis, vm, skip_first=False, normalize=norm, out=vn)
        sim = torch.empty(Bt, Bv, device=dev)
        ops.gemm(tn, vn, Bt, Bv, H, out32=sim)
        return sim

    def decoder_caption(self, *a, **kw):
        raise NotImplementedError("decoder path (SURVEY.md section 8a, caption rows) is not built yet in univl_amd")
