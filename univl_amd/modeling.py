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
import weakref

import torch
from torch import nn

from . import _ab, _lib, ops
from .engine import FlatParams, Plan
from .parallel import BucketReducer, broadcast_parameters

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


class _CrossEmbeddingsParams(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)
        self.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, cfg.hidden_size)
        self.LayerNorm = LayerNorm(cfg.hidden_size)


class CrossModel(nn.Module):
    """Parameters of module_cross.py:356-362."""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.embeddings = _CrossEmbeddingsParams(cfg)
        self.encoder = _EncoderParams(cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers)
        self.pooler = _PoolerParams(cfg.hidden_size)


class _MHAParams(nn.Module):
    def __init__(self, Hd):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(Hd, Hd), nn.Linear(Hd, Hd), nn.Linear(Hd, Hd)


class _DecoderAttentionParams(nn.Module):
    def __init__(self, Hd):
        super().__init__()
        self.att = _MHAParams(Hd)
        self.output = _SelfOutputParams(Hd, Hd)


class _DecoderLayerParams(nn.Module):
    """module_decoder.py:279-285."""

    def __init__(self, Hd, I):
        super().__init__()
        self.slf_attn = _DecoderAttentionParams(Hd)
        self.enc_attn = _DecoderAttentionParams(Hd)
        self.intermediate = _IntermediateParams(Hd, I)
        self.output = _SelfOutputParams(I, Hd)


class _DecoderEmbeddingsParams(nn.Module):
    """module_decoder.py:294-307: word / position tables are BERT's (tied)."""

    def __init__(self, cfg, word_w, pos_w):
        super().__init__()
        self.word_embeddings = nn.Embedding(word_w.shape[0], word_w.shape[1])
        self.position_embeddings = nn.Embedding(pos_w.shape[0], pos_w.shape[1])
        self.word_embeddings.weight = word_w
        self.position_embeddings.weight = pos_w
        self.LayerNorm = LayerNorm(cfg.hidden_size)


class _HeadTransformParams(nn.Module):
    def __init__(self, Hd):
        super().__init__()
        self.dense = nn.Linear(Hd, Hd)
        self.LayerNorm = LayerNorm(Hd)


class _LMPredictionHeadParams(nn.Module):
    """module_bert.py:314-325 / module_decoder.py:170-178: tied decoder.weight + own bias."""

    def __init__(self, Hd, emb_w):
        super().__init__()
        self.transform = _HeadTransformParams(Hd)
        self.decoder = nn.Linear(emb_w.size(1), emb_w.size(0), bias=False)
        self.decoder.weight = emb_w
        self.bias = nn.Parameter(torch.zeros(emb_w.size(0)))


class _OnlyMLMHeadParams(nn.Module):
    def __init__(self, Hd, emb_w):
        super().__init__()
        self.predictions = _LMPredictionHeadParams(Hd, emb_w)


class _VisualLMPredictionHeadParams(nn.Module):
    """module_visual.py:298-306: weight tied to the visual input projection (768,1024), bias (1024)."""

    def __init__(self, Hd, vis_w):
        super().__init__()
        self.transform = _HeadTransformParams(Hd)
        self.weight = vis_w
        self.bias = nn.Parameter(torch.zeros(vis_w.size(1)))


class _VisualOnlyMLMHeadParams(nn.Module):
    def __init__(self, Hd, vis_w):
        super().__init__()
        self.predictions = _VisualLMPredictionHeadParams(Hd, vis_w)


class _DecoderClassifierParams(nn.Module):
    def __init__(self, Hd, emb_w):
        super().__init__()
        self.cls = _OnlyMLMHeadParams(Hd, emb_w)


class _DecoderParams(nn.Module):
    def __init__(self, Hd, I, L):
        super().__init__()
        self.layer = nn.ModuleList([_DecoderLayerParams(Hd, I) for _ in range(L)])


class DecoderModel(nn.Module):
    """Parameters of module_decoder.py:351-370."""

    def __init__(self, cfg, word_w, pos_w):
        super().__init__()
        self.config = cfg
        self.embeddings = _DecoderEmbeddingsParams(cfg, word_w, pos_w)
        self.decoder = _DecoderParams(cfg.hidden_size, cfg.intermediate_size, cfg.num_decoder_layers)
        self.classifier = _DecoderClassifierParams(cfg.hidden_size, word_w)


def _loss_out(step):
    """A fresh 0-d tensor holding the step's loss (the step's own buffer is rewritten by the next forward).  A kernel node,
    not clone(): a device-to-device memcpy is a ~10 us node of the captured step."""
    out = torch.empty((), device=step.loss.device, dtype=step.loss.dtype)
    if out.is_cuda and _ab.get("copy_kernel"):
        from . import ops
        ops.copy_many([(out, step.loss)])
    else:
        out.copy_(step.loss[0])
    return out


class _StepLossFn(torch.autograd.Function):
    """loss = UniVL.forward(...): the whole forward is one static plan, the whole backward another; parameter gradients
    are written straight into the flat gradient buffer (p.grad are views of it)."""

    @staticmethod
    def forward(ctx, anchor, model, step):
        model._run_plan(step.fwd, step)
        ctx.model, ctx.step, ctx.anchor = model, step, anchor
        return _loss_out(step)

    @staticmethod
    def backward(ctx, gout):
        ctx.model._run_backward(ctx.step, gout)
        # Under a stock torch DistributedDataParallel wrapper the anchor is the one parameter DDP still tracks
        # (UniVL._ddp_params_and_buffers_to_ignore lists all others): its autograd hook must fire every iteration.
        ganchor = torch.zeros_like(ctx.anchor) if ctx.model._implicit_dp else None
        return ganchor, None, None


class _LossTensor(torch.Tensor):
    """The scalar returned by UniVL.forward.  It is an ordinary autograd tensor (grad_fn = _StepLossFn), but a plain
    `loss.backward()` -- what main_task_retrieval.py:342 does -- runs the backward plan directly in the calling
    thread instead of going through the autograd engine's device thread (less overhead, and a multi-stream plan can
    then be captured into a hipGraph together with the forward).  Anything else (`(loss / k).backward()`,
    `loss.mean()`, explicit gradients) takes the normal autograd route and ends in _StepLossFn.backward."""

    def __float__(self):
        # `total_loss += float(loss)` (main_task_retrieval.py:344) on a tensor that requires grad: read the detached value
        # (same number, without torch's "converting a tensor with requires_grad=True to a scalar" warning every step)
        a = self.__dict__.get("_async")
        if a is not None:                  # univl_amd.graphed: the value was copied to pinned memory right after the forward;
            a[1].synchronize()             # waiting for THAT copy does not wait for the backward / optimizer behind it
            return float(a[0][0])
        return torch.Tensor.__float__(self.detach())

    def item(self):
        return float(self)

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        ms = self.__dict__.get("_univl", None)
        if ms is not None and gradient is None and inputs is None and not create_graph and not ms[0]._implicit_dp:
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

        self.graph_backward = False
        # operational switches a training process may turn off on the model (UNIVL_AB is refused outside measurement harnesses)
        self.auto_graph = bool(_ab.get("auto_graph"))      # the unchanged loop switches to graph replay on its own
        self.auto_dp = bool(_ab.get("auto_dp"))            # built-in gradient exchange inside an initialised process group
        self.dp_capture = bool(_ab.get("dp_capture"))      # RCCL exchange captured into the step graph (False: host-issued collectives)
        self.auto_ride = _ab.get("adam_ride") != "0"       # optimizer.step() leaves its launch to the next forward (bf16, one process)
        self._operand_pairs = str(_ab.get("pairs"))        # see the `operand_pairs` property
        self._stage_one, self._stage_two = True, False
        if _check_attr("stage_two", tc):
            self._stage_one, self._stage_two = False, tc.stage_two
        self.train_sim_after_cross = bool(self._stage_one and _check_attr("train_sim_after_cross", tc))

        # modeling.py:133-169 -- same construction order, same conditions
        if hasattr(tc, "text_num_hidden_layers"):
            bert_config.num_hidden_layers = tc.text_num_hidden_layers
        self.bert = BertModel(bert_config)
        word_w = self.bert.embeddings.word_embeddings.weight
        pos_w = self.bert.embeddings.position_embeddings.weight
        if hasattr(tc, "visual_num_hidden_layers"):
            visual_config.num_hidden_layers = tc.visual_num_hidden_layers
        self.visual = VisualModel(visual_config)
        vis_w = self.visual.embeddings.word_embeddings.weight
        self.cross, self.decoder = None, None
        if self._stage_one is False or self.train_sim_after_cross:
            if hasattr(tc, "cross_num_hidden_layers"):
                cross_config.num_hidden_layers = tc.cross_num_hidden_layers
            self.cross = CrossModel(cross_config)
            if self.train_sim_after_cross is False:
                if hasattr(tc, "decoder_num_hidden_layers"):
                    decoder_config.num_decoder_layers = tc.decoder_num_hidden_layers
                self.decoder = DecoderModel(decoder_config, word_w, pos_w)
            if tc.do_pretrain:
                self.cls = _OnlyMLMHeadParams(bert_config.hidden_size, word_w)
                self.cls_visual = _VisualOnlyMLMHeadParams(visual_config.hidden_size, vis_w)
            self.similarity_dense = nn.Linear(bert_config.hidden_size, 1)
        self.normalize_video = NormalizeVideo(tc)
        self.apply(self.init_weights)

        self._flat = None
        self._steps = {}
        self._param_events = {}        # filled by a pipelined optimizer update in flight (univl_amd.graphed)
        self._pending_update = None    # the optimizer holding a deferred update, if any
        self._rider_update = None      # riding optimizer update (graphed.GraphedTrainStep): dict(desc, groups, max_blocks) of a prepared BertAdam update
                                       # that THIS forward applies -- prologue launches + riders of its forward products
        self._in_pipelined_call = False
        self._reducer = None
        self._dp_checked, self._implicit_dp = False, False
        self._seed_dev = None
        self._seed = int(getattr(tc, "seed", 42))
        dt = getattr(tc, "compute_dtype", None) or os.environ.get("UNIVL_COMPUTE_DTYPE", "bf16")
        self.compute_dtype = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32,
                              "float32": torch.float32}[str(dt).replace("torch.", "")]
        self.dropout_prob = float(getattr(tc, "dropout_prob", bert_config.hidden_dropout_prob))
        # A stock torch DistributedDataParallel wrapper (main_task_retrieval.py:197-198) must not reduce these
        # gradients a second time: they never pass through autograd's accumulators (the backward plan writes the flat
        # gradient buffer directly and exchanges it itself, see _auto_data_parallel).  DDP reads this list.
        self._ddp_params_and_buffers_to_ignore = self._ddp_ignore_list()

    ANCHOR = "normalize_video.visual_norm2d.bias"

    def _ddp_ignore_list(self):
        names = []
        for mn, m in self.named_modules():
            for pn, _ in list(m.named_parameters(recurse=False)) + list(m.named_buffers(recurse=False)):
                fqn = "%s.%s" % (mn, pn) if mn else pn
                if fqn != self.ANCHOR:
                    names.append(fqn)
        return names

    def _auto_data_parallel(self):
        """First training forward inside an initialised process group with no explicit enable_data_parallel(): the model
        is presumably wrapped the way the reference's scripts wrap it (stock DDP, find_unused_parameters=True).  Turn
        the built-in gradient exchange on, broadcast rank 0's parameters like DDP's constructor does, and route
        loss.backward() through autograd so that DDP's bookkeeping for the anchor parameter stays consistent."""
        self._dp_checked = True
        import torch.distributed as dist
        if not self.auto_dp or not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size() > 1:
            self.enable_data_parallel()
            self._implicit_dp = True

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
        if self.__dict__.get("_pending_update") is not None:
            self._flush_pending()           # .to() / .float() re-allocate the parameters a pending update still points at
        r = super()._apply(fn, *a, **kw)
        self._flat, self._steps = None, {}      # parameters were re-allocated (e.g. .to(device)): re-flatten lazily
        return r

    def _flush_pending(self):
        """A training step of univl_amd.graphed.GraphedTrainStep(pipeline_optimizer=True) leaves its BertAdam update to be
        applied next to the following forward pass; everything else that reads the parameters applies it first."""
        o = self._pending_update
        if o is not None and not self._in_pipelined_call:
            o.flush()

    def named_parameters(self, *a, **kw):
        """(parameters() goes through here.)  A BertAdam update that optimizer.step() left for the next forward (the unchanged training
        loop, _adopt_pending_update) is applied before anybody is handed the parameters through the module API; the training loop
        itself asks for them only between backward and step (clip_grad_norm_(model.parameters(), ...), main_task_retrieval.py:347),
        where nothing is pending."""
        o = self.__dict__.get("_pending_update")
        if o is not None and getattr(o, "_auto_deferred", False):
            self._flush_pending()
        return super().named_parameters(*a, **kw)

    def state_dict(self, *a, **kw):
        self._flush_pending()
        fl = self._flat
        if fl is not None and getattr(fl, "shard_reducer", None) is not None and not fl.master_complete:
            raise RuntimeError("UniVL.state_dict(): the fp32 master weights are sharded over the ranks (sharded optimizer) -- "
                               "call model.consolidate_parameters() on EVERY rank first (a collective)")
        return super().state_dict(*a, **kw)

    def train(self, mode=True):
        if not mode:
            self._flush_pending()
        return super().train(mode)

    def mark_params_dirty(self):
        """Tell the model its fp32 parameters were modified outside univl_amd.optimization.BertAdam (the bf16
        shadow the GEMMs read is refreshed before the next forward)."""
        self._flush_pending()               # nothing stays pending across an outside edit (the edit itself should come after a flush:
                                            # INTEGRATION.md, "what a deferred update can and cannot hide")
        if self._flat is not None:
            self._flat.shadow_valid = False

    def load_state_dict(self, *a, **kw):
        self._flush_pending()               #  never on top of the loaded weights
        r = super().load_state_dict(*a, **kw)
        self.mark_params_dirty()
        return r

    def _replicate_for_data_parallel(self):
        self._flush_pending()
        replica = super()._replicate_for_data_parallel()
        replica._flat, replica._steps, replica._reducer, replica._seed_dev = None, {}, None, None
        replica._param_events, replica._pending_update, replica._in_pipelined_call = {}, None, False
        replica._used_names = {}
        replica._dp_checked, replica._implicit_dp, replica.graph_backward = True, False, False
        return replica

    @property
    def flat(self):
        if self._flat is None:
            named = list(self.named_parameters())
            if not named and getattr(self, "_is_replica", False):
                # nn.parallel.replicate (util.py:22, the eval path of main_task_retrieval.py:402-436): a replica's
                # weights are plain tensors kept in _former_parameters; it gets its own flat storage on its device
                seen = set()
                for mn, m in self.named_modules():
                    for k, t in getattr(m, "_former_parameters", {}).items():
                        if t is not None and id(t) not in seen:
                            seen.add(id(t))
                            named.append((("%s.%s" % (mn, k)) if mn else k, t))
            p0 = named[0][1]
            if not p0.is_cuda:
                raise RuntimeError("univl_amd.UniVL runs only on a HIP device (call model.to('cuda')); no CPU fallback")
            _lib.lib()
            if os.environ.get("UNIVL_DETERMINISTIC", "0") == "1" and not _lib.deterministic():
                with torch.cuda.device(p0.device):         # fixed-order reductions (bit-reproducible runs): include/univl_hip.h
                    _lib.set_deterministic(True)
            self._flat = FlatParams(named, p0.device, self.compute_dtype)
            self._flat.owner = weakref.ref(self)
            # forward plans are built with rider slots (engine.Plan.add_gemm_rider: a plain product unless an update rides) wherever a
            # BertAdam update CAN ride with the next forward -- the captured step of graphed.GraphedTrainStep and, round 5, the unchanged
            # training loop (optimization.BertAdam.step defers itself, UniVL.forward applies it: _adopt_pending_update)
            self._flat.adam_ride = self.compute_dtype == torch.bfloat16 and _ab.get("adam_ride") != "0"
            self._flat.operand_pairs = self._operand_pairs
            self._seed_dev = torch.zeros(1, device=p0.device, dtype=torch.int64)
            self._steps = {}
        return self._flat

    @property
    def operand_pairs(self):
        """Precision of the bf16 step's forward products in stacks of at most 768 tokens (16 pairs x 48): '' (default) plain bf16 MFMA
        operands; 'x' / 'w' / 'xw': the activation / weight / both operands as PAIRS of bf16 (hi + lo = 16 mantissa bits,
        include/univl_hip.h: UnivlGemm.A_lo / B_lo).  'xw' brings the median per-tensor gradient error against the reference's fp32
        gradients from 9e-3 to 5.6e-3 and the pretrain configuration's global error from 1.2e-2 to 7.3e-3, at +22 % step time at
        4 pairs per GPU (DESIGN.md section 2 has the table) -- which is why it is a choice and not the default.  Setting it rebuilds
        the plans on the next forward."""
        return self._operand_pairs

    @operand_pairs.setter
    def operand_pairs(self, value):
        value = "" if value is None else str(value)
        if value not in ("", "x", "w", "xw"):
            raise ValueError("operand_pairs: '', 'x', 'w' or 'xw' (got %r)" % (value,))
        if value != self._operand_pairs:
            self._flush_pending()
            self._operand_pairs = value
            if self._flat is not None:
                self._flat.operand_pairs = value
                self._steps = {}

    def enable_data_parallel(self, process_group=None, broadcast=True, loopback=False, force=False, shard_optimizer=None):
        """Re-homes the reference's DDP wrap (main_task_retrieval.py:197-198) onto per-layer RCCL all-reduces of
        the flat gradient buffer, overlapped with backward (univl_amd.parallel).  Call after model.to(device) and
        torch.distributed.init_process_group; with world_size 1 it is a no-op."""
        fl = self.flat
        self._dp_checked, self._implicit_dp = True, False
        if broadcast:
            broadcast_parameters(fl.p32, 0, process_group)
            fl.shadow_valid = False
        self._reducer = BucketReducer(fl.g32, process_group, loopback=loopback, force=force)
        if not self._reducer.active:
            self._reducer = None
        elif self.dp_capture and not loopback and fl.device.type == "cuda":
            # RCCL: a communicator of our own, so that the exchange is captured into the step's hipGraph (univl_amd.rccl)
            try:
                with torch.cuda.device(fl.device):
                    self._reducer.enable_capture()
            except Exception as ex:      # noqa: BLE001 -- the process-group path below does the same exchange from the host
                import sys
                print("[univl_amd] RCCL communicator for captured gradient exchange unavailable (%s: %s); using torch.distributed "
                      "collectives between captured segments" % (type(ex).__name__, ex), file=sys.stderr)
        fl.owned, fl.shard_reducer, fl.master_complete = None, None, True
        fl.partition_version = getattr(fl, "partition_version", 0) + 1      # chunk-table caches of the optimizer are keyed on it
        if shard_optimizer is None:
            shard_optimizer = os.environ.get("UNIVL_SHARD_OPT", "0") == "1"
        if shard_optimizer and self._reducer is not None and not loopback:
            # reduce-scatter of the gradients, clip + BertAdam on this rank's 1/world of every bucket, all-gather of the
            # bf16 shadow (univl_amd.parallel).  The fp32 master and the moments of the other ranks' pieces go stale on this
            # rank: consolidate_parameters() / optimizer.consolidate() (collectives) bring them together for a checkpoint.
            from .parallel import shard_partition
            self._reducer.set_partition(shard_partition(fl, self._reducer.world))
            fl.owned, fl.shard_reducer = self._reducer.owned, self._reducer
        self._steps = {}
        return self

    def consolidate_parameters(self):
        """Sharded optimizer: COLLECTIVE (every rank calls it) -- gathers every rank's pieces of the fp32 master weights, after
        which state_dict() is complete on every rank."""
        fl = self.flat
        red = getattr(fl, "shard_reducer", None)
        if red is not None and not fl.master_complete:
            self._flush_pending()
            red.join()
            red.all_gather_ranges(fl.p32)
            red.join()
            fl.master_complete = True
        return self

    def step_kind(self, has_caption):
        """Which of the reference's forward() branches applies (modeling.py:204-267)."""
        tc = self.task_config
        if self._stage_one:
            return "align" if self.train_sim_after_cross else "joint"
        if tc.do_pretrain:
            return "pretrain" if has_caption else "pretrain_nocap"     # modeling.py:238: the decoder loss needs captions
        if tc.task_type == "caption":
            return "caption"
        return "align"                       # stage two, task_type "retrieval": cross-encoder similarity + CrossEn

    def used_parameter_names(self, kind=None):
        """Parameters that receive a gradient in this configuration (the reference leaves `.grad = None` on the rest:
        dead poolers modeling.py:307,310; the cross pooler / similarity_dense on the caption path)."""
        kind = kind or self.step_kind(True)
        cached = self.__dict__.setdefault("_used_names", {}).get(kind)
        if cached is not None:
            return cached
        out = self._used_names[kind] = []
        for n, _ in self.named_parameters():
            if n.startswith("bert.pooler") or n.startswith("visual.pooler"):
                continue
            if kind == "caption" and (n.startswith("cross.pooler") or n.startswith("similarity_dense")):
                continue
            if kind == "align" and (n.startswith("decoder.") or n.startswith("cls")):
                continue
            if kind == "pretrain_nocap" and n.startswith("decoder."):
                continue
            out.append(n)
        return out

    # ---------------------------------------------------------------------------------------- execution
    def _get_step(self, kind, B, W, F):
        key = (kind, B, W, F, self.training)
        st = self._steps.get(key)
        if st is None:
            from .steps import build_step
            st = build_step(self, kind, B, W, F, self.training)
            self._steps[key] = st
        return st

    AUTO_GRAPH_AFTER = 2
    GRAPH_SIGS = 4

    def _run_plan(self, plan, st):
        """Enqueue a forward / backward plan.  Launched kernel by kernel from Python the host is the bottleneck at small
        batch (~10 us per launch against 2-10 us kernels), so once a step signature has been run AUTO_GRAPH_AFTER times
        eagerly its plans are captured into hipGraphs (engine.Plan.run_graphed: one graph, or captured segments around
        host-issued collectives) and replayed from then on -- the unchanged training loop of main_task_retrieval.py:333-352
        then runs close to the fully captured step of graphed.GraphedTrainStep.  UNIVL_AUTO_GRAPH=0 turns it off; inside
        somebody else's capture the plan is always enqueued directly."""
        hot = self.auto_graph and st.calls > self.AUTO_GRAPH_AFTER
        if (self.graph_backward or hot) and not torch.cuda.is_current_stream_capturing():
            if plan.rider_keys:
                # a captured plan holds the riding update's descriptor and chunk ranges: replay only while they are the ones captured
                # (steady state of a training loop: same optimizer tables, same buffers), else capture again
                rd = plan.riders
                sig = None if rd is None else (bytes(rd["desc"]), tuple(sorted(rd["ranges"].items())), int(rd.get("max_blocks", 0)))
                if getattr(plan, "_rider_sig", None) != sig:
                    # captured segments are kept PER signature: under gradient accumulation the forward after step() carries
                    # riders and the following ones do not -- both graphs are captured once and replayed, not re-captured at every
                    # transition.  At most GRAPH_SIGS signatures are kept (steady state: "no riders" + one update descriptor).
                    cache = plan.__dict__.setdefault("_seg_cache", {})
                    if plan._segments is not None:
                        cache[getattr(plan, "_rider_sig", None)] = plan._segments
                        while len(cache) > self.GRAPH_SIGS:
                            cache.pop(next(iter(cache)))
                    plan._segments = cache.pop(sig, None)
                    if plan._segments is None:
                        self.graph_captures = getattr(self, "graph_captures", 0) + 1
                if plan._segments is None and sig is not None:
                    import ctypes as C                  # the rider kernels' large-LDS opt-in must not happen inside the capture
                    _lib.check(_lib.lib().univl_gemm_rider_prime(C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm_rider_prime")
                plan._rider_sig = sig
            plan.run_graphed()
        else:
            plan.run()

    def _run_backward(self, st, gout):
        fl = self.flat
        used = self.used_parameter_names(st.kind)
        fresh = all(fl.params[n].grad is None for n in (used[0], used[-1]))
        if gout is not None:                 # st.gout holds 1.0 between backwards (steps.Step): plain loss.backward() writes nothing
            st.gout.copy_(gout.reshape(1).to(torch.float32))
        if fl._pending is not None:          # a deferred clip nobody consumed: it scales the OLD gradients, as torch's did
            from .optimization import apply_pending_clip
            apply_pending_clip(fl)
        if fl.g32._version != getattr(fl, "_g32_tv", fl.g32._version):
            fl.mark_all_word_rows()          # the gradients were edited through torch since the last backward: rows unknown
            if getattr(fl, "_word_rows", None) is not None:
                fl._word_rows[1][1] = 1
        fl.grad_version += 1
        self._run_plan(st.backward_plan(fresh), st)
        plan = st.backward_plan(fresh)
        fl.fused = (dict(version=fl.grad_version, names=plan.fused_names, tv=fl.g32._version)
                    if plan.fused_names else None)
        fl._g32_tv = fl.g32._version
        if getattr(plan, "rows_mode", False):
            fl.word_rows_version = fl.grad_version            # the row list describes exactly these gradients
        else:
            fl.mark_all_word_rows()                            # a writer that lists no rows (tied heads, token gather above the cap)
            if getattr(fl, "_word_rows", None) is not None:
                fl._word_rows[1][1] = 1                        # a dense writer ran: every row counts as listed from now on
        fl.attach_grads(used)
        if gout is not None:
            st.gout.fill_(1.0)

    def forward(self, input_ids, token_type_ids, attention_mask, video, video_mask=None,
                pairs_masked_text=None, pairs_token_labels=None, masked_video=None, video_labels_index=None,
                input_caption_ids=None, decoder_mask=None, output_caption_ids=None):
        """modeling.py:188-271.  Returns the scalar loss in training mode, None otherwise."""
        if not self.training:
            return None
        W, F = input_ids.shape[-1], video_mask.shape[-1]
        B = input_ids.numel() // W
        kind = self.step_kind(input_caption_ids is not None)
        if kind == "caption" and input_caption_ids is None:
            # modeling.py:238-254: without captions the caption task has no loss term at all (the reference returns the
            # float 0.0, whose .backward() fails in the script)
            raise RuntimeError("UniVL.forward: the caption path needs input_caption_ids / decoder_mask / output_caption_ids")
        if not self._dp_checked:
            self._auto_data_parallel()
        adopted, riding = False, False
        try:
            adopted = self._adopt_pending_update()
            if not adopted:
                self._flush_pending()
            fl = self.flat
            if getattr(fl, "shard_reducer", None) is not None:
                fl.shard_reducer.join()            # the all-gather of the updated shadow must have landed
            fl.refresh_shadow()
            st = self._get_step(kind, B, W, F)
            st.enc.load(input_ids, token_type_ids, attention_mask, video, video_mask)
            if kind in ("pretrain", "pretrain_nocap"):
                st.enc_m.load(pairs_masked_text, token_type_ids, attention_mask, masked_video, video_mask)
                st.heads.load(pairs_token_labels, video_labels_index)
            if st.decoder is not None:
                st.decoder.load(input_caption_ids, decoder_mask, output_caption_ids)
            anchor = fl.params[self.ANCHOR]
            st.calls += 1
            ru = self._rider_update
            if ru is not None:
                riding = True
                self._start_riding_update(st, ru)
            try:
                if torch.is_grad_enabled() and anchor.requires_grad:
                    out = _StepLossFn.apply(anchor, self, st).as_subclass(_LossTensor)
                    out._univl = (self, st)
                    return out
                self._run_plan(st.fwd, st)
                return _loss_out(st)
            finally:
                st.fwd.riders = None
        except BaseException:
            if adopted:
                self._unadopt_pending_update(riding)
            raise
        finally:
            # an update this forward adopted is never left behind: either it went out with the plan, or the except branch
            # above handed it back to the optimizer; graphed.GraphedTrainStep sets and clears its own descriptor around its calls
            if adopted:
                self._rider_update = None

    def _unadopt_pending_update(self, riding):
        """forward() adopted the optimizer's pending update and failed.  Before any launch of the update went out (bad batch shape, out of
        memory while a new step was built, a failing join): hand it back -- it is pending again and the next forward (or any flush) applies
        it exactly once.  After _start_riding_update began enqueueing it cannot be taken back (some chunk ranges are applied, the ones
        riding in the failed plan are not): apply the whole set of ranges that have not gone out as plain launches, so that every
        parameter sees the update exactly once, and leave nothing pending."""
        o, ru = self._pending_update, self._rider_update
        self._rider_update = None
        if o is None or ru is None:
            return
        if not riding:
            o._deferred, o._auto_deferred = True, True
            self.auto_ride_count = getattr(self, "auto_ride_count", 1) - 1
            return
        import ctypes as C
        for st in self._steps.values():
            rd = getattr(getattr(st, "fwd", None), "riders", None)
            if rd is None or rd.get("desc") is not ru["desc"]:
                continue
            h = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            done = {}
            for key, slot in rd["used"]:
                done.setdefault(key, set()).add(slot)
            for key, (c0, n) in rd["ranges"].items():
                if key in done:
                    # a stack whose first products already carried a part: the slots are fractions of the range only the plan knows;
                    # nothing here can tell which chunks went out -- refuse to guess
                    raise RuntimeError("univl_amd.UniVL.forward failed after a part of the riding BertAdam update was enqueued; the "
                                       "parameters hold a partial update (model.auto_ride = False avoids riding updates)")
                _lib.check(_lib.lib().univl_bert_adam_range(C.byref(rd["desc"]), c0, n, 0, 0, h), "bert_adam_range")
            st.fwd.riders = None

    def _adopt_pending_update(self):
        """The unchanged training loop (main_task_retrieval.py:333-353): optimizer.step() left its BertAdam update pending
        (optimization.BertAdam.step: auto-deferred), and THIS forward applies it -- the chunk groups the plan cannot carry as launches in
        front of it, the rest as extra workgroups of its own products (_start_riding_update) -- exactly what the captured step of
        graphed.GraphedTrainStep does.  Anything else that reads the parameters flushes the update first (_flush_pending)."""
        o = self._pending_update
        if (o is None or self._in_pipelined_call or self._rider_update is not None or not getattr(o, "has_pending", False)
                or not getattr(o, "_auto_deferred", False)):
            return False
        fl = self.flat
        if not getattr(fl, "adam_ride", False) or getattr(fl, "shard_reducer", None) is not None or torch.cuda.is_current_stream_capturing():
            return False
        self._rider_update = dict(desc=o._last_desc, groups=o.chunk_groups(), max_blocks=0)
        o._deferred = False                   # from here on the update counts as applied (launch_deferred's bookkeeping)
        o._auto_deferred = False
        # fl.shadow_valid is left as it is: the update rewrites the bf16 shadow of the tensors it updates, and a shadow that was marked
        # dirty (mark_params_dirty, a broadcast) is still refreshed from the fp32 master by this forward, in front of the riders
        self.auto_ride_count = getattr(self, "auto_ride_count", 0) + 1
        return True

    def _start_riding_update(self, st, ru):
        """Riding optimizer update (univl_amd.graphed, UNIVL_ADAM_RIDE): the BertAdam update of the previous iteration is applied BY this
        forward -- chunk groups the forward plan cannot carry (embedding tables, vectors, the first layer of each stack, the
        cross encoder / decoder) as ordinary launches on the calling stream right now, the others as extra workgroups of the
        forward products of the layer before (engine.Plan.add_gemm_rider)."""
        import ctypes as C
        from . import _lib
        L, d = _lib.lib(), ru["desc"]
        h = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ranges, first = {}, True
        for key, c0, n in ru["groups"]:
            if key in st.fwd.rider_keys and key not in ranges:
                ranges[key] = (c0, n)
                continue
            _lib.check(L.univl_bert_adam_range(C.byref(d), c0, n, 1 if first else 0, 0, h), "bert_adam_range")
            first = False
        if first:                     # nothing went out as a prologue launch: the per-tensor scalars still have to be prepared
            _lib.check(L.univl_bert_adam_range(C.byref(d), 0, 0, 1, 0, h), "bert_adam_range")
        st.fwd.riders = dict(desc=d, ranges=ranges, max_blocks=int(ru.get("max_blocks", 0)), used=set())

    def get_sequence_visual_output(self, input_ids, token_type_ids, attention_mask, video, video_mask, shaped=False):
        """modeling.py:299-313.  `shaped=True` means the caller already flattened the pair dim AND normalised the video
        (only UniVL.forward does that in the reference); external callers use shaped=False."""
        W, F = input_ids.shape[-1], video_mask.shape[-1]
        B = input_ids.numel() // W
        self._flush_pending()
        fl = self.flat
        fl.refresh_shadow()
        was = self.training
        self.training = False                      # feature extraction never applies dropout plans' training variant
        try:
            # shaped=True (modeling.py:300-305 skipped): `video` is already float32 [B, F, video_dim] and normalised
            st = self._get_step("features_shaped" if shaped else ("joint" if self.cross is None else "features"), B, W, F)
        finally:
            self.training = was
        st.enc.load(input_ids, token_type_ids, attention_mask, video, video_mask)
        st.fwd.run(upto=st.fwd_encoders_len)
        return st.enc.seq_out.view(B, W, -1).clone(), st.enc.vis_out.view(B, F, -1).clone()

    def get_similarity_logits(self, sequence_output, visual_output, attention_mask, video_mask, shaped=False,
                              _pretrain_joint=False):
        """modeling.py:377-391: cross-encoder similarity when stage two / train_sim_after_cross, else masked means,
        L2 normalisation unless use_mil, text . video^T."""
        attention_mask = attention_mask.reshape(-1, attention_mask.shape[-1])
        video_mask = video_mask.reshape(-1, video_mask.shape[-1])
        _lib.lib()
        dev = sequence_output.device
        if dev.type != "cuda":
            raise RuntimeError("univl_amd.UniVL.get_similarity_logits needs HIP device tensors; no CPU fallback")
        Bt, W, Hd = sequence_output.shape
        Bv, F, _ = visual_output.shape
        seq = sequence_output.to(torch.float32).contiguous()
        vis = visual_output.to(torch.float32).contiguous()
        am = attention_mask.to(dev, torch.int64).contiguous()
        vm = video_mask.to(dev, torch.int64).contiguous()
        if (self._stage_two and _pretrain_joint is False) or self.train_sim_after_cross:
            return self._cross_similarity_eval(seq, vis, am, vm)
        norm = not bool(self.task_config.use_mil)
        ldt, ldv = (Bt + 3) // 4 * 4, (Bv + 3) // 4 * 4
        tn = torch.zeros(ldt, Hd, device=dev)
        vn = torch.zeros(ldv, Hd, device=dev)
        ops.pool_fwd(Bt, W, seq, am, skip_first=True, normalize=norm, out=tn)
        ops.pool_fwd(Bv, F, vis, vm, skip_first=False, normalize=norm, out=vn)
        sim = torch.empty(Bt, Bv, device=dev)
        ops.gemm(tn, vn, Bt, Bv, Hd, out32=sim)
        return sim

    def _cross_similarity_eval(self, seq, vis, am, vm, chunk_rows=5):
        """_cross_similarity (modeling.py:341-375) for cached features: every (text, video) pair through the cross
        encoder, `chunk_rows` text rows at a time (the reference's step_size = 5)."""
        from .steps import Ctx, CrossRun, PoolerSim, RowFeatures
        from .engine import Plan
        Bt, W, _ = seq.shape
        Bv, F, _ = vis.shape
        self._flush_pending()
        self.flat.refresh_shadow()
        out = torch.empty(Bt, Bv, device=seq.device)
        for lo in range(0, Bt, chunk_rows):
            n = min(chunk_rows, Bt - lo)
            key = ("xsim", n, Bv, W, F)
            ev = self._steps.get(key)
            if ev is None:
                cx = Ctx(self, False)
                feats = RowFeatures(cx, n, Bv, W, F)
                pairs = [(i, j) for i in range(n) for j in range(Bv)]
                run = CrossRun(cx, feats, [a for a, _ in pairs], [b for _, b in pairs])
                plan = Plan()
                run.build_forward(plan)
                pooler = PoolerSim(cx, run, n, Bv, None)
                pooler.build_forward(plan)
                ev = self._steps[key] = (feats, run, pooler, plan)
            feats, run, pooler, plan = ev
            feats.load(seq[lo:lo + n], vis, am[lo:lo + n], vm)
            plan.run()
            out[lo:lo + n].copy_(pooler.sim.view(n, Bv))
        return out

    def decoder_caption(self, sequence_output, visual_output, input_ids, attention_mask, video_mask, input_caption_ids,
                        decoder_mask, shaped=False, get_logits=False):
        """modeling.py:409-428: cross encoder on cat(text, video) then the decoder; returns the (n, len, vocab) logits or
        their argmax."""
        if self.decoder is None:
            raise RuntimeError("decoder_caption: this model was built without a decoder (stage one)")
        from .steps import Ctx, CrossRun, DecoderRun, RowFeatures
        from .engine import Plan
        attention_mask = attention_mask.reshape(-1, attention_mask.shape[-1])
        video_mask = video_mask.reshape(-1, video_mask.shape[-1])
        input_caption_ids = input_caption_ids.reshape(-1, input_caption_ids.shape[-1])
        decoder_mask = decoder_mask.reshape(-1, decoder_mask.shape[-1])
        B, W, _ = sequence_output.shape
        F, Wd = visual_output.shape[1], input_caption_ids.shape[-1]
        if sequence_output.device.type != "cuda":
            raise RuntimeError("univl_amd.UniVL.decoder_caption needs HIP device tensors; no CPU fallback")
        self._flush_pending()
        self.flat.refresh_shadow()
        key = ("caption_eval", B, W, F, Wd)
        ev = self._steps.get(key)
        if ev is None:
            cx = Ctx(self, False)
            feats = RowFeatures(cx, B, B, W, F)
            run = CrossRun(cx, feats, list(range(B)), list(range(B)))
            plan = Plan()
            run.build_forward(plan)
            dec = DecoderRun(cx, run, Wd, with_loss=False)
            dec.build_forward(plan)
            ev = self._steps[key] = (feats, run, dec, plan)
        feats, run, dec, plan = ev
        feats.load(sequence_output.to(torch.float32), visual_output.to(torch.float32), attention_mask, video_mask)
        dec.load(input_caption_ids, decoder_mask)
        plan.run()
        V = self.bert_config.vocab_size
        scores = dec.head.logits.view(B, Wd, -1)[:, :, :V]
        if get_logits:
            return scores.clone()
        return torch.max(scores, -1)[1]
