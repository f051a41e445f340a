"""Static execution plans for the four ways the reference drives `UniVL.forward` (modules/modeling.py:188-271):

    joint     stage one, FT-Joint:  encoders -> mean-pool similarity -> ranking / MIL-NCE loss          (:206-211)
    align     stage one, FT-Align (--train_sim_after_cross): encoders -> every (text, video) pair through the cross
              encoder -> pooler -> similarity_dense -> loss                                              (:341-375)
    caption   stage two, task_type "caption": encoders -> cross encoder -> decoder -> vocabulary CE      (:238-254)
    pretrain  stage two, do_pretrain: clean + masked encoder passes, cross encoder, MLM + MFM heads, joint
              similarity, decoder, cross-encoder alignment -- the five losses of :212-267

Each component below owns its persistent workspace and can append its forward and backward kernels to a Plan.
Gradient flow between components goes through fp32 accumulation buffers (dseq/dvis of an encoder pass, dcross of a
cross-encoder run) that the backward plan zeroes once and every consumer adds into.
"""
import os

import numpy as np
import torch

from . import _ab, _lib, ops
from .engine import DecoderStack, EncoderStack, GradState, Plan, _SiteCounter, _gemm_desc

H = 768


def _e(dev):
    from .engine import _workspace
    return _workspace(dev, zeros=True)


class Ctx:
    """Everything the components share for one compiled step."""

    def __init__(self, model, training):
        self.model, self.fl, self.training = model, model.flat, training
        self.tc = model.task_config
        self.dev, self.ct, self.dt = self.fl.device, self.fl.compute_dtype, self.fl.dt
        self.bf = self.ct == torch.bfloat16
        self.p = model.dropout_prob if training else 0.0
        self.seed, self.seed_dev = model._seed, model._seed_dev
        self.sites = _SiteCounter()
        self.e = _e(self.dev)
        self.red = model._reducer
        # UNIVL_STAMPS=1 (measurement, scripts/probe_branches.py): device wall-clock stamps between the nodes of the step, so that the
        # start / end of the two encoder branches inside a captured replay can be read WITHOUT a profiler attached
        self.stamps = {} if _ab.get("stamps") else None
        if self.stamps is not None:
            self.stamp_buf = torch.zeros(64, dtype=torch.int64, device=self.dev)
            model._stamps = (self.stamps, self.stamp_buf)

    def stamp(self, plan, name, stream=0):
        if self.stamps is None:
            return
        slot = self.stamps.setdefault(name, len(self.stamps))
        buf = self.stamp_buf
        plan.add_callable(lambda: ops.stamp(buf[slot:slot + 1]), stream)


def stage_input(dst, src):
    """dst (static device buffer) <- src, asynchronously on the current stream.  Host sources (the reference's loaders
    hand over pageable tensors: pin_memory=False, float64 video -- SURVEY.md section 8f row 3) are copied as they are:
    measured on the MI355X box (scripts/mb_h2d.py) the runtime's pageable path moves the 1.6 MB float64 video in 40 us,
    a private pinned staging buffer is no faster and shows 60-90 ms stalls, and a host-side float32 cast costs more
    than the bytes it saves.  The float64 -> float32 conversion happens inside the NormalizeVideo LayerNorm kernel."""
    src = torch.as_tensor(src)
    if src.dtype != dst.dtype:
        src = src.to(dst.dtype)
    dst.copy_(src.reshape(dst.shape), non_blocking=True)


def stage_inputs(pairs):
    """[(dst, src), ...] -> dst <- src for every pair.  Sources that already live on the destination's device with the same
    dtype and a contiguous layout (the usual case once a loader hands over device tensors, and always under
    graphed.GraphedTrainStep) go through ONE copy kernel (ops.copy_many): as separate dst.copy_(src) calls they are memcpy
    nodes of the captured step, ~10 us each in the round-2 kernel trace.  Everything else (host tensors, other dtypes, strided
    views) takes stage_input."""
    fast = []
    use_kernel = bool(_ab.get("copy_kernel"))        # A/B
    for dst, src in pairs:
        src = torch.as_tensor(src)
        if (use_kernel and src.device == dst.device and dst.is_cuda and src.dtype == dst.dtype and src.numel() == dst.numel()
                and src.is_contiguous() and dst.is_contiguous()):
            if src.data_ptr() != dst.data_ptr():
                fast.append((dst, src))
        else:
            stage_input(dst, src)
    if fast:
        ops.copy_many(fast)


class EncoderPass:
    """NormalizeVideo + BertModel + VisualModel for one set of inputs (modeling.py:196-202, 299-313)."""

    def __init__(self, cx, B, W, F, s_text=0, s_vis=2, normalized_input=False):
        self.cx, self.B, self.W, self.F = cx, B, W, F
        self.normalized_input = bool(normalized_input)     # get_sequence_visual_output(shaped=True): video arrives normalised
        e, ct, bf, fl = cx.e, cx.ct, cx.bf, cx.fl
        D = cx.tc.video_dim
        self.D, self.Tt, self.Tv = D, B * W, B * F
        # (Round 4, measured and removed: the video stack BEHIND the text stack on one stream where every product fills the chip by
        # itself -- 13.40 / 13.40 vs 11.99 / 12.11 ms at 128 pairs, 12.03 vs 9.25 at 64, 6.60 vs 5.30 at 32, profiles/r04m_ab_serial_branches.txt:
        # two streams also overlap one branch's tails and non-GEMM kernels with the other's products.)
        self.ST, self.SV = s_text, s_vis
        i64 = torch.int64
        self.ids, self.type_ids, self.amask = e(B, W, dtype=i64), e(B, W, dtype=i64), e(B, W, dtype=i64)
        self.video, self.vmask = e(B * F, D, dtype=torch.float64), e(B, F, dtype=i64)
        self.vy, self.vst, self.vn32 = e(self.Tv, D), e(self.Tv, 2), e(self.Tv, D)
        self.vn_op = e(self.Tv, D, dtype=ct) if bf else self.vn32
        self.ve, self.vest, self.v0_32 = e(self.Tv, H), e(self.Tv, 2), e(self.Tv, H)
        self.v0_16 = e(self.Tv, H, dtype=ct) if bf else self.v0_32
        self.te, self.test, self.t0_32 = e(self.Tt, H), e(self.Tt, 2), e(self.Tt, H)
        self.t0_16 = e(self.Tt, H, dtype=ct) if bf else self.t0_32
        self.dseq, self.dvis = e(self.Tt, H), e(self.Tv, H)
        self.de_op = e(self.Tv, H, dtype=ct)
        self.dvnorm = e(self.Tv, D)            # grad wrt the normalised video (accumulated: encoder + MFM loss)
        m = cx.model
        self.text = EncoderStack(fl, "bert", m.bert_config.num_hidden_layers, B, W, self.amask, cx.p, cx.seed_dev, cx.sites,
                                 s_main=s_text, s_side=s_text)
        self.vis = EncoderStack(fl, "visual", m.visual_config.num_hidden_layers, B, F, self.vmask, cx.p, cx.seed_dev, cx.sites,
                                s_main=s_vis, s_side=s_vis)
        self.off_t, self.off_v = cx.sites.next(), cx.sites.next()
        # lo halves of the stacks' input pairs (EncoderStack.pair_x) and of the normalised video (A operand of the video embedding product)
        self.t0_lo = e(self.Tt, H, dtype=ct) if self.text.pair_x else None
        self.v0_lo = e(self.Tv, H, dtype=ct) if self.vis.pair_x else None
        self.vn_lo = e(self.Tv, D, dtype=ct) if (self.vis.pair_x and not self.normalized_input) else None
        self.seq_out, self.seq_out16 = self.text.output()
        self.vis_out, self.vis_out16 = self.vis.output()

    N = dict(nv_g="normalize_video.visual_norm2d.weight", nv_b="normalize_video.visual_norm2d.bias",
             vw="visual.embeddings.word_embeddings.weight", vb="visual.embeddings.word_embeddings.bias",
             vpos="visual.embeddings.position_embeddings.weight", vlg="visual.embeddings.LayerNorm.weight",
             vlb="visual.embeddings.LayerNorm.bias", bw="bert.embeddings.word_embeddings.weight",
             bp="bert.embeddings.position_embeddings.weight", bt="bert.embeddings.token_type_embeddings.weight",
             blg="bert.embeddings.LayerNorm.weight", blb="bert.embeddings.LayerNorm.bias")

    def load(self, input_ids, token_type_ids, attention_mask, video, video_mask):
        B, W, F = self.B, self.W, self.F
        stage_inputs([(self.ids, input_ids), (self.type_ids, token_type_ids), (self.amask, attention_mask),
                      (self.vn32 if self.normalized_input else self.video, video), (self.vmask, video_mask)])

    def build_forward(self, fwd):
        cx, n, fl, dt, bf = self.cx, self.N, self.cx.fl, self.cx.dt, self.cx.bf
        W32, p, B, W, F, D, Tv = fl.w32, cx.p, self.B, self.W, self.F, self.D, self.Tv
        ST, SV = self.ST, self.SV
        # split-K accumulation targets; and the arrival counters of the LayerNorm folds (every launch leaves them zero: clearing them with
        # the arenas, in the same launch, only makes a step self-healing after a launch that did not complete)
        fwd.add_zeros([st.yarena for st in (self.text, self.vis) if st.zero_y] +
                      [st.ln_ctr for st in (self.text, self.vis) if st.ln_ctr is not None], ST)
        cx.stamp(fwd, "f_fork", ST)
        fwd.fork(ST, SV)           # the video encoder runs concurrently with the text encoder
        cx.stamp(fwd, "f_vis_start", SV)
        if self.normalized_input:
            if bf:
                fwd.add_callable(lambda: ops.cast_bf16(self.vn32, self.vn_op), SV)
        else:
            fwd.add("univl_layernorm_fwd", ops.layernorm_desc(
                dt, Tv, D, x=self.video, x_f64=True, gamma=W32(n["nv_g"]), beta=W32(n["nv_b"]), y=self.vy, stats=self.vst,
                out32=self.vn32, out16=self.vn_op if bf else None, out16_lo=self.vn_lo), SV)
        fwd.add("univl_gemm", _gemm_desc(dt, self.vn_op, D, fl.wop(n["vw"]), D, Tv, H, D, out32=self.ve, ldc=H, bias=W32(n["vb"]),
                                         a_lo=self.vn_lo, b_lo=fl.wlo(n["vw"]) if self.vis.pair_w else None), SV)
        fwd.add("univl_layernorm_fwd", ops.layernorm_desc(
            dt, Tv, H, x=self.ve, pos=W32(n["vpos"]), pos_period=F, gamma=W32(n["vlg"]), beta=W32(n["vlb"]), y=self.ve,
            stats=self.vest, out32=self.v0_32, out16=self.v0_16 if bf else None, p_post=p, seed=cx.seed, off_post=self.off_v,
            seed_dev=cx.seed_dev, out16_lo=self.v0_lo), SV)
        cx.stamp(fwd, "f_text_start", ST)
        fwd.add("univl_embed_text_fwd", ops.embed_text_desc(
            dt, B, W, self.ids, W32(n["bw"]), W32(n["bp"]), W32(n["blg"]), W32(n["blb"]), type_ids=self.type_ids,
            type_emb=W32(n["bt"]), y=self.te, stats=self.test, out32=self.t0_32, out16=self.t0_16 if bf else None, p_post=p,
            seed=cx.seed, off_post=self.off_t, seed_dev=cx.seed_dev, out16_lo=self.t0_lo), ST)
        self.vis.build_forward(fwd, self.v0_32, self.v0_16, cx.training, zero_arena=False, x16_lo=self.v0_lo)
        cx.stamp(fwd, "f_vis_end", SV)
        self.text.build_forward(fwd, self.t0_32, self.t0_16, cx.training, zero_arena=False, x16_lo=self.t0_lo)
        cx.stamp(fwd, "f_text_end", ST)
        fwd.join(SV, ST)
        cx.stamp(fwd, "f_join", ST)

    def zero_list(self):
        """Accumulation buffers a backward clears before anything adds into them."""
        return ([self.dseq, self.dvis, self.dvnorm] + [st.garena for st in (self.text, self.vis) if st.zero_g] +
                [st.ln_ctr_b for st in (self.text, self.vis) if st.ln_ctr_b is not None])

    def build_backward(self, bwd, gs, hook=None):
        """Consumes self.dseq / self.dvis (+ whatever was accumulated into self.dvnorm)."""
        cx, n, fl, dt = self.cx, self.N, self.cx.fl, self.cx.dt
        W32, G, p, B, W, F, D, Tv = fl.w32, fl.g, cx.p, self.B, self.W, self.F, self.D, self.Tv
        ST, SV = self.ST, self.SV
        cx.stamp(bwd, "b_fork", ST)
        bwd.fork(ST, SV)
        cx.stamp(bwd, "b_vis_start", SV)
        cx.stamp(bwd, "b_text_start", ST)
        # The two stacks are independent until the join below and run on two streams; they are emitted INTERLEAVED
        # (text layers : video layers in the ratio of their depths) so that plan order follows time order -- gradient
        # exchange points and hipGraph segment cuts (Plan.run_graphed) then fall between layers of both stacks.
        gv = self.vis.backward_layers(bwd, self.dvis, self.v0_32, self.v0_16, gs, cx.training, layer_hook=hook, zero_arena=False)
        gt = self.text.backward_layers(bwd, self.dseq, self.t0_32, self.t0_16, gs, cx.training, layer_hook=hook, zero_arena=False)
        ratio = max(1, int(round(self.text.L / max(1, self.vis.L))))
        done_t = done_v = False
        while not (done_t and done_v):
            for _ in range(ratio):
                if not done_t and next(gt, None) is None:
                    done_t = True
                    self._text_tail(bwd, self.text.bwd_out)
                    cx.stamp(bwd, "b_text_end", ST)
            if not done_v and next(gv, None) is None:
                done_v = True
                self._video_tail(bwd, gs, self.vis.bwd_out)
                cx.stamp(bwd, "b_vis_end", SV)
        bwd.join(SV, ST)
        cx.stamp(bwd, "b_join", ST)

    # From this many rows per position the position-table gradients are built by a gather over the per-token gradient rows
    # (univl_rows_gather_sum) instead of B atomics per table element inside the fused kernels (128 pairs: embed_bwd 173 us, the video
    # embedding's LayerNorm backward 189 us -- profiles/r03q_bench_b128_kernel_stats.csv).  UNIVL_DPOS_GATHER_MIN=0: never.
    def _gather_dpos(self):
        thr = _ab.get("dpos_gather_min")        # read when the plan is built (a test lowers it)
        return thr > 0 and self.B >= thr

    def _video_tail(self, bwd, gs, dxv):
        cx, n, fl, dt = self.cx, self.N, self.cx.fl, self.cx.dt
        W32, G, p, F, D, Tv, SV = fl.w32, fl.g, cx.p, self.F, self.D, self.Tv, self.SV
        gather = self._gather_dpos()
        if gather and getattr(self, "dxe", None) is None:
            self.dxe = cx.e(Tv, H)                       # grad wrt the pre-LayerNorm sum = grad of the position rows
        bwd.add("univl_layernorm_bwd", ops.layernorm_desc(
            dt, Tv, H, gamma=W32(n["vlg"]), y=self.ve, stats=self.vest, dout=dxv, dxd16=self.de_op, dgamma=G(n["vlg"]),
            dbeta=G(n["vlb"]), dbias=G(n["vb"]), dpos=None if gather else G(n["vpos"]), dx32=self.dxe if gather else None,
            pos_period=F, p_post=p, seed=cx.seed, off_post=self.off_v, seed_dev=cx.seed_dev), SV)
        if gather:
            bwd.add_callable(lambda: ops.rows_gather_sum(self.dxe, F, G(n["vpos"])[:F]), SV)
        bwd.add("univl_gemm", _gemm_desc(dt, self.de_op, H, self.vn_op, D, H, D, Tv, trans_a=1, trans_b=1, out32=G(n["vw"]),
                                         ldc=D, accumulate=gs.acc(n["vw"])), SV)
        bwd.add("univl_gemm", _gemm_desc(dt, self.de_op, H, fl.wop(n["vw"]), D, Tv, D, H, trans_b=1, out32=self.dvnorm, ldc=D,
                                         accumulate=True), SV)
        bwd.add("univl_layernorm_bwd", ops.layernorm_desc(
            dt, Tv, D, gamma=W32(n["nv_g"]), y=self.vy, stats=self.vst, dout=self.dvnorm, dgamma=G(n["nv_g"]), dbeta=G(n["nv_b"])), SV)

    def _text_tail(self, bwd, dxt):
        cx, n, fl, dt = self.cx, self.N, self.cx.fl, self.cx.dt
        W32, G, p, B, W, ST = fl.w32, fl.g, cx.p, self.B, self.W, self.ST
        sparse = getattr(self, "sparse_word_grad", False)
        gather = self._gather_dpos()
        if (sparse or gather) and getattr(self, "drows", None) is None:
            self.drows = cx.e(self.Tt, H)
        bwd.add("univl_embed_text_bwd", ops.embed_text_desc(
            dt, B, W, self.ids, W32(n["bw"]), W32(n["bp"]), W32(n["blg"]), W32(n["blb"]), type_ids=self.type_ids,
            type_emb=W32(n["bt"]), y=self.te, stats=self.test, p_post=p, seed=cx.seed, off_post=self.off_t,
            seed_dev=cx.seed_dev, dout=dxt, dword=G(n["bw"]), dpos=None if gather else G(n["bp"]), dtype_emb=G(n["bt"]),
            dgamma=G(n["blg"]), dbeta=G(n["blb"]), drows=self.drows if (sparse or gather) else None), ST)
        if gather:
            # the per-token rows: position table by the gather; word table by the scatter kernel (few rows share an id) unless the
            # data-parallel exchange takes the rows themselves
            bwd.add_callable(lambda: ops.rows_gather_sum(self.drows, W, G(n["bp"])[:W]), ST)
            if not sparse:
                bwd.add_callable(lambda: ops.embed_scatter(self.ids.view(-1), self.drows, 1.0, G(n["bw"])), ST)
        rows = getattr(self, "word_rows", None)
        if rows is not None and not sparse:            # remember which table rows this backward wrote (engine.FlatParams.word_rows)
            lst, meta, reset = rows
            bwd.add_callable(lambda: ops.rows_append(self.ids.view(-1), lst, meta, reset, cx.fl.word_ever), ST)


class SimLoss:
    """loss_fct on a [n, ld] similarity matrix (modeling.py:172-184): MaxMarginRankingLoss / MILNCELoss / CrossEn."""

    def __init__(self, cx, kind, n):
        self.cx, self.kind, self.n = cx, kind, n
        self.loss = cx.e(1)
        tc = cx.tc
        self.wts = None
        if kind == "maxmargin":
            bsz = tc.batch_size // tc.n_gpu
            if tc.negative_weighting and tc.n_pair > 1 and bsz > 1:     # until_module.py:238-243
                easy = 1 - tc.hard_negative_rate
                alpha = easy / ((bsz - 1) * (1 - easy))
                mm = np.kron((1 - alpha) * np.eye(bsz) + alpha, np.ones((tc.n_pair, tc.n_pair))) * (bsz * (1 - easy))
                self.wts = torch.tensor(mm, dtype=torch.float32, device=cx.dev).contiguous()

    def build(self, fwd, sim, dsim):
        """sim / dsim: [n, ld] views."""
        tc, n = self.cx.tc, self.n
        if self.kind == "milnce":
            bs, npair = n // tc.n_pair, tc.n_pair
            fwd.add_callable(lambda: ops.milnce_loss(sim, bs, npair, self.loss, dsim))
        elif self.kind == "crossen":
            fwd.add_callable(lambda: ops.crossen_loss(sim, self.loss, dsim))
        else:
            margin, wts = float(tc.margin), self.wts
            fwd.add_callable(lambda: ops.maxmargin_loss(sim, margin, wts, self.loss, dsim))


class JointSim:
    """get_similarity_logits, mean-pooling branch (modeling.py:327-339, 383-389)."""

    def __init__(self, cx, enc, loss_kind):
        self.cx, self.enc = cx, enc
        B = enc.B
        e = cx.e
        self.ldp = (B + 3) // 4 * 4
        self.tmean, self.tn, self.vmean, self.vn = e(B, H), e(self.ldp, H), e(B, H), e(self.ldp, H)
        self.sim, self.dsim = e(self.ldp, self.ldp), e(self.ldp, self.ldp)
        self.dtn, self.dvn = e(B, H), e(B, H)
        self.norm = not bool(cx.tc.use_mil)
        # one launch for both poolings, one for the whole similarity-head backward (UNIVL_FUSED_SIM=0: the separate kernels); up to
        # 256 rows the per-row loop over the other modality's matrix is cheaper than two extra launches
        self.fused = bool(_ab.get("fused_sim")) and B <= 256
        self.lossfn = SimLoss(cx, loss_kind, B)
        self.loss = self.lossfn.loss

    def build_forward(self, fwd):
        enc, B = self.enc, self.enc.B
        dt_ = ops.pool_desc(B, enc.W, enc.seq_out, enc.amask, skip_first=True, normalize=self.norm, mean=self.tmean, out=self.tn)
        dv_ = ops.pool_desc(B, enc.F, enc.vis_out, enc.vmask, skip_first=False, normalize=self.norm, mean=self.vmean, out=self.vn)
        if self.fused:
            fwd.add_pair_call("univl_pool_pair_fwd", dt_, dv_)          # text and video pooling: one launch
        else:
            fwd.add("univl_pool_fwd", dt_)
            fwd.add("univl_pool_fwd", dv_)
        fwd.add("univl_gemm", _gemm_desc(_lib.DT_F32, self.tn, H, self.vn, H, B, B, H, out32=self.sim, ldc=self.ldp))
        self.lossfn.build(fwd, self.sim[:B], self.dsim[:B])

    def build_backward(self, bwd, gout):
        enc, B, ldp = self.enc, self.enc.B, self.ldp
        if self.fused:
            # torch.matmul(text, video.t()) backward (modeling.py:389), the upstream factor and both pooling backwards in ONE launch
            # (UnivlPool.dsim): five kernels of the chain between the forward and the encoders' backward become one
            bwd.add_pair_call(
                "univl_pool_pair_bwd",
                ops.pool_desc(B, enc.W, enc.seq_out, enc.amask, skip_first=True, normalize=self.norm, mean=self.tmean, out=self.tn,
                              dx=enc.dseq, accumulate=True, dsim=self.dsim, other=self.vn, n_other=B, transpose=False, gscale=gout),
                ops.pool_desc(B, enc.F, enc.vis_out, enc.vmask, skip_first=False, normalize=self.norm, mean=self.vmean, out=self.vn,
                              dx=enc.dvis, accumulate=True, dsim=self.dsim, other=self.tn, n_other=B, transpose=True, gscale=gout))
            return
        bwd.add_callable(lambda: ops.scale_by_device_scalar(self.dsim, gout))
        bwd.add("univl_gemm", _gemm_desc(_lib.DT_F32, self.dsim, ldp, self.vn, H, B, H, B, trans_b=1, out32=self.dtn, ldc=H))
        bwd.add("univl_gemm", _gemm_desc(_lib.DT_F32, self.dsim, ldp, self.tn, H, B, H, B, trans_a=1, trans_b=1, out32=self.dvn, ldc=H))
        bwd.add("univl_pool_bwd", ops.pool_desc(B, enc.W, enc.seq_out, enc.amask, skip_first=True, normalize=self.norm,
                                                mean=self.tmean, out=self.tn, dout=self.dtn, dx=enc.dseq, accumulate=True))
        bwd.add("univl_pool_bwd", ops.pool_desc(B, enc.F, enc.vis_out, enc.vmask, skip_first=False, normalize=self.norm,
                                                mean=self.vmean, out=self.vn, dout=self.dvn, dx=enc.dvis, accumulate=True))


class CrossRun:
    """CrossModel over P sequences cat(text row tidx[p], video row vidx[p]) (modeling.py:315-325, module_cross.py:364-394)."""

    def __init__(self, cx, enc, tidx, vidx, stream=0):
        self.cx, self.enc = cx, enc
        self.P = len(tidx)
        self.W, self.F = enc.W, enc.F
        self.S = S = enc.W + enc.F
        self.T = T = self.P * S
        e, ct, bf, fl = cx.e, cx.ct, cx.bf, cx.fl
        self.sm = stream
        self.tidx = torch.tensor(list(tidx), dtype=torch.int32, device=cx.dev)
        self.vidx = torch.tensor(list(vidx), dtype=torch.int32, device=cx.dev)
        self.cmask = e(self.P, S, dtype=torch.int64)
        self.concat, self.cst = e(T, H), e(T, 2)
        self.c0_32 = e(T, H)
        self.c0_16 = e(T, H, dtype=ct) if bf else self.c0_32
        self.postype, self.dpostype = e(S, H), e(S, H)
        self.dcross = e(T, H)
        self.dconcat = e(T, H)
        L = cx.model.cross_config.num_hidden_layers
        self.stack = EncoderStack(fl, "cross", L, self.P, S, self.cmask, cx.p, cx.seed_dev, cx.sites, s_main=stream, s_side=stream)
        self.off = cx.sites.next()
        self.out32, self.out16 = self.stack.output()

    N = dict(pos="cross.embeddings.position_embeddings.weight", typ="cross.embeddings.token_type_embeddings.weight",
             lg="cross.embeddings.LayerNorm.weight", lb="cross.embeddings.LayerNorm.bias")

    def build_forward(self, fwd):
        cx, enc, n, fl, dt, sm = self.cx, self.enc, self.N, self.cx.fl, self.cx.dt, self.sm
        W32 = fl.w32
        P, W, F, S, T = self.P, self.W, self.F, self.S, self.T
        fwd.add_callable(lambda: ops.postype_fwd(W32(n["pos"]), W32(n["typ"]), W, S, self.postype), sm)
        fwd.add_callable(lambda: ops.pair_concat_fwd(enc.seq_out, enc.vis_out, enc.amask, enc.vmask, self.tidx, self.vidx, P, W, F,
                                                     self.concat, self.cmask), sm)
        fwd.add("univl_layernorm_fwd", ops.layernorm_desc(
            dt, T, H, x=self.concat, pos=self.postype, pos_period=S, gamma=W32(n["lg"]), beta=W32(n["lb"]), y=self.concat,
            stats=self.cst, out32=self.c0_32, out16=self.c0_16 if cx.bf else None, p_post=cx.p, seed=cx.seed, off_post=self.off,
            seed_dev=cx.seed_dev), sm)
        self.stack.build_forward(fwd, self.c0_32, self.c0_16, cx.training)

    def zero_list(self):
        return [self.dcross, self.dpostype]

    def build_backward(self, bwd, gs, hook=None):
        """Consumes self.dcross (accumulated by the consumers); adds into enc.dseq / enc.dvis."""
        cx, enc, n, fl, dt, sm = self.cx, self.enc, self.N, self.cx.fl, self.cx.dt, self.sm
        W32, G = fl.w32, fl.g
        P, W, F, S, T = self.P, self.W, self.F, self.S, self.T
        dx0 = self.stack.build_backward(bwd, self.dcross, self.c0_32, self.c0_16, gs, cx.training, layer_hook=hook)
        bwd.add("univl_layernorm_bwd", ops.layernorm_desc(
            dt, T, H, gamma=W32(n["lg"]), y=self.concat, stats=self.cst, dout=dx0, dx32=self.dconcat, dgamma=G(n["lg"]),
            dbeta=G(n["lb"]), dpos=self.dpostype, pos_period=S, p_post=cx.p, seed=cx.seed, off_post=self.off,
            seed_dev=cx.seed_dev), sm)
        bwd.add_callable(lambda: ops.pair_concat_bwd(self.dconcat, self.tidx, self.vidx, P, W, F, enc.dseq, enc.dvis), sm)
        bwd.add_callable(lambda: ops.postype_bwd(self.dpostype, W, S, G(n["pos"]), G(n["typ"])), sm)


class PoolerSim:
    """CrossPooler + similarity_dense on a CrossRun over all (text, video) pairs (modeling.py:369-373)."""

    def __init__(self, cx, run, Bt, Bv, loss_kind):
        self.cx, self.run, self.Bt, self.Bv = cx, run, Bt, Bv
        e, ct = cx.e, cx.ct
        P = run.P
        self.pre, self.pooled, self.sim, self.dsim = e(P, H), e(P, H), e(P), e(P)
        self.dpooled, self.dpre = e(P, H), e(P, H, dtype=ct)
        self.lossfn = SimLoss(cx, loss_kind, Bt) if loss_kind else None
        self.loss = self.lossfn.loss if self.lossfn else None

    N = dict(pw="cross.pooler.dense.weight", pb="cross.pooler.dense.bias", sw="similarity_dense.weight", sb="similarity_dense.bias")

    def build_forward(self, fwd):
        cx, run, n, fl, dt = self.cx, self.run, self.N, self.cx.fl, self.cx.dt
        P, S = run.P, run.S
        x0 = run.out16            # first token of every sequence: row stride S*H
        fwd.add("univl_gemm", _gemm_desc(dt, x0, S * H, fl.wop(n["pw"]), H, P, H, H, out32=self.pre, ldc=H, bias=fl.w32(n["pb"])), run.sm)
        fwd.add_callable(lambda: ops.tanh_fwd(self.pre, self.pooled), run.sm)
        fwd.add_callable(lambda: ops.simdense_fwd(self.pooled, fl.w32(n["sw"]), fl.w32(n["sb"]), self.sim), run.sm)
        if self.lossfn:
            self.lossfn.build(fwd, self.sim.view(self.Bt, self.Bv), self.dsim.view(self.Bt, self.Bv))

    def build_backward(self, bwd, gs, gout):
        cx, run, n, fl, dt = self.cx, self.run, self.N, self.cx.fl, self.cx.dt
        P, S, G = run.P, run.S, fl.g
        bwd.add_callable(lambda: ops.scale_by_device_scalar(self.dsim, gout), run.sm)
        bwd.add_callable(lambda: ops.simdense_bwd(self.dsim, self.pooled, fl.w32(n["sw"]), self.dpooled, G(n["sw"]), G(n["sb"])), run.sm)
        bwd.add_callable(lambda: ops.tanh_bwd(self.dpooled, self.pooled, self.dpre), run.sm)
        bwd.add("univl_gemm", _gemm_desc(dt, self.dpre, H, run.out16, S * H, H, H, P, trans_a=1, trans_b=1, out32=G(n["pw"]), ldc=H,
                                         accumulate=gs.acc(n["pw"]), dbias=G(n["pb"])), run.sm)
        bwd.add("univl_gemm", _gemm_desc(dt, self.dpre, H, fl.wop(n["pw"]), H, P, H, H, trans_b=1, out32=run.dcross, ldc=S * H,
                                         accumulate=True), run.sm)


class VocabHead:
    """BertLMPredictionHead tied to BERT's word table + CrossEntropyLoss(ignore_index=-1)
    (module_bert.py:299-330 / module_decoder.py:156-183; modeling.py:252-254, 273-276)."""

    def __init__(self, cx, prefix, T, stream=0):
        self.cx, self.prefix, self.T, self.sm = cx, prefix, T, stream
        e, ct, bf = cx.e, cx.ct, cx.bf
        V = cx.model.bert_config.vocab_size
        self.V, self.ldv = V, (V + 7) // 8 * 8
        self.u, self.hy, self.hst = e(T, H, dtype=ct), e(T, H), e(T, 2)
        self.h32 = e(T, H)
        self.h16 = e(T, H, dtype=ct) if bf else self.h32
        self._logits = None                      # materialised only where somebody reads them (evaluation / decoding, vocab_ce=0)
        self.dlogits = e(T, self.ldv, dtype=ct)
        self.labels = e(T, dtype=torch.int64)
        self.loss, self.scratch = e(1), e(2)
        self.dh, self.dg, self.du = e(T, H), e(T, H), e(T, H, dtype=ct)
        # K16 (univl_vocab_ce_fwd / _bwd): per-row (max, sum exp) pairs of the 128-column tiles, the label's logit, the row's log-sum-exp
        self.k16 = None
        self.slots = (V + 127) // 128

    @property
    def logits(self):
        if self._logits is None:
            self._logits = self.cx.e(self.T, self.ldv)
        return self._logits

    def _k16_desc(self, x16, bias, table):
        cx, T = self.cx, self.T
        e = cx.e
        self.k16 = k = dict(partial=e(T, self.slots, 2), label_logit=e(T), lse=e(T), rowloss=e(T))
        d = _lib.VocabCE()
        d.dtype, d.rows, d.V, d.K = cx.dt, T, self.V, H
        d.x, d.ldx, d.table, d.ldt, d.bias = x16.data_ptr(), H, table.data_ptr(), H, bias.data_ptr()
        d.labels, d.ignore_index, d.slots = self.labels.data_ptr(), -1, self.slots
        d.partial, d.label_logit, d.lse, d.rowloss = (k["partial"].data_ptr(), k["label_logit"].data_ptr(), k["lse"].data_ptr(),
                                                      k["rowloss"].data_ptr())
        d.scratch2, d.loss = self.scratch.data_ptr(), self.loss.data_ptr()
        d.gout, d.dlogits, d.lddl = None, self.dlogits.data_ptr(), self.ldv
        k["desc"] = d
        k["keep"] = (x16, bias, table)
        return d

    def names(self):
        p = self.prefix
        return dict(tw=p + ".transform.dense.weight", tb=p + ".transform.dense.bias", lg=p + ".transform.LayerNorm.weight",
                    lb=p + ".transform.LayerNorm.bias", bias=p + ".bias", emb="bert.embeddings.word_embeddings.weight")

    def build_forward(self, fwd, x16, with_loss=True):
        cx, fl, dt, n, T, sm = self.cx, self.cx.fl, self.cx.dt, self.names(), self.T, self.sm
        W32 = fl.w32
        fwd.add("univl_gemm", _gemm_desc(dt, x16, H, fl.wop(n["tw"]), H, T, H, H, out32=self.hy, ldc=H, bias=W32(n["tb"]),
                                         aux=self.u, ldaux=H, gelu="fwd"), sm)
        fwd.add("univl_layernorm_fwd", ops.layernorm_desc(dt, T, H, x=self.hy, gamma=W32(n["lg"]), beta=W32(n["lb"]), y=self.hy,
                                                          stats=self.hst, out32=self.h32, out16=self.h16 if cx.bf else None), sm)
        if with_loss and _ab.get("vocab_ce"):
            # K16: the product's epilogue keeps the online log-softmax statistics; the [T, 30522] logits never exist (module_bert.py:327-330 +
            # modeling.py:253 / 275 in one entry point); the backward recomputes the product into dlogits
            fwd.add("univl_vocab_ce_fwd", self._k16_desc(self.h16, W32(n["bias"]), fl.wop(n["emb"])), sm)
            return
        fwd.add("univl_gemm", _gemm_desc(dt, self.h16, H, fl.wop(n["emb"]), H, T, self.V, H, out32=self.logits, ldc=self.ldv,
                                         bias=W32(n["bias"])), sm)
        if with_loss:
            fwd.add_callable(lambda: ops.ce_loss(self.logits, self.labels, self.V, self.scratch, self.loss, self.dlogits), sm)

    def build_backward(self, bwd, gs, gout, x16, dx32, accumulate_dx):
        """dx32 receives the gradient wrt the head input (accumulated if accumulate_dx)."""
        cx, fl, dt, n, T, sm = self.cx, self.cx.fl, self.cx.dt, self.names(), self.T, self.sm
        W32, G = fl.w32, fl.g
        if self.k16 is not None:
            db = _lib.VocabCE.from_buffer_copy(self.k16["desc"])      # the forward's descriptor + the upstream gradient of this loss term
            db.gout = gout.data_ptr()
            self.k16["gout"] = gout
            bwd.add("univl_vocab_ce_bwd", db, sm)
        else:
            bwd.add_callable(lambda: ops.scale_ct(self.dlogits, gout), sm)
        # dh = dlogits . E contracts over the 30522-word vocabulary with only (T / 64) x 12 output tiles: unsplit, each workgroup
        # walks 239 K steps alone (a ~240 us latency chain at T = 512).  Split over the vocabulary so that ~512 workgroups share it
        # (fp32 atomics into the pre-zeroed dh; one slice in deterministic mode).  UNIVL_VOCAB_DGRAD_SPLIT=0: unsplit.
        ks = 1
        if _ab.get("vocab_dgrad_split"):
            ks = max(1, min(16, 512 // max(1, ((T + 63) // 64) * (H // 64))))
        if ks > 1:
            bwd.add_zeros([self.dh], sm)
        bwd.add("univl_gemm", _gemm_desc(dt, self.dlogits, self.ldv, fl.wop(n["emb"]), H, T, H, self.V, trans_b=1, out32=self.dh, ldc=H,
                                         ksplit=ks), sm)
        bwd.add("univl_gemm", _gemm_desc(dt, self.dlogits, self.ldv, self.h16, H, self.V, H, T, trans_a=1, trans_b=1, out32=G(n["emb"]),
                                         ldc=H, accumulate=True, dbias=G(n["bias"])), sm)
        bwd.add("univl_layernorm_bwd", ops.layernorm_desc(dt, T, H, gamma=W32(n["lg"]), y=self.hy, stats=self.hst, dout=self.dh,
                                                          dx32=self.dg, dgamma=G(n["lg"]), dbeta=G(n["lb"])), sm)
        bwd.add_callable(lambda: ops.gelu_bwd(self.dg, self.u, self.du), sm)
        bwd.add("univl_gemm", _gemm_desc(dt, self.du, H, x16, H, H, H, T, trans_a=1, trans_b=1, out32=G(n["tw"]), ldc=H,
                                         accumulate=gs.acc(n["tw"]), dbias=G(n["tb"])), sm)
        bwd.add("univl_gemm", _gemm_desc(dt, self.du, H, fl.wop(n["tw"]), H, T, H, H, trans_b=1, out32=dx32, ldc=H,
                                         accumulate=accumulate_dx), sm)


class RowFeatures:
    """Stand-in for an EncoderPass when the caller supplies sequence_output / visual_output tensors
    (decoder_caption, get_similarity_logits on cached features: main_task_caption.py:450, main_task_retrieval.py:376)."""

    def __init__(self, cx, Bt, Bv, W, F):
        e = cx.e
        self.B, self.W, self.F = Bt, W, F
        self.seq_out, self.vis_out = e(Bt * W, H), e(Bv * F, H)
        self.amask, self.vmask = e(Bt, W, dtype=torch.int64), e(Bv, F, dtype=torch.int64)
        self.dseq, self.dvis = None, None

    def load(self, seq, vis, amask, vmask):
        self.seq_out.copy_(seq.reshape(self.seq_out.shape), non_blocking=True)
        self.vis_out.copy_(vis.reshape(self.vis_out.shape), non_blocking=True)
        self.amask.copy_(amask.reshape(self.amask.shape), non_blocking=True)
        self.vmask.copy_(vmask.reshape(self.vmask.shape), non_blocking=True)


class DecoderRun:
    """DecoderModel.forward (module_decoder.py:372-406) on top of a row-wise CrossRun + the caption CE (modeling.py:246-254)."""

    def __init__(self, cx, run, Wd, with_loss=True):
        self.cx, self.run, self.Wd = cx, run, Wd
        B = run.P
        self.B, self.T = B, B * Wd
        e, ct, bf, fl = cx.e, cx.ct, cx.bf, cx.fl
        T = self.T
        self.cap_ids, self.dmask = e(B, Wd, dtype=torch.int64), e(B, Wd, dtype=torch.int64)
        self.ey, self.est, self.e0_32 = e(T, H), e(T, 2), e(T, H)
        self.e0_16 = e(T, H, dtype=ct) if bf else self.e0_32
        L = cx.model.decoder_config.num_decoder_layers
        self.stack = DecoderStack(fl, L, B, Wd, run.S, self.dmask, run.cmask, cx.p, cx.seed_dev, cx.sites, stream=run.sm)
        self.head = VocabHead(cx, "decoder.classifier.cls.predictions", T, stream=run.sm)
        self.off = cx.sites.next()
        self.dout = e(T, H)
        self.with_loss = with_loss
        self.loss = self.head.loss

    N = dict(bw="bert.embeddings.word_embeddings.weight", bp="bert.embeddings.position_embeddings.weight",
             lg="decoder.embeddings.LayerNorm.weight", lb="decoder.embeddings.LayerNorm.bias")

    def load(self, input_caption_ids, decoder_mask, output_caption_ids=None):
        self.cap_ids.copy_(input_caption_ids.reshape(self.B, self.Wd), non_blocking=True)
        self.dmask.copy_(decoder_mask.reshape(self.B, self.Wd), non_blocking=True)
        if output_caption_ids is not None:
            self.head.labels.copy_(output_caption_ids.reshape(-1), non_blocking=True)

    def build_forward(self, fwd):
        cx, n, fl, dt, sm = self.cx, self.N, self.cx.fl, self.cx.dt, self.run.sm
        W32 = fl.w32
        fwd.add("univl_embed_text_fwd", ops.embed_text_desc(
            dt, self.B, self.Wd, self.cap_ids, W32(n["bw"]), W32(n["bp"]), W32(n["lg"]), W32(n["lb"]), y=self.ey, stats=self.est,
            out32=self.e0_32, out16=self.e0_16 if cx.bf else None, p_post=cx.p, seed=cx.seed, off_post=self.off,
            seed_dev=cx.seed_dev), sm)
        self.stack.build_forward(fwd, self.e0_32, self.e0_16, self.run.out16, cx.training)
        self.head.build_forward(fwd, self.stack.output()[1], with_loss=self.with_loss)

    def build_backward(self, bwd, gs, gout):
        cx, n, fl, dt, sm = self.cx, self.N, self.cx.fl, self.cx.dt, self.run.sm
        W32, G = fl.w32, fl.g
        self.head.build_backward(bwd, gs, gout, self.stack.output()[1], self.dout, accumulate_dx=False)
        dx0 = self.stack.build_backward(bwd, self.dout, self.e0_32, self.e0_16, self.run.out16, self.run.dcross, gs, cx.training)
        bwd.add("univl_embed_text_bwd", ops.embed_text_desc(
            dt, self.B, self.Wd, self.cap_ids, W32(n["bw"]), W32(n["bp"]), W32(n["lg"]), W32(n["lb"]), y=self.ey, stats=self.est,
            p_post=cx.p, seed=cx.seed, off_post=self.off, seed_dev=cx.seed_dev, dout=dx0, dword=G(n["bw"]), dpos=G(n["bp"]),
            dgamma=G(n["lg"]), dbeta=G(n["lb"])), sm)


class PretrainHeads:
    """MLM + MFM heads on a row-wise CrossRun over the MASKED inputs (modeling.py:224-231, 273-297)."""

    def __init__(self, cx, run, enc_clean):
        self.cx, self.run, self.clean = cx, run, enc_clean
        e, ct, bf = cx.e, cx.ct, cx.bf
        B, W, F = run.P, run.W, run.F
        self.B, self.W, self.F = B, W, F
        Tt, Tv, D = B * W, B * F, cx.tc.video_dim
        self.Tt, self.Tv, self.D = Tt, Tv, D
        self.arange = torch.arange(B, dtype=torch.int32, device=cx.dev)
        self.tpart, self.vpart = e(Tt, H), e(Tv, H)
        self.tpart16 = e(Tt, H, dtype=ct) if bf else self.tpart
        self.vpart16 = e(Tv, H, dtype=ct) if bf else self.vpart
        self.dtpart, self.dvpart = e(Tt, H), e(Tv, H)
        self.mlm = VocabHead(cx, "cls.predictions", Tt, stream=run.sm)
        # visual head
        self.u, self.hy, self.hst, self.h32 = e(Tv, H, dtype=ct), e(Tv, H), e(Tv, 2), e(Tv, H)
        self.h16 = e(Tv, H, dtype=ct) if bf else self.h32
        self.scores = e(Tv, D, dtype=ct)
        self.ldl = (Tv + 7) // 8 * 8
        self.logits = e(Tv, self.ldl)
        self.dlogits_ct = e(Tv, self.ldl, dtype=ct) if bf else self.logits
        self.dscores = e(Tv, D, dtype=ct)
        self.dh, self.dg, self.du = e(Tv, H), e(Tv, H), e(Tv, H, dtype=ct)
        self.vlabels = e(Tv, dtype=torch.int64)
        self.nce_loss, self.scratch = e(1), e(2)

    VN = dict(tw="cls_visual.predictions.transform.dense.weight", tb="cls_visual.predictions.transform.dense.bias",
              lg="cls_visual.predictions.transform.LayerNorm.weight", lb="cls_visual.predictions.transform.LayerNorm.bias",
              bias="cls_visual.predictions.bias", vw="visual.embeddings.word_embeddings.weight")

    def load(self, pairs_token_labels, video_labels_index):
        self.mlm.labels.copy_(pairs_token_labels.reshape(-1), non_blocking=True)
        self.vlabels.copy_(video_labels_index.reshape(-1), non_blocking=True)

    def build_forward(self, fwd):
        cx, run, fl, dt, n, sm = self.cx, self.run, self.cx.fl, self.cx.dt, self.VN, self.run.sm
        W32 = fl.w32
        B, W, F, Tv, D = self.B, self.W, self.F, self.Tv, self.D
        # torch.split(cross_output, [W, F], dim=1)  (modeling.py:225)
        fwd.add_callable(self.tpart.zero_, sm)
        fwd.add_callable(self.vpart.zero_, sm)
        fwd.add_callable(lambda: ops.pair_concat_bwd(run.out32, self.arange, self.arange, B, W, F, self.tpart, self.vpart), sm)
        if cx.bf:
            fwd.add_callable(lambda: ops.cast_bf16(self.tpart, self.tpart16), sm)
            fwd.add_callable(lambda: ops.cast_bf16(self.vpart, self.vpart16), sm)
        self.mlm.build_forward(fwd, self.tpart16)
        # VisualLMPredictionHead (module_visual.py:298-311): LN(gelu(dense(x))) . W_vis[768,1024] + bias
        fwd.add("univl_gemm", _gemm_desc(dt, self.vpart16, H, fl.wop(n["tw"]), H, Tv, H, H, out32=self.hy, ldc=H, bias=W32(n["tb"]),
                                         aux=self.u, ldaux=H, gelu="fwd"), sm)
        fwd.add("univl_layernorm_fwd", ops.layernorm_desc(dt, Tv, H, x=self.hy, gamma=W32(n["lg"]), beta=W32(n["lb"]), y=self.hy,
                                                          stats=self.hst, out32=self.h32, out16=self.h16 if cx.bf else None), sm)
        fwd.add("univl_gemm", _gemm_desc(dt, self.h16, H, fl.wop(n["vw"]), D, Tv, D, H, trans_b=1, out16=self.scores, ldc=D,
                                         bias=W32(n["bias"])), sm)
        # logits_matrix = afm_scores . video^T against all normalised CLEAN frames of the batch (modeling.py:282-285)
        fwd.add("univl_gemm", _gemm_desc(dt, self.scores, D, self.clean.vn_op, D, Tv, Tv, D, out32=self.logits, ldc=self.ldl), sm)
        fwd.add_callable(lambda: ops.mfm_nce_loss(self.logits[:, :Tv], self.clean.vmask, self.vlabels, self.scratch, self.nce_loss,
                                                  self.logits[:, :Tv]), sm)
        if cx.bf:
            fwd.add_callable(lambda: ops.cast_bf16(self.logits, self.dlogits_ct), sm)

    def build_backward(self, bwd, gs, gout):
        cx, run, fl, dt, n, sm = self.cx, self.run, self.cx.fl, self.cx.dt, self.VN, self.run.sm
        W32, G = fl.w32, fl.g
        B, W, F, Tv, D, ldl = self.B, self.W, self.F, self.Tv, self.D, self.ldl
        dl = self.dlogits_ct
        bwd.add_callable(lambda: ops.scale_ct(dl, gout), sm)
        # d scores = dlogits . video ;  d video += dlogits^T . scores
        bwd.add("univl_gemm", _gemm_desc(dt, dl, ldl, self.clean.vn_op, D, Tv, D, Tv, trans_b=1, out16=self.dscores, ldc=D), sm)
        bwd.add("univl_gemm", _gemm_desc(dt, dl, ldl, self.scores, D, Tv, D, Tv, trans_a=1, trans_b=1, out32=self.clean.dvnorm, ldc=D,
                                         accumulate=True), sm)
        bwd.add_callable(lambda: ops.colsum(self.dscores, G(n["bias"])), sm)
        # scores = h . W_vis: dh = dscores . W_vis^T ; dW_vis += h^T . dscores
        bwd.add("univl_gemm", _gemm_desc(dt, self.dscores, D, fl.wop(n["vw"]), D, Tv, H, D, out32=self.dh, ldc=H), sm)
        bwd.add("univl_gemm", _gemm_desc(dt, self.h16, H, self.dscores, D, H, D, Tv, trans_a=1, trans_b=1, out32=G(n["vw"]), ldc=D,
                                         accumulate=gs.acc(n["vw"])), sm)
        bwd.add("univl_layernorm_bwd", ops.layernorm_desc(dt, Tv, H, gamma=W32(n["lg"]), y=self.hy, stats=self.hst, dout=self.dh,
                                                          dx32=self.dg, dgamma=G(n["lg"]), dbeta=G(n["lb"])), sm)
        bwd.add_callable(lambda: ops.gelu_bwd(self.dg, self.u, self.du), sm)
        bwd.add("univl_gemm", _gemm_desc(dt, self.du, H, self.vpart16, H, H, H, Tv, trans_a=1, trans_b=1, out32=G(n["tw"]), ldc=H,
                                         accumulate=gs.acc(n["tw"]), dbias=G(n["tb"])), sm)
        bwd.add("univl_gemm", _gemm_desc(dt, self.du, H, fl.wop(n["tw"]), H, Tv, H, H, trans_b=1, out32=self.dvpart, ldc=H), sm)
        self.mlm.build_backward(bwd, gs, gout, self.tpart16, self.dtpart, accumulate_dx=False)
        # d cross_output = cat(d text part, d video part): written in full (this run has no other consumer)
        bwd.add_callable(lambda: ops.pair_concat_fwd(self.dtpart, self.dvpart, run.enc.amask, run.enc.vmask, self.arange, self.arange,
                                                     B, W, F, run.dcross, None), sm)


# ------------------------------------------------------------------------------------------------ step assembly
class Step:
    """fwd / bwd plans + the buffers UniVL.forward fills for one (rows, max_words, max_frames, training) signature."""

    def __init__(self, cx):
        self.cx = cx
        self.loss, self.gout = cx.e(1), cx.e(1)
        self.gout.fill_(1.0)         # invariant: holds 1.0 between backwards (UniVL._run_backward restores it after an explicit gout)
        self.fwd = Plan()
        self.bwd = {}
        self.loss_terms = []
        self.calls = 0               # training forwards through this step (UniVL._run_plan: eager first, graphs later)

    def finish_forward(self):
        self.fwd.wait_point("all")               # nothing after the forward may overtake an optimizer update in flight
        terms = self.loss_terms
        if len(terms) == 1:
            self.loss = terms[0]         # the loss kernel's own output buffer IS the step's loss: no copy node
        else:
            def add():
                torch.add(terms[0], terms[1], out=self.loss)
                for t in terms[2:]:
                    self.loss.add_(t)
            self.fwd.add_callable(add)

    def backward_plan(self, fresh):
        if fresh not in self.bwd:
            self.bwd[fresh] = self._build_bwd(fresh)
        return self.bwd[fresh]


def _ddp_hook(cx, fl, model, kind):
    """Gradient exchange points for the data-parallel path (univl_amd.parallel): the hook is called by every encoder
    stack after each layer (backward order) and emits an eager all-reduce once UNIVL_BUCKET_MB of gradients are
    pending."""
    red = cx.red
    if red is None:
        return None, None, None
    from .parallel import BucketSchedule, layer_buckets
    buckets = layer_buckets(fl, model.used_parameter_names(kind))
    sched = BucketSchedule(float(os.environ.get("UNIVL_BUCKET_MB", "80")) * 2 ** 20)

    # (Round 4, measured and removed: the per-tensor gradient norms of every exchanged bucket taken on the communication stream right
    # behind its all-reduce, instead of one streaming pass over the gradients after the join.  The data-parallel schedule on one GPU
    # went from 2.56 to 3.16 ms per step at 4 pairs and from 3.87 to 4.60 at 16 (profiles/r04c_dp_bucket_norms_bf16_exchange.txt): a
    # second stream that fills the chip with 2600-workgroup streaming kernels delays every kernel of the latency-bound backward chain
    # by more than the 0.1 ms pass it hides -- the same lesson as the side-stream optimizer update of round 2.)
    def emit(plan):
        ranges = sched.take()
        if ranges:
            # host-issued between captured segments (torch's process group), or -- with a library-held RCCL communicator
            # (parallel.BucketReducer.enable_capture) -- an ordinary, capturable part of the plan
            if red.capturable:
                plan.add_callable(lambda streams: red.reduce_ranges(ranges, after=streams), with_streams=True)
            else:
                plan.add_callable(lambda: red.reduce_ranges(ranges), eager=True)

    def hook(plan, prefix, l, stream):
        key = (prefix, l)
        if key in buckets["layers"] and sched.add(*buckets["layers"][key]):
            emit(plan)
    return hook, buckets, (sched, emit)


def build_step(model, kind, B, W, F, training):
    cx = Ctx(model, training)
    fl, tc = cx.fl, cx.tc
    st = Step(cx)
    st.kind, st.B, st.W, st.F = kind, B, W, F
    fwd = st.fwd
    fwd.external = model._param_events          # events of an optimizer update still in flight (univl_amd.graphed)
    fwd.wait_point("base")                      # embedding tables, every vector, the matrices outside the layer stacks
    cx.stamp(fwd, "f_begin")
    if cx.p > 0:
        fwd.add_callable(lambda: ops.bump_counter(cx.seed_dev))
    st.enc = enc = EncoderPass(cx, B, W, F, normalized_input=(kind == "features_shaped"))
    # Round 5: the first layer of the cross encoder / the decoder is read after both encoder stacks; while an update rides, its chunks go
    # with the products of the text / video stack's LAST layer (which carry nothing otherwise) instead of a launch in front of the forward
    if _ab.get("tail_ride"):
        if kind in ("align", "caption", "pretrain", "pretrain_nocap") and model.cross is not None:
            enc.text.tail_key = ("layer", "cross", 0)
        if kind in ("caption", "pretrain") and model.decoder is not None:
            enc.vis.tail_key = ("layer", "decoder", 0)
    enc.build_forward(fwd)
    st.fwd_encoders_len = len(fwd)
    stage_two = bool(model._stage_two)
    loss_kind = "crossen" if stage_two else ("milnce" if tc.use_mil else "maxmargin")
    pre_kind = "milnce" if tc.use_mil else "maxmargin"          # _pretrain_sim_loss_fct (modeling.py:179-184)
    st.enc_m = st.joint = st.run_pairs = st.pooler = st.run_rows = st.decoder = st.heads = st.run_heads = None
    pairs = [(i, j) for i in range(B) for j in range(B)]
    rows = list(range(B))
    if kind == "joint":
        st.joint = JointSim(cx, enc, loss_kind)
        st.joint.build_forward(fwd)
        st.loss_terms.append(st.joint.loss)
    elif kind == "align":
        st.run_pairs = CrossRun(cx, enc, [a for a, _ in pairs], [b for _, b in pairs])
        st.run_pairs.build_forward(fwd)
        st.pooler = PoolerSim(cx, st.run_pairs, B, B, loss_kind)
        st.pooler.build_forward(fwd)
        st.loss_terms.append(st.pooler.loss)
    elif kind == "caption":
        st.run_rows = CrossRun(cx, enc, rows, rows)
        st.run_rows.build_forward(fwd)
        st.decoder = DecoderRun(cx, st.run_rows, W)
        st.decoder.build_forward(fwd)
        st.loss_terms.append(st.decoder.loss)
    elif kind in ("pretrain", "pretrain_nocap"):
        st.enc_m = enc_m = EncoderPass(cx, B, W, F)            # masked text / masked video pass (modeling.py:221)
        enc_m.build_forward(fwd)
        st.run_heads = CrossRun(cx, enc_m, rows, rows)
        st.run_heads.build_forward(fwd)
        st.heads = PretrainHeads(cx, st.run_heads, enc)
        st.heads.build_forward(fwd)
        st.loss_terms += [st.heads.mlm.loss, st.heads.nce_loss]
        st.joint = JointSim(cx, enc, pre_kind)                  # _pretrain_joint on the CLEAN outputs (:233-236)
        st.joint.build_forward(fwd)
        st.loss_terms.append(st.joint.loss)
        if kind == "pretrain":                                   # modeling.py:238-254: only when captions are given
            st.run_rows = CrossRun(cx, enc_m, rows, rows)        # _get_decoder_score re-runs the cross encoder (:404)
            st.run_rows.build_forward(fwd)
            st.decoder = DecoderRun(cx, st.run_rows, W)
            st.decoder.build_forward(fwd)
            st.loss_terms.append(st.decoder.loss)
        st.run_pairs = CrossRun(cx, enc_m, [a for a, _ in pairs], [b for _, b in pairs])   # alignment (:258-267)
        st.run_pairs.build_forward(fwd)
        st.pooler = PoolerSim(cx, st.run_pairs, B, B, loss_kind)
        st.pooler.build_forward(fwd)
        st.loss_terms.append(st.pooler.loss)
    elif kind in ("features", "features_shaped"):
        pass                                   # encoders only (get_sequence_visual_output on a stage-two model)
    else:
        raise ValueError(kind)
    cx.stamp(fwd, "f_end")
    if st.loss_terms:
        st.finish_forward()
    else:
        fwd.wait_point("all")

    def build_bwd(fresh):
        bwd = Plan()
        # Sparse exchange of the word-embedding gradient (94 MB dense, <= B*W non-zero rows): only where the token
        # gather is the table's sole gradient source (no tied decoder / MLM head in this step).
        sparse = (cx.red is not None and kind in ("joint", "align") and bool(_ab.get("sparse_emb")))
        enc.sparse_word_grad = sparse
        # gradient norms from the wgrad epilogues: single-GPU only (after an all-reduce the local sums are not the
        # norms of the averaged gradients) and only where every matrix has one writer per backward
        fuse = (cx.red is None and kind in ("joint", "align", "caption") and bool(_ab.get("fused_norms")))
        gs = GradState(fl, fresh, fuse_sumsq=fuse)
        hook, buckets, sched = _ddp_hook(cx, fl, model, kind)
        # A layer reports its gradient slice to the exchange schedule only after its LAST writer in this backward: the
        # cross stack runs up to three times (pair similarity, decoder rows, pretrain heads) and the text / video stacks
        # twice on the pretrain path (masked pass, clean pass), every pass accumulating into the same slice.  An
        # all-reduce issued after an earlier pass would race with the later read-modify-write weight gradients.
        cross_runs = [r for r in (st.run_pairs, st.run_rows, st.run_heads) if r is not None]     # emission order below
        last_cross = cross_runs[-1] if cross_runs else None
        hook_for = lambda r: hook if r is last_cross else None
        # Everything a backward clears before its kernels add into it goes into ONE launch.  Retrieval configurations
        # (the token gather is the word table's only gradient source): the 94 MB table is not cleared as a whole, only the
        # rows the previous backward wrote (engine.FlatParams.word_rows).
        world = 1 if cx.red is None else max(1, cx.red.world)
        rows_mode = (kind in ("joint", "align") and bool(_ab.get("sparse_rows"))
                     and (cx.red is None or sparse)        # a dense all-reduce of the table fills rows nobody listed
                     and enc.Tt * (world if sparse else 1) <= fl.WORD_ROWS_CAP)
        zeros = [fl.sumsq, fl.partials] if fuse else []
        enc.word_rows = None
        if rows_mode:
            lst, meta = fl.word_rows()
            enc.word_rows = (lst, meta, fresh)
            if fresh:
                zeros += fl.v_region_without_word_table()
                bwd.add_callable(lambda: ops.rows_zero(fl.g(fl.WORD), lst, meta))
        elif fresh:
            zeros.append(fl.g32[:fl.v_end])
        zeros += enc.zero_list()
        if st.enc_m is not None:
            zeros += st.enc_m.zero_list()
        for r in (st.run_pairs, st.run_rows):
            if r is not None:
                zeros += r.zero_list()
        if st.run_heads is not None:
            zeros.append(st.run_heads.dpostype)
        cx.stamp(bwd, "b_begin")
        bwd.add_zeros(zeros)
        g = st.gout
        if st.pooler is not None:
            st.pooler.build_backward(bwd, gs, g)
            st.run_pairs.build_backward(bwd, gs, hook_for(st.run_pairs))
        if st.decoder is not None:
            st.decoder.build_backward(bwd, gs, g)
            st.run_rows.build_backward(bwd, gs, hook_for(st.run_rows))
        if st.heads is not None:
            st.heads.build_backward(bwd, gs, g)
            st.run_heads.build_backward(bwd, gs, hook_for(st.run_heads))
        if st.joint is not None:
            st.joint.build_backward(bwd, g)
        if st.enc_m is not None:
            st.enc_m.build_backward(bwd, gs, None)          # the clean pass below is the last writer of bert.* / visual.*
        enc.build_backward(bwd, gs, hook)
        if cx.red is not None:
            red, tail = cx.red, buckets["tail"]
            if sparse:
                from .parallel import subtract_range
                wname = EncoderPass.N["bw"]
                wo, wk, _ = fl.index[wname]
                tail = subtract_range(tail, wo, wo + (wk + 63) // 64 * 64)
                world = max(1, red.world)
                ids_all = torch.zeros(world, enc.Tt, dtype=torch.int64, device=cx.dev)
                rows_all = torch.zeros(world, enc.Tt, H, device=cx.dev)
                st.sparse_exchange = dict(tokens=enc.Tt, bytes=rows_all.numel() * 4 + ids_all.numel() * 8)

                def gather_tokens():
                    red.gather(enc.ids.view(-1), ids_all)
                    red.gather(enc.drows, rows_all)
                if red.capturable:
                    bwd.add_callable(lambda streams: (red.gather(enc.ids.view(-1), ids_all, after=streams),
                                                      red.gather(enc.drows, rows_all)), with_streams=True)
                else:
                    bwd.add_callable(gather_tokens, eager=True)
            for (s0, e0) in tail:
                sched[0].add(s0, e0)
            sched[1](bwd)                                  # whatever is still pending + the tail
            bwd.add_callable(red.join, eager=not red.capturable)
            if sparse:                                     # rebuild the dense table gradient: mean over ranks
                bwd.add_callable(lambda: ops.embed_scatter(ids_all.view(-1), rows_all.view(-1, H), 1.0 / world, fl.g(wname)))
                if rows_mode:
                    bwd.add_callable(lambda: ops.rows_append(ids_all.view(-1), lst, meta, fresh, fl.word_ever))
            st.exchange_points = list(sched[0].cuts)
        gs.finish(bwd)
        cx.stamp(bwd, "b_end")
        bwd.fused_names = frozenset(gs.covered)
        bwd.rows_mode = rows_mode
        return bwd

    st._build_bwd = build_bwd
    return st
