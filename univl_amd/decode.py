"""Beam-search caption decoding with cached state.

The reference's caption evaluation (main_task_caption.py:434-618 with modules/beam.py) calls
`model.decoder_caption(...)` once per generated token on the COMPLETE prefixes of all n_inst x 5 beams: the 2-layer
cross encoder over cat(text, video) and the 3-layer decoder are recomputed from scratch up to max_words times, the
beams are re-assembled on the host with one `.item()` per token, and only the last position's logits are used
(`dec_output[:, -1, :]`, :452).  In eval mode everything that does not depend on the newest token is a pure function
of earlier inputs, so here:

  * the cross encoder runs ONCE per instance (not per beam, not per step) and each decoder layer's encoder-attention
    K/V projections of its output are computed once;
  * the decoder keeps a key/value cache per beam: a step embeds one token per beam, projects q/k/v for that position
    only, attends over the cache (attention kernel with Sq = 1 and a cache batch stride), and the five beams of an
    instance share the instance's encoder K/V by running encoder attention as B = n_inst, Sq = n_beams;
  * beam bookkeeping (Beam.advance, beam.py:63-87) stays on the device: top-k over the flattened (beam x vocab)
    log-probabilities, back-pointers = id // vocab, tokens = id % vocab, the cache rows re-ordered by back-pointer with
    one gather per layer; the host reads one "all instances done" flag per step.  Finished instances keep their slots
    (static shapes) and are frozen, which is what removing them (collate_active_info, :404-416) amounts to.

Same results as the reference's procedure on the same logits: first step uses beam 0's distribution only (beam.py:69),
an instance is done when its top beam emits EOS (beam.py:84), the reported hypothesis is the best-scored beam walked
back through the back-pointers (beam.py:108-116, collect_hypothesis_and_scores n_best = 1).
"""
import torch

from . import ops
from .engine import Plan, _gemm_desc
from .steps import Ctx, CrossRun, RowFeatures, VocabHead, H


class CaptionBeamSearch:
    """Compiled decoding session for a fixed (n_inst, max_words W, max_frames F, beam size, max_len)."""

    NH, I = 12, 3072

    def __init__(self, model, n_inst, W, F, n_bm=5, max_len=None, use_graphs=True):
        if model.decoder is None:
            raise RuntimeError("CaptionBeamSearch: this model was built without a decoder (stage one)")
        self.model, self.n_inst, self.W, self.F, self.n_bm = model, n_inst, W, F, n_bm
        self.use_graphs = bool(use_graphs)     # each step plan (one per position) is captured once into a hipGraph
        self.Tmax = Tmax = int(max_len or model.task_config.max_words)
        assert Tmax <= model.decoder_config.max_target_embeddings
        model.flat.refresh_shadow()
        cx = self.cx = Ctx(model, False)
        e, ct, bf, fl, dt = cx.e, cx.ct, cx.bf, cx.fl, cx.dt
        self.R = R = n_inst * n_bm
        self.V = V = model.bert_config.vocab_size
        self.L = L = model.decoder_config.num_decoder_layers
        rows = list(range(n_inst))
        # ---- once per batch: cross encoder per instance + encoder K/V of every decoder layer
        self.feats = RowFeatures(cx, n_inst, n_inst, W, F)
        self.run = run = CrossRun(cx, self.feats, rows, rows)
        self.S = S = run.S
        self.setup = Plan()
        run.build_forward(self.setup)
        from .engine import DecoderStack
        names = DecoderStack._names
        self.nm = [names(None, l) for l in range(L)]
        self.kv2 = [e(n_inst * S, 2 * H, dtype=ct) for _ in range(L)]
        for l in range(L):
            nm = self.nm[l]
            self.setup.add("univl_gemm", _gemm_desc(dt, run.out16, H, fl.wop_fused(nm["c_kv_w"]), H, n_inst * S, 2 * H, H,
                                                    out16=self.kv2[l], ldc=2 * H, bias=fl.w32_fused(nm["c_kv_b"])))
        # ---- per step buffers (R rows)
        self.ids = e(R, dtype=torch.int64)
        self.ey, self.est, self.e32 = e(R, H), e(R, 2), e(R, H)
        self.e16 = e(R, H, dtype=ct) if bf else self.e32
        self.cache = [[torch.zeros(R, Tmax, 2 * H, device=cx.dev, dtype=ct) for _ in range(2)] for _ in range(L)]
        self.ws = []
        for l in range(L):
            w = dict(q1=e(R, H, dtype=ct), ctx1=e(R, H, dtype=ct), lse1=e(R * self.NH), y1=e(R, H), st1=e(R, 2), a32=e(R, H),
                     q2=e(R, H, dtype=ct), ctx2=e(R, H, dtype=ct), lse2=e(R * self.NH), y2=e(R, H), st2=e(R, 2), d32=e(R, H),
                     u=e(R, self.I, dtype=ct), f=e(R, self.I, dtype=ct), y3=e(R, H), st3=e(R, 2), o32=e(R, H))
            for k in ("a", "d", "o"):
                w[k + "16"] = e(R, H, dtype=ct) if bf else w[k + "32"]
            self.ws.append(w)
        self.head = VocabHead(cx, "decoder.classifier.cls.predictions", R)
        self.src = e(R, dtype=torch.int32)               # cache row each beam continues from
        self.steps = {}
        self.base = torch.arange(n_inst, device=cx.dev, dtype=torch.int64)[:, None] * n_bm

    # ------------------------------------------------------------------------------------------ step plans
    def _step_plan(self, t):
        pl = self.steps.get(t)
        if pl is not None:
            return pl
        cx, fl, dt, R, Tmax, S = self.cx, self.cx.fl, self.cx.dt, self.R, self.Tmax, self.S
        W32 = fl.w32
        es = 2 if cx.bf else 4
        pl = Plan()
        cur, prev = t % 2, (t + 1) % 2
        pos = W32("bert.embeddings.position_embeddings.weight")[t:]
        pl.add("univl_embed_text_fwd", ops.embed_text_desc(
            dt, R, 1, self.ids, W32("bert.embeddings.word_embeddings.weight"), pos, W32("decoder.embeddings.LayerNorm.weight"),
            W32("decoder.embeddings.LayerNorm.bias"), y=self.ey, stats=self.est, out32=self.e32,
            out16=self.e16 if cx.bf else None))
        x32, x16 = self.e32, self.e16
        for l in range(self.L):
            nm, ws = self.nm[l], self.ws[l]
            cache = self.cache[l][cur]
            if t > 0:        # beams continue from re-ordered parents: gather positions [0, t) of the parent rows
                src_c, n_rows, stride, nbytes = self.cache[l][prev], R, Tmax * 2 * H * es, t * 2 * H * es
                pl.add_callable(lambda s=src_c, d=cache, st=stride, nb=nbytes: ops.gather_rows(s, d, self.src, R, st, nb))
            wqkv, bqkv = fl.wop_fused(nm["s_qkv_w"]), fl.w32_fused(nm["s_qkv_b"])
            pl.add("univl_gemm", _gemm_desc(dt, x16, H, wqkv[:H], H, R, H, H, out16=ws["q1"], ldc=H, bias=bqkv[:H]))
            kv_slot = cache[:, t]                                      # [R, 2H] view, row stride Tmax*2H
            pl.add("univl_gemm", _gemm_desc(dt, x16, H, wqkv[H:], H, R, 2 * H, H, out16=kv_slot, ldc=Tmax * 2 * H, bias=bqkv[H:]))
            pl.add("univl_attention_fwd", ops.attention_desc(
                dt, R, self.NH, 1, t + 1, ws["q1"], H, (cache, 0), 2 * H, (cache, H), 2 * H, ws["ctx1"], H, ws["lse1"],
                bsk=Tmax * 2 * H, bsv=Tmax * 2 * H))
            pl.add("univl_gemm", _gemm_desc(dt, ws["ctx1"], H, fl.wop(nm["s_o_w"]), H, R, H, H, out32=ws["y1"], ldc=H,
                                            bias=W32(nm["s_o_b"])))
            pl.add("univl_layernorm_fwd", ops.layernorm_desc(
                dt, R, H, x=ws["y1"], residual=x32, gamma=W32(nm["s_ln_g"]), beta=W32(nm["s_ln_b"]), y=ws["y1"], stats=ws["st1"],
                out32=ws["a32"], out16=ws["a16"] if cx.bf else None))
            pl.add("univl_gemm", _gemm_desc(dt, ws["a16"], H, fl.wop(nm["c_q_w"]), H, R, H, H, out16=ws["q2"], ldc=H,
                                            bias=W32(nm["c_q_b"])))
            kv = self.kv2[l]
            pl.add("univl_attention_fwd", ops.attention_desc(
                dt, self.n_inst, self.NH, self.n_bm, S, ws["q2"], H, (kv, 0), 2 * H, (kv, H), 2 * H, ws["ctx2"], H, ws["lse2"],
                key_mask=self.run.cmask))
            pl.add("univl_gemm", _gemm_desc(dt, ws["ctx2"], H, fl.wop(nm["c_o_w"]), H, R, H, H, out32=ws["y2"], ldc=H,
                                            bias=W32(nm["c_o_b"])))
            pl.add("univl_layernorm_fwd", ops.layernorm_desc(
                dt, R, H, x=ws["y2"], residual=ws["a32"], gamma=W32(nm["c_ln_g"]), beta=W32(nm["c_ln_b"]), y=ws["y2"],
                stats=ws["st2"], out32=ws["d32"], out16=ws["d16"] if cx.bf else None))
            pl.add("univl_gemm", _gemm_desc(dt, ws["d16"], H, fl.wop(nm["w1"]), H, R, self.I, H, out16=ws["f"], ldc=self.I,
                                            bias=W32(nm["b1"]), aux=ws["u"], ldaux=self.I, gelu="fwd"))
            pl.add("univl_gemm", _gemm_desc(dt, ws["f"], self.I, fl.wop(nm["w2"]), self.I, R, H, self.I, out32=ws["y3"], ldc=H,
                                            bias=W32(nm["b2"])))
            pl.add("univl_layernorm_fwd", ops.layernorm_desc(
                dt, R, H, x=ws["y3"], residual=ws["d32"], gamma=W32(nm["ln_g"]), beta=W32(nm["ln_b"]), y=ws["y3"], stats=ws["st3"],
                out32=ws["o32"], out16=ws["o16"] if cx.bf else None))
            x32, x16 = ws["o32"], ws["o16"]
        self.head.build_forward(pl, x16, with_loss=False)
        pl.add_callable(lambda: ops.log_softmax_rows(self.head.logits, self.V))
        pl.keepalive = (pos,)
        self.steps[t] = pl
        return pl

    # ------------------------------------------------------------------------------------------------- run
    @torch.no_grad()
    def step_logprobs(self, t, last_tokens, parents=None):
        """One cached decoder step: log-probabilities [R, V] of the token after position t, given each beam's token at
        position t and (t > 0) the cache row it continues from.  Exposed for the parity tests."""
        self.ids.copy_(last_tokens.reshape(-1))
        if t > 0:
            self.src.copy_(parents.reshape(-1).to(torch.int32))
        pl = self._step_plan(t)
        if self.use_graphs and not torch.cuda.is_current_stream_capturing():
            pl.run_graphed()
        else:
            pl.run()
        return self.head.logits[:, :self.V]

    @torch.no_grad()
    def encode(self, sequence_output, visual_output, input_mask, video_mask):
        self.model.flat.refresh_shadow()
        self.feats.load(sequence_output.to(torch.float32), visual_output.to(torch.float32),
                        input_mask.reshape(-1, input_mask.shape[-1]), video_mask.reshape(-1, video_mask.shape[-1]))
        self.setup.run()

    @torch.no_grad()
    def __call__(self, sequence_output, visual_output, input_mask, video_mask, bos, eos, max_len=None):
        """Returns (hypotheses: list of n_inst token lists, as collect_hypothesis_and_scores(n_best=1) gives them,
        scores: [n_inst] fp32 tensor of the best beams' accumulated log-probabilities)."""
        n, nb, V, dev = self.n_inst, self.n_bm, self.V, self.cx.dev
        max_len = min(int(max_len or self.Tmax), self.Tmax)
        self.encode(sequence_output, visual_output, input_mask, video_mask)
        scores = torch.zeros(n, nb, device=dev)
        done = torch.zeros(n, dtype=torch.bool, device=dev)
        length = torch.zeros(n, dtype=torch.int64, device=dev)
        tokens = torch.full((n, nb), int(bos), dtype=torch.int64, device=dev)
        parents = torch.arange(nb, device=dev, dtype=torch.int64).expand(n, nb).contiguous()
        prev_ks, next_ys = [], []
        for t in range(max_len):
            lp = self.step_logprobs(t, tokens, (self.base + parents) if t > 0 else None).view(n, nb, V)
            if t == 0:
                best, ids = lp[:, 0, :].topk(nb, dim=1, largest=True, sorted=True)             # beam.py:69-72
            else:
                best, ids = (lp + scores[:, :, None]).view(n, nb * V).topk(nb, dim=1, largest=True, sorted=True)
            pk, ny = ids // V, ids % V
            act = ~done
            scores = torch.where(act[:, None], best, scores)
            parents = torch.where(act[:, None], pk, torch.arange(nb, device=dev).expand(n, nb))
            tokens = torch.where(act[:, None], ny, tokens)
            prev_ks.append(pk)
            next_ys.append(ny)
            length += act.to(torch.int64)
            done = done | (act & (ny[:, 0] == int(eos)))                                      # beam.py:84
            if bool(done.all()):
                break
        # hypotheses: best beam (index 0 after the sorted top-k) walked back through the back-pointers (beam.py:108-116)
        pks, nys = torch.stack(prev_ks).cpu(), torch.stack(next_ys).cpu()
        lens = length.cpu().tolist()
        hyps = []
        for i in range(n):
            k, hyp = 0, []
            for j in range(lens[i] - 1, -1, -1):
                hyp.append(int(nys[j, i, k]))
                k = int(pks[j, i, k])
            hyps.append(hyp[::-1])
        return hyps, scores[:, 0].clone()
