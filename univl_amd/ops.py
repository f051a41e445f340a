"""Tensor-level wrappers over the C ABI (one call = one enqueue on torch's current HIP stream).

PyTorch is used here only for device memory and streams.  Every function fills the C descriptor from tensor
pointers/strides and raises RuntimeError on a non-zero status.  The execution plans in `univl_amd.engine` build the
same descriptors once and replay them; these wrappers are the eager form (unit tests, small host-side ops)."""
import ctypes as C

import torch

from . import _lib
from ._lib import DT_BF16, DT_F32

_BYREF = C.byref


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def dtype_code(dtype):
    if dtype == torch.float32:
        return DT_F32
    if dtype == torch.bfloat16:
        return DT_BF16
    raise RuntimeError("univl_amd: unsupported compute dtype %s (float32 or bfloat16)" % dtype)


def _require_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("univl_amd kernels need HIP device tensors (got %s); there is no CPU fallback" % t.device)


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D view"
    return t.stride(0)


def probe_layouts():
    out = torch.zeros(1792, device="cuda", dtype=torch.float32)
    _lib.check(_lib.lib().univl_probe_layouts(_p(out), 1792, _stream()), "probe_layouts")
    return out


def gemm_desc(A, B, M, N, K, *, trans_a=False, trans_b=False, out32=None, out16=None, bias=None, residual=None, aux=None,
              gelu=None, accumulate=False, dbias=None, dbias_atomic=False, ksplit=1, tile=0, alpha=1.0, sumsq=None,
              sumsq_rows=0, sumsq_stride=0, stages=0, waves=0, aux_f32=False, a_lo=None, b_lo=None, out16_lo=None):
    """a_lo / b_lo / out16_lo: the lo halves of operand pairs (include/univl_hip.h: UnivlGemm.A_lo), same shape and strides as A / B / out16."""
    _require_gpu(A, B, out32, out16, a_lo, b_lo, out16_lo)
    d = _lib.Gemm()
    d.dtype = dtype_code(A.dtype)
    assert B.dtype == A.dtype
    d.trans_a, d.trans_b = int(trans_a), int(trans_b)
    d.M, d.N, d.K = M, N, K
    d.A, d.lda, d.B, d.ldb = _p(A), _ld(A), _p(B), _ld(B)
    d.C32, d.C16 = _p(out32), _p(out16)
    d.ldc = _ld(out32) if out32 is not None else _ld(out16)
    if out32 is not None and out16 is not None:
        assert _ld(out32) == _ld(out16)
    d.bias = _p(bias)
    d.R, d.ldr = _p(residual), (_ld(residual) if residual is not None else 0)
    d.aux, d.ldaux = _p(aux), (_ld(aux) if aux is not None else 0)
    d.dbias = _p(dbias)
    d.alpha = alpha
    flags = 0
    if accumulate:
        flags |= _lib.GEMM_ACCUM
    if gelu == "fwd":
        flags |= _lib.GEMM_GELU_FWD
    elif gelu == "bwd":
        flags |= _lib.GEMM_GELU_BWD
    if dbias_atomic:
        flags |= _lib.GEMM_DBIAS_ATOMIC
    if aux_f32:
        flags |= _lib.GEMM_AUX_F32
    d.flags, d.ksplit, d.tile = flags, ksplit, tile
    d.sumsq, d.sumsq_rows, d.sumsq_stride = _p(sumsq), sumsq_rows, sumsq_stride
    d.stages, d.waves = int(stages), int(waves)
    for lo, hi in ((a_lo, A), (b_lo, B), (out16_lo, out16)):
        assert lo is None or (lo.dtype == hi.dtype and lo.stride() == hi.stride())
    d.A_lo, d.B_lo, d.C16_lo = _p(a_lo), _p(b_lo), _p(out16_lo)
    return d


def gemm(A, B, M, N, K, **kw):
    """C[M,N] = epi(alpha * A_op . B_op^T); A/B are 2-D row-major views ([rows,K] or, if trans_x, [K,rows])."""
    d = gemm_desc(A, B, M, N, K, **kw)
    _lib.check(_lib.lib().univl_gemm(_BYREF(d), _stream()), "gemm")


def gemm_group(descs, max_blocks=0):
    """Independent GEMMs (same dtype and operand layouts, at most GEMM_GROUP_MAX) in one launch; max_blocks > 0 caps the
    grid (the kernel walks its tiles)."""
    arr = (_lib.Gemm * len(descs))(*descs)
    _lib.check(_lib.lib().univl_gemm_group_limited(arr, len(descs), int(max_blocks), _stream()), "gemm_group")


def gemm_pair(dgrad, wgrad, dry_run=False):
    """One launch for a dgrad product and the weight-gradient product fed by the same upstream gradient
    (univl_gemm_pair).  Returns False when the C side does not take the pair (not bf16 / not on the 64 tile)."""
    rc = _lib.lib().univl_gemm_pair(_BYREF(dgrad), _BYREF(wgrad), int(bool(dry_run)), _stream())
    if rc == -3:
        return False
    _lib.check(rc, "gemm_pair")
    return True


def gemm_ln(gemm, ln, counters, dry_run=False):
    """A forward product and the LayerNorm that consumes its fp32 output in ONE launch (univl_gemm_ln).  Returns False when the C side
    does not carry the pair (deterministic mode, other tiles / layouts); counters: int32 [2 * ceil(M / 64)], zero."""
    rc = _lib.lib().univl_gemm_ln(_BYREF(gemm), _BYREF(ln), C.c_void_p(counters.data_ptr()), None, 0, 0, 0, int(bool(dry_run)), _stream())
    if rc == _lib.EUNSUPPORTED:
        return False
    _lib.check(rc, "gemm_ln")
    return True


def gemm_pair_ln(dgrad, wgrad, ln, counters, dry_run=False):
    """univl_gemm_pair with the LayerNorm backward fed by the dgrad's fp32 output finished inside the launch (univl_gemm_pair_ln)."""
    rc = _lib.lib().univl_gemm_pair_ln(_BYREF(dgrad), _BYREF(wgrad), _BYREF(ln), C.c_void_p(counters.data_ptr()), int(bool(dry_run)), _stream())
    if rc == _lib.EUNSUPPORTED:
        return False
    _lib.check(rc, "gemm_pair_ln")
    return True


def layernorm_desc(dtype, rows, N, *, x=None, x_f64=False, residual=None, pos=None, pos_period=0, gamma=None,
                   beta=None, eps=1e-12, y=None, stats=None, out32=None, out16=None, p_pre=0.0, p_post=0.0, seed=0,
                   off_pre=0, off_post=0, seed_dev=None, dout=None, dx32=None, dxd32=None, dxd16=None, dgamma=None,
                   dbeta=None, dbias=None, dpos=None, out16_lo=None):
    d = _lib.LayerNorm()
    d.dtype, d.rows, d.N, d.x_f64 = dtype, rows, N, int(x_f64)
    d.x, d.residual, d.pos, d.pos_period = _p(x), _p(residual), _p(pos), pos_period
    d.gamma, d.beta, d.eps = _p(gamma), _p(beta), eps
    d.y, d.stats, d.out32, d.out16 = _p(y), _p(stats), _p(out32), _p(out16)
    d.p_pre, d.p_post, d.seed, d.off_pre, d.off_post, d.seed_dev = p_pre, p_post, seed, off_pre, off_post, _p(seed_dev)
    d.dout, d.dx32, d.dxd32, d.dxd16 = _p(dout), _p(dx32), _p(dxd32), _p(dxd16)
    d.dgamma, d.dbeta, d.dbias, d.dpos = _p(dgamma), _p(dbeta), _p(dbias), _p(dpos)
    d.out16_lo = _p(out16_lo)
    return d


def layernorm_fwd(**kw):
    d = layernorm_desc(**kw)
    _lib.check(_lib.lib().univl_layernorm_fwd(_BYREF(d), _stream()), "layernorm_fwd")


def layernorm_bwd(**kw):
    d = layernorm_desc(**kw)
    _lib.check(_lib.lib().univl_layernorm_bwd(_BYREF(d), _stream()), "layernorm_bwd")


def attention_desc(dtype, B, H, Sq, Sk, q, ldq, k, ldk, v, ldv, out, ldo, lse, *, key_mask=None, causal=False,
                   p_drop=0.0, seed=0, offset=0, seed_dev=None, dout=None, lddo=0, dq=None, lddq=0, dk=None, lddk=0,
                   dv=None, lddv=0, bsk=0, bsv=0, out_lo=None):
    """q/k/v/out/... are (tensor, element_offset) pairs or tensors; ld in elements."""
    def ptr(t):
        if t is None:
            return None
        if isinstance(t, tuple):
            return C.c_void_p(t[0].data_ptr() + t[1] * t[0].element_size())
        return C.c_void_p(t.data_ptr())
    d = _lib.Attention()
    d.dtype, d.B, d.H, d.Sq, d.Sk = dtype, B, H, Sq, Sk
    d.q, d.ldq, d.k, d.ldk, d.v, d.ldv = ptr(q), ldq, ptr(k), ldk, ptr(v), ldv
    d.key_mask, d.causal = _p(key_mask), int(causal)
    d.out, d.ldo, d.lse = ptr(out), ldo, _p(lse)
    d.p_drop, d.seed, d.offset, d.seed_dev = p_drop, seed, offset, _p(seed_dev)
    d.dout, d.lddo, d.dq, d.lddq, d.dk, d.lddk, d.dv, d.lddv = ptr(dout), lddo, ptr(dq), lddq, ptr(dk), lddk, ptr(dv), lddv
    d.bsk, d.bsv = bsk, bsv
    d.out_lo = ptr(out_lo)
    return d


def attention_fwd(*a, **kw):
    d = attention_desc(*a, **kw)
    _lib.check(_lib.lib().univl_attention_fwd(_BYREF(d), _stream()), "attention_fwd")


def attention_bwd(*a, **kw):
    d = attention_desc(*a, **kw)
    _lib.check(_lib.lib().univl_attention_bwd(_BYREF(d), _stream()), "attention_bwd")


def attention_bwd_fused(attn, odgrad, owgrad=None, dry_run=False):
    """univl_attention_bwd with the attention-output dgrad that produces its upstream gradient computed inside the launch, the weight
    gradient of that projection optionally riding (univl_attention_bwd_fused).  Returns False where the C side does not carry the pair."""
    rc = _lib.lib().univl_attention_bwd_fused(_BYREF(attn), _BYREF(odgrad), _BYREF(owgrad) if owgrad is not None else None,
                                              int(bool(dry_run)), _stream())
    if rc == _lib.EUNSUPPORTED:
        return False
    _lib.check(rc, "attention_bwd_fused")
    return True


def attention_fwd_fused(attn, qkv, dry_run=False):
    """univl_attention_fwd with the q | k | v projection computed inside the launch (univl_attention_fwd_fused, no riding optimizer
    chunks).  Returns False where the C side does not carry the pair."""
    rc = _lib.lib().univl_attention_fwd_fused(_BYREF(attn), _BYREF(qkv), None, 0, 0, 0, int(bool(dry_run)), _stream())
    if rc == _lib.EUNSUPPORTED:
        return False
    _lib.check(rc, "attention_fwd_fused")
    return True


def embed_text_desc(dtype, B, S, ids, word, pos, gamma, beta, *, type_ids=None, type_emb=None, eps=1e-12, y=None,
                    stats=None, out32=None, out16=None, p_post=0.0, seed=0, off_post=0, seed_dev=None, dout=None,
                    dword=None, dpos=None, dtype_emb=None, dgamma=None, dbeta=None, drows=None, out16_lo=None):
    d = _lib.EmbedText()
    d.dtype, d.B, d.S, d.N = dtype, B, S, 768
    d.ids, d.type_ids = _p(ids), _p(type_ids)
    d.word, d.pos, d.type = _p(word), _p(pos), _p(type_emb)
    d.gamma, d.beta, d.eps = _p(gamma), _p(beta), eps
    d.y, d.stats, d.out32, d.out16 = _p(y), _p(stats), _p(out32), _p(out16)
    d.p_post, d.seed, d.off_post, d.seed_dev = p_post, seed, off_post, _p(seed_dev)
    d.dout, d.dword, d.dpos, d.dtype_emb, d.dgamma, d.dbeta = _p(dout), _p(dword), _p(dpos), _p(dtype_emb), _p(dgamma), _p(dbeta)
    d.drows = _p(drows)
    d.out16_lo = _p(out16_lo)
    return d


def embed_text_fwd(*a, **kw):
    d = embed_text_desc(*a, **kw)
    _lib.check(_lib.lib().univl_embed_text_fwd(_BYREF(d), _stream()), "embed_text_fwd")


def embed_text_bwd(*a, **kw):
    d = embed_text_desc(*a, **kw)
    _lib.check(_lib.lib().univl_embed_text_bwd(_BYREF(d), _stream()), "embed_text_bwd")


def pool_desc(B, S, x, mask, *, skip_first, normalize, mean=None, out=None, dout=None, dx=None, ldx_row=768, accumulate=False,
              dsim=None, other=None, n_other=0, transpose=False, gscale=None):
    """dsim / other / n_other / transpose / gscale: the backward takes its upstream gradient from d loss / d sim instead of dout
    (include/univl_hip.h: UnivlPool.dsim)."""
    d = _lib.Pool()
    d.B, d.S, d.N = B, S, 768
    d.x, d.ldx_row, d.mask = _p(x), ldx_row, _p(mask)
    d.skip_first, d.normalize = int(skip_first), int(normalize)
    d.mean, d.out, d.dout, d.dx = _p(mean), _p(out), _p(dout), _p(dx)
    d.accumulate = int(accumulate)
    d.dsim, d.ldsim = _p(dsim), (dsim.stride(0) if dsim is not None else 0)
    d.other, d.n_other, d.transpose, d.gscale = _p(other), int(n_other), int(bool(transpose)), _p(gscale)
    return d


def pool_fwd(*a, **kw):
    d = pool_desc(*a, **kw)
    _lib.check(_lib.lib().univl_pool_fwd(_BYREF(d), _stream()), "pool_fwd")


def pool_bwd(*a, **kw):
    d = pool_desc(*a, **kw)
    _lib.check(_lib.lib().univl_pool_bwd(_BYREF(d), _stream()), "pool_bwd")


def pool_pair_fwd(da, db):
    _lib.check(_lib.lib().univl_pool_pair_fwd(_BYREF(da), _BYREF(db), _stream()), "pool_pair_fwd")


def pool_pair_bwd(da, db):
    _lib.check(_lib.lib().univl_pool_pair_bwd(_BYREF(da), _BYREF(db), _stream()), "pool_pair_bwd")


def maxmargin_loss(sim, margin, weight, loss, dsim):
    """sim/dsim: [n, ld] row-major views (ld = stride(0) >= n)."""
    n = sim.shape[0]
    assert dsim.stride(0) == sim.stride(0)
    _lib.check(_lib.lib().univl_maxmargin_loss(_p(sim), n, sim.stride(0), margin, _p(weight), _p(loss), _p(dsim), _stream()), "maxmargin")


def crossen_loss(sim, loss, dsim):
    assert dsim.stride(0) == sim.stride(0)
    _lib.check(_lib.lib().univl_crossen_loss(_p(sim), sim.shape[0], sim.stride(0), _p(loss), _p(dsim), _stream()), "crossen")


def milnce_loss(sim, batch_size, n_pair, loss, dsim):
    assert dsim.stride(0) == sim.stride(0)
    _lib.check(_lib.lib().univl_milnce_loss(_p(sim), batch_size, n_pair, sim.stride(0), _p(loss), _p(dsim), _stream()), "milnce")


def scale_by_device_scalar(x, s):
    _lib.check(_lib.lib().univl_scale_by_device_scalar(_p(x), x.numel(), _p(s), _stream()), "scale_by_device_scalar")


def zero_many(tensors):
    """One launch that clears up to 16 device buffers."""
    import ctypes as C
    L = _lib.lib()
    for i in range(0, len(tensors), 16):
        ts = tensors[i:i + 16]
        ptrs = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        sizes = (C.c_int64 * len(ts))(*[t.numel() * t.element_size() for t in ts])
        _lib.check(L.univl_zero_many(ptrs, sizes, len(ts), _stream()), "zero_many")


def copy_many(pairs):
    """One launch for several device-to-device copies: pairs = [(dst, src), ...], same device, contiguous, equal byte sizes."""
    import ctypes as C
    L = _lib.lib()
    for i in range(0, len(pairs), 16):
        ps = pairs[i:i + 16]
        srcs = (C.c_void_p * len(ps))(*[s.data_ptr() for _, s in ps])
        dsts = (C.c_void_p * len(ps))(*[d.data_ptr() for d, _ in ps])
        sizes = (C.c_int64 * len(ps))(*[d.numel() * d.element_size() for d, _ in ps])
        _lib.check(L.univl_copy_many(srcs, dsts, sizes, len(ps), _stream()), "copy_many")


def rows_zero(table, lst, meta):
    _lib.check(_lib.lib().univl_rows_zero(_p(table), table.shape[0], _p(lst), _p(meta), _stream()), "rows_zero")


def rows_append(ids, lst, meta, reset, ever=None):
    """ever: optional uint8 [rows] sticky "row has been written" flags (engine.FlatParams.word_ever)."""
    _lib.check(_lib.lib().univl_rows_append(_p(ids), ids.numel(), _p(lst), lst.numel(), _p(meta), int(bool(reset)), _p(ever),
                                            0 if ever is None else ever.numel(), _stream()), "rows_append")


def rows_sumsq(table, lst, meta, out):
    _lib.check(_lib.lib().univl_rows_sumsq(_p(table), table.shape[0], _p(lst), _p(meta), _p(out), _stream()), "rows_sumsq")


def embed_scatter(ids, rows, scale, dword):
    _require_gpu(ids, rows, dword)
    _lib.check(_lib.lib().univl_embed_scatter(_p(ids), _p(rows), ids.numel(), float(scale), _p(dword), _stream()), "embed_scatter")


def rows_gather_sum(rows, period, out):
    """out[s, :] += sum of rows[s::period, :] (fixed order); rows [n_rows, n] fp32, out [period, n] fp32 (a view of a table's first rows)."""
    _require_gpu(rows, out)
    _lib.check(_lib.lib().univl_rows_gather_sum(_p(rows), rows.shape[0], int(period), rows.shape[1], _p(out), _stream()), "rows_gather_sum")


def sumsq_finish(partials, seg, start, count, out):
    _lib.check(_lib.lib().univl_sumsq_finish(_p(partials), _p(seg), _p(start), _p(count), seg.numel(), _p(out), _stream()),
               "sumsq_finish")


def gather_rows(src, dst, idx, rows, row_stride_bytes, copy_bytes):
    _require_gpu(src, dst, idx)
    _lib.check(_lib.lib().univl_gather_rows(_p(src), _p(dst), _p(idx), rows, row_stride_bytes, copy_bytes, _stream()), "gather_rows")


def log_softmax_rows(x, n):
    """x: [rows, ld] fp32 (ld >= n), in place over the first n columns."""
    _require_gpu(x)
    _lib.check(_lib.lib().univl_log_softmax_rows(_p(x), x.shape[0], n, x.stride(0), _stream()), "log_softmax_rows")


def rank_counts(sim):
    """sim: [n, n] fp32 device tensor (row stride >= n).  Returns (gt, eq) int32 [n]."""
    _require_gpu(sim)
    n = sim.shape[0]
    assert sim.shape[1] == n and sim.dtype == torch.float32 and sim.stride(1) == 1
    gt = torch.empty(n, dtype=torch.int32, device=sim.device)
    eq = torch.empty(n, dtype=torch.int32, device=sim.device)
    _lib.check(_lib.lib().univl_rank_counts(_p(sim), n, sim.stride(0), _p(gt), _p(eq), _stream()), "rank_counts")
    return gt, eq


def cast_bf16(src, dst):
    _lib.check(_lib.lib().univl_cast_bf16(_p(src), _p(dst), src.numel(), _stream()), "cast_bf16")


def cast_bf16_pair(src, hi, lo):
    """hi <- bf16(src), lo <- bf16(src - hi); hi may be None (only the lo half is written)."""
    _lib.check(_lib.lib().univl_cast_bf16_pair(_p(src), _p(hi), _p(lo), src.numel(), _stream()), "cast_bf16_pair")


def cast_f32(src16, dst):
    _lib.check(_lib.lib().univl_cast_f32(_p(src16), _p(dst), src16.numel(), _stream()), "cast_f32")


def stamp(out):
    """out: one int64 / uint64 device word <- the device wall clock when this node runs (measurement, include/univl_hip.h)."""
    _lib.check(_lib.lib().univl_stamp(_p(out), _stream()), "stamp")


def bump_counter(ctr):
    _lib.check(_lib.lib().univl_bump_counter(_p(ctr), _stream()), "bump_counter")


def pair_concat_fwd(seq, vis, amask, vmask, tidx, vidx, P, W, F, out, out_mask):
    _lib.check(_lib.lib().univl_pair_concat_fwd(_p(seq), _p(vis), _p(amask), _p(vmask), _p(tidx), _p(vidx), P, W, F, _p(out),
                                                _p(out_mask), _stream()), "pair_concat_fwd")


def pair_concat_bwd(dout, tidx, vidx, P, W, F, dseq, dvis):
    _lib.check(_lib.lib().univl_pair_concat_bwd(_p(dout), _p(tidx), _p(vidx), P, W, F, _p(dseq), _p(dvis), _stream()), "pair_concat_bwd")


def postype_fwd(pos, type_emb, W, S, out):
    _lib.check(_lib.lib().univl_postype_fwd(_p(pos), _p(type_emb), W, S, _p(out), _stream()), "postype_fwd")


def postype_bwd(dtable, W, S, dpos, dtype_emb):
    _lib.check(_lib.lib().univl_postype_bwd(_p(dtable), W, S, _p(dpos), _p(dtype_emb), _stream()), "postype_bwd")


def tanh_fwd(x, y):
    _lib.check(_lib.lib().univl_tanh_fwd(_p(x), _p(y), x.numel(), _stream()), "tanh_fwd")


def tanh_bwd(dy, y, dx):
    _lib.check(_lib.lib().univl_tanh_bwd(dtype_code(dx.dtype), _p(dy), _p(y), _p(dx), y.numel(), _stream()), "tanh_bwd")


def gelu_bwd(dg, u, du):
    _lib.check(_lib.lib().univl_gelu_bwd(dtype_code(du.dtype), _p(dg), _p(u), _p(du), u.numel(), _stream()), "gelu_bwd")


def simdense_fwd(x, w, b, out):
    _lib.check(_lib.lib().univl_simdense_fwd(_p(x), _p(w), _p(b), x.shape[0], _p(out), _stream()), "simdense_fwd")


def simdense_bwd(ds, x, w, dx, dw, db):
    _lib.check(_lib.lib().univl_simdense_bwd(_p(ds), _p(x), _p(w), x.shape[0], _p(dx), _p(dw), _p(db), _stream()), "simdense_bwd")


def ce_loss(logits, labels, V, scratch2, loss, dlogits, ignore_index=-1):
    """logits: [rows, ld] fp32 view; dlogits: [rows, lddl] compute-type view."""
    _lib.check(_lib.lib().univl_ce_loss(dtype_code(dlogits.dtype), _p(logits), logits.stride(0), _p(labels), logits.shape[0], V,
                                        ignore_index, _p(scratch2), _p(loss), _p(dlogits), dlogits.stride(0), _stream()), "ce_loss")


def vocab_ce_desc(x, table, bias, labels, dlogits, V, ignore_index=-1):
    """K16 descriptor (include/univl_hip.h: UnivlVocabCE) + the buffers it owns: x [rows, K], table [V(+pad), K] in the compute type,
    dlogits [rows, lddl] in the compute type.  Returns (desc, buffers) -- keep `buffers` alive as long as the descriptor is used."""
    rows, K = x.shape
    slots = (V + 127) // 128
    dev = x.device
    b = dict(partial=torch.empty(rows, slots, 2, device=dev), label_logit=torch.zeros(rows, device=dev), lse=torch.empty(rows, device=dev),
             rowloss=torch.empty(rows, device=dev), scratch=torch.zeros(2, device=dev), loss=torch.zeros(1, device=dev),
             keep=(x, table, bias, labels, dlogits))
    d = _lib.VocabCE()
    d.dtype, d.rows, d.V, d.K = dtype_code(x.dtype), rows, V, K
    d.x, d.ldx, d.table, d.ldt = _p(x), x.stride(0), _p(table), table.stride(0)
    d.bias, d.labels, d.ignore_index, d.slots = (_p(bias) if bias is not None else None), _p(labels), ignore_index, slots
    d.partial, d.label_logit, d.lse, d.rowloss = _p(b["partial"]), _p(b["label_logit"]), _p(b["lse"]), _p(b["rowloss"])
    d.scratch2, d.loss, d.gout = _p(b["scratch"]), _p(b["loss"]), None
    d.dlogits, d.lddl = _p(dlogits), dlogits.stride(0)
    return d, b


def vocab_ce_fwd(desc):
    _lib.check(_lib.lib().univl_vocab_ce_fwd(C.byref(desc), _stream()), "vocab_ce_fwd")


def vocab_ce_bwd(desc):
    _lib.check(_lib.lib().univl_vocab_ce_bwd(C.byref(desc), _stream()), "vocab_ce_bwd")


def mfm_nce_loss(logits, vmask, labels, scratch2, loss, dlogits):
    n = logits.shape[0]
    _lib.check(_lib.lib().univl_mfm_nce_loss(_p(logits), logits.stride(0), _p(vmask), _p(labels), n, _p(scratch2), _p(loss),
                                             _p(dlogits), dlogits.stride(0), _stream()), "mfm_nce_loss")


def colsum(x, out):
    """out[c] += sum_r x[r, c]; x: [rows, ld] view in the compute type."""
    _lib.check(_lib.lib().univl_colsum(dtype_code(x.dtype), _p(x), x.stride(0), x.shape[0], x.shape[1], _p(out), _stream()), "colsum")


def scale_ct(x, s):
    _lib.check(_lib.lib().univl_scale_ct_by_device_scalar(dtype_code(x.dtype), _p(x), x.numel(), _p(s), _stream()), "scale_ct")
