"""ctypes binding of libunivl_hip.so (include/univl_hip.h).  The library is REQUIRED: there is no CPU or
PyTorch fallback anywhere in univl_amd -- if the shared object is missing or a call fails, a RuntimeError is
raised (the reference's convention is exception-based too, e.g. modules/module_bert.py:152-155)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UNIVL_LIB") or os.path.join(_HERE, "lib", "libunivl_hip.so")   # UNIVL_LIB: A/B builds

DT_F32, DT_BF16 = 0, 1
GEMM_ACCUM, GEMM_GELU_FWD, GEMM_GELU_BWD, GEMM_DBIAS_ATOMIC, GEMM_NT_OUT, GEMM_AUX_F32 = 1, 2, 4, 16, 32, 512
GEMM_GROUP_MAX = 4
EUNSUPPORTED = -3          # csrc/common.h: the entry point does not carry this case (callers fall back)

vp, i32, i64, f32, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64


class Gemm(C.Structure):
    _fields_ = [("dtype", i32), ("trans_a", i32), ("trans_b", i32), ("M", i32), ("N", i32), ("K", i32),
                ("A", vp), ("lda", i64), ("B", vp), ("ldb", i64), ("C32", vp), ("C16", vp), ("ldc", i64),
                ("bias", vp), ("R", vp), ("ldr", i64), ("aux", vp), ("ldaux", i64), ("dbias", vp),
                ("alpha", f32), ("flags", i32), ("ksplit", i32), ("tile", i32), ("sumsq", vp), ("sumsq_rows", i32), ("sumsq_stride", i32),
                ("stages", i32), ("waves", i32), ("A_lo", vp), ("B_lo", vp), ("C16_lo", vp)]


class LayerNorm(C.Structure):
    _fields_ = [("dtype", i32), ("rows", i32), ("N", i32), ("x_f64", i32), ("x", vp), ("residual", vp),
                ("pos", vp), ("pos_period", i32), ("gamma", vp), ("beta", vp), ("eps", f32), ("y", vp),
                ("stats", vp), ("out32", vp), ("out16", vp), ("p_pre", f32), ("p_post", f32), ("seed", u64),
                ("off_pre", u64), ("off_post", u64), ("seed_dev", vp), ("dout", vp), ("dx32", vp), ("dxd32", vp), ("dxd16", vp),
                ("dgamma", vp), ("dbeta", vp), ("dbias", vp), ("dpos", vp), ("out16_lo", vp)]


class Attention(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("H", i32), ("Sq", i32), ("Sk", i32), ("q", vp), ("ldq", i64),
                ("k", vp), ("ldk", i64), ("v", vp), ("ldv", i64), ("key_mask", vp), ("causal", i32),
                ("out", vp), ("ldo", i64), ("lse", vp), ("p_drop", f32), ("seed", u64), ("offset", u64), ("seed_dev", vp),
                ("dout", vp), ("lddo", i64), ("dq", vp), ("lddq", i64), ("dk", vp), ("lddk", i64),
                ("dv", vp), ("lddv", i64), ("bsk", i64), ("bsv", i64), ("out_lo", vp)]


class EmbedText(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("S", i32), ("N", i32), ("ids", vp), ("type_ids", vp), ("word", vp),
                ("pos", vp), ("type", vp), ("gamma", vp), ("beta", vp), ("eps", f32), ("y", vp), ("stats", vp),
                ("out32", vp), ("out16", vp), ("p_post", f32), ("seed", u64), ("off_post", u64), ("seed_dev", vp), ("dout", vp),
                ("dword", vp), ("dpos", vp), ("dtype_emb", vp), ("dgamma", vp), ("dbeta", vp), ("drows", vp), ("out16_lo", vp)]


class Pool(C.Structure):
    _fields_ = [("B", i32), ("S", i32), ("N", i32), ("x", vp), ("ldx_row", i64), ("mask", vp),
                ("skip_first", i32), ("normalize", i32), ("mean", vp), ("out", vp), ("dout", vp), ("dx", vp), ("accumulate", i32),
                ("dsim", vp), ("ldsim", i64), ("other", vp), ("n_other", i32), ("transpose", i32), ("gscale", vp)]


class Seg(C.Structure):
    _fields_ = [("offset", i64), ("numel", i64), ("lr", f32), ("weight_decay", f32), ("max_grad_norm", f32),
                ("active", i32)]


class Adam(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("p16", vp), ("segs", vp), ("nseg", i32),
                ("chunk_seg", vp), ("chunk_off", vp), ("chunk_len", vp), ("nchunk", i32), ("sumsq", vp),
                ("coef", vp), ("step", vp), ("b1", f32), ("b2", f32), ("eps", f32), ("warmup", f32),
                ("t_total", i32), ("seg_scalars", vp), ("schedule", i32), ("row_flags", vp), ("flag_seg", i32), ("row_len", i32), ("p16_lo", vp)]


class VocabCE(C.Structure):
    _fields_ = [("dtype", i32), ("rows", i32), ("V", i32), ("K", i32), ("x", vp), ("ldx", i64), ("table", vp), ("ldt", i64), ("bias", vp),
                ("labels", vp), ("ignore_index", i32), ("slots", i32), ("partial", vp), ("label_logit", vp), ("lse", vp), ("rowloss", vp),
                ("scratch2", vp), ("loss", vp), ("gout", vp), ("dlogits", vp), ("lddl", i64)]


_STRUCTS = [Gemm, LayerNorm, Attention, EmbedText, Pool, Seg, Adam, VocabCE]
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    """Load the shared object (once).  Raises RuntimeError when it is missing or its ABI does not match."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libunivl_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` (hipcc --offload-arch=gfx950).  univl_amd has no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.univl_last_error.restype = C.c_char_p
    L.univl_struct_size.argtypes = [i32]
    for k, st in enumerate(_STRUCTS):
        n = L.univl_struct_size(k)
        if n != C.sizeof(st):
            raise RuntimeError("ABI mismatch for %s: library %d bytes, ctypes %d" % (st.__name__, n, C.sizeof(st)))
    for name in ("univl_vocab_ce_fwd", "univl_vocab_ce_bwd", "univl_gemm", "univl_layernorm_fwd", "univl_layernorm_bwd", "univl_attention_fwd",
                 "univl_attention_bwd", "univl_embed_text_fwd", "univl_embed_text_bwd", "univl_pool_fwd",
                 "univl_pool_bwd", "univl_bert_adam"):
        getattr(L, name).argtypes = [vp, vp]
        getattr(L, name).restype = i32
    L.univl_pool_pair_fwd.argtypes = [vp, vp, vp]
    L.univl_pool_pair_bwd.argtypes = [vp, vp, vp]
    L.univl_gemm_group.argtypes = [vp, i32, vp]
    L.univl_gemm_group.restype = i32
    L.univl_gemm_group_limited.argtypes = [vp, i32, i32, vp]
    L.univl_gemm_pair.argtypes = [vp, vp, i32, vp]
    L.univl_gemm_rider.argtypes = [vp, vp, i32, i32, i32, vp]
    L.univl_gemm_rider_prime.argtypes = [vp]
    L.univl_gemm_rider_fits.argtypes = [vp]
    L.univl_gemm_ln.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.univl_gemm_ln.restype = i32
    L.univl_gemm_pair_ln.argtypes = [vp, vp, vp, vp, i32, vp]
    L.univl_gemm_pair_ln.restype = i32
    L.univl_gemm_tile_map.argtypes = [i32, i32, i32, i32, i32, C.POINTER(i32)]
    L.univl_embed_scatter.argtypes = [vp, vp, i64, f32, vp, vp]
    L.univl_rows_gather_sum.argtypes = [vp, i32, i32, i32, vp, vp]
    L.univl_rows_zero.argtypes = [vp, i64, vp, vp, vp]
    L.univl_rows_append.argtypes = [vp, i32, vp, i32, vp, i32, vp, i64, vp]
    L.univl_rows_sumsq.argtypes = [vp, i64, vp, vp, vp, vp]
    L.univl_zero_many.argtypes = [vp, vp, i32, vp]
    L.univl_copy_many.argtypes = [vp, vp, vp, i32, vp]
    L.univl_maxmargin_loss.argtypes = [vp, i32, i32, f32, vp, vp, vp, vp]
    L.univl_crossen_loss.argtypes = [vp, i32, i32, vp, vp, vp]
    L.univl_milnce_loss.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    L.univl_gather_rows.argtypes = [vp, vp, vp, i32, i64, i64, vp]
    L.univl_log_softmax_rows.argtypes = [vp, i32, i32, i64, vp]
    L.univl_rank_counts.argtypes = [vp, i32, i64, vp, vp, vp]
    L.univl_scale_by_device_scalar.argtypes = [vp, i64, vp, vp]
    L.univl_pair_concat_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]
    L.univl_pair_concat_bwd.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp]
    L.univl_postype_fwd.argtypes = [vp, vp, i32, i32, vp, vp]
    L.univl_postype_bwd.argtypes = [vp, i32, i32, vp, vp, vp]
    L.univl_tanh_fwd.argtypes = [vp, vp, i64, vp]
    L.univl_tanh_bwd.argtypes = [i32, vp, vp, vp, i64, vp]
    L.univl_gelu_bwd.argtypes = [i32, vp, vp, vp, i64, vp]
    L.univl_colsum.argtypes = [i32, vp, i64, i32, i32, vp, vp]
    L.univl_scale_ct_by_device_scalar.argtypes = [i32, vp, i64, vp, vp]
    L.univl_simdense_fwd.argtypes = [vp, vp, vp, i32, vp, vp]
    L.univl_simdense_bwd.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp]
    L.univl_ce_loss.argtypes = [i32, vp, i64, vp, i32, i32, i32, vp, vp, vp, i64, vp]
    L.univl_mfm_nce_loss.argtypes = [vp, i64, vp, vp, i32, vp, vp, vp, i64, vp]
    L.univl_grad_sumsq.argtypes = [vp, vp, i32, vp, vp, vp, i32, vp, vp]
    L.univl_sumsq_finish.argtypes = [vp, vp, vp, vp, i32, vp, vp]
    L.univl_clip_coef.argtypes = [vp, vp, i32, f32, vp, vp]
    L.univl_scale_grads.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp]
    L.univl_cast_bf16.argtypes = [vp, vp, i64, vp]
    L.univl_cast_bf16_pair.argtypes = [vp, vp, vp, i64, vp]
    L.univl_cast_f32.argtypes = [vp, vp, i64, vp]
    L.univl_bert_adam_range.argtypes = [vp, i32, i32, i32, i32, vp]
    L.univl_bump_counter.argtypes = [vp, vp]
    L.univl_probe_layouts.argtypes = [vp, i32, vp]
    L.univl_stamp.argtypes = [vp, vp]
    L.univl_device_info.argtypes = [C.POINTER(i32), C.c_char_p, i32]
    L.univl_init.argtypes = [i32]
    L.univl_allreduce_bucket.argtypes = [vp, C.c_size_t, i32, i32, vp, vp]
    L.univl_set_deterministic.argtypes = [i32]
    _lib = L
    return L


def deterministic():
    """UNIVL_DETERMINISTIC=1 (read when a model is flattened onto its device) or set_deterministic(True)."""
    return bool(lib().univl_get_deterministic())


def set_deterministic(on=True):
    """Fixed-order reductions in every kernel launched from now on (include/univl_hip.h: univl_set_deterministic) -- two runs on the
    same inputs give bit-identical results.  Allocates the library's scratch ring on the CURRENT device: call it outside a stream
    capture, after torch.cuda.set_device.  Plans / hipGraphs built earlier keep the mode they were built in (a model rebuilds its
    plans when its _steps are cleared, e.g. by model.to(device))."""
    check(lib().univl_set_deterministic(1 if on else 0), "set_deterministic")


EXPORTED = ["univl_last_error", "univl_version", "univl_struct_size", "univl_device_info", "univl_init", "univl_destroy",
            "univl_allreduce_bucket", "univl_set_deterministic", "univl_get_deterministic", "univl_gemm", "univl_gemm_group_limited", "univl_gemm_group", "univl_gemm_tile_map", "univl_gemm256_layout", "univl_gemm_pair", "univl_gemm_rider", "univl_gemm_rider_fits", "univl_gemm_rider_prime", "univl_gemm_ln", "univl_gemm_pair_ln",
            "univl_layernorm_fwd", "univl_layernorm_bwd", "univl_attention_fwd", "univl_attention_bwd", "univl_attention_bwd_fused", "univl_attention_fwd_fused",
            "univl_embed_text_fwd", "univl_embed_text_bwd", "univl_embed_scatter", "univl_rows_gather_sum", "univl_rows_zero", "univl_rows_append",
            "univl_rows_sumsq", "univl_zero_many", "univl_copy_many", "univl_pool_fwd", "univl_pool_bwd", "univl_pool_pair_fwd", "univl_pool_pair_bwd",
            "univl_maxmargin_loss", "univl_crossen_loss", "univl_milnce_loss", "univl_rank_counts", "univl_gather_rows", "univl_log_softmax_rows", "univl_scale_by_device_scalar", "univl_pair_concat_fwd", "univl_pair_concat_bwd", "univl_postype_fwd", "univl_postype_bwd", "univl_tanh_fwd",
            "univl_tanh_bwd", "univl_gelu_bwd", "univl_colsum", "univl_scale_ct_by_device_scalar", "univl_simdense_fwd", "univl_simdense_bwd", "univl_ce_loss", "univl_vocab_ce_fwd", "univl_vocab_ce_bwd", "univl_mfm_nce_loss", "univl_grad_sumsq", "univl_sumsq_finish",
            "univl_clip_coef", "univl_scale_grads", "univl_bert_adam", "univl_bert_adam_range", "univl_cast_bf16", "univl_cast_bf16_pair", "univl_cast_f32", "univl_bump_counter", "univl_probe_layouts", "univl_stamp"]


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("libunivl_hip %s failed (%d): %s" % (what, rc, lib().univl_last_error().decode()))
