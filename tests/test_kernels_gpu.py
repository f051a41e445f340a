"""Per-kernel parity: every C-ABI entry point against the oracle's restatement of the reference op sequence it
replaces (oracle/univl_oracle.py), on seeded inputs.  Needs a real MI355X (`-m gpu`).

Tolerances: fp32 mode 1e-3 absolute on O(1) quantities as BASELINE.json's north_star states (observed ~1e-5);
bf16 mode 1e-2 on quantities of O(1) after normalising by the tensor's scale (bf16 operands, fp32 accumulate)."""
import math
import os

import numpy as np
import pytest
import torch

import univl_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from univl_amd import ops, _lib

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def tol(dtype):
    return 1e-3 if dtype == torch.float32 else 1e-2


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------------------------------------ probes
def _tv(a, b, salt):
    return float(((a * 7 + b * 13 + salt * 5 + a * b) % 5) - 2)


def test_probe_layouts():
    """MFMA operand / accumulator lane maps and the transpose read, as assumed by csrc/common.h."""
    out = ops.probe_layouts().cpu().numpy()
    for (CH, base_k, base_t, base_c) in ((32, 0, 256, 1024), (16, 512, 768, 1280)):
        X = np.array([[_tv(r, k, 1) for k in range(CH)] for r in range(32)])
        Y = np.array([[_tv(r, k, 2) for k in range(CH)] for r in range(16)])
        Z = np.array([[_tv(r, k, 3) for k in range(CH)] for r in range(16)])
        Cref = X[:16] @ Y.T
        np.testing.assert_array_equal(out[base_k:base_k + 256].reshape(16, 16), Cref, err_msg=f"K-major CH={CH}")
        np.testing.assert_array_equal(out[base_t:base_t + 256].reshape(16, 16), Cref, err_msg=f"T-major CH={CH}")
        S = X[:CH] @ Y.T               # [CH,16]: rows are the contraction index of the chained product
        np.testing.assert_array_equal(out[base_c:base_c + 256].reshape(16, 16), Z @ S, err_msg=f"chain CH={CH}")
    # raw transpose-read dump: lane l (group g = l>>4, i = l&15) must receive img[4g + e][i], e = 0..3
    raw = out[1536:1792].reshape(64, 4)
    exp = np.array([[64 * (4 * (l >> 4) + e) + (l & 15) for e in range(4)] for l in range(64)], dtype=np.float32)
    np.testing.assert_array_equal(raw, exp)


# -------------------------------------------------------------------------------------------------- GEMM
GEMM_SHAPES = [(192, 768, 768), (60, 2304, 768), (192, 1000, 768), (200, 768, 3072), (4, 4, 768), (36, 1, 768),
               (300, 264, 100)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("tile", [64, 128])
def test_gemm_forward_bias(dtype, M, N, K, tile):
    Kp = (K + 7) // 8 * 8
    A = torch.zeros(M, Kp); A[:, :K] = gen(M, K, seed=1)
    B = torch.zeros(N, Kp); B[:, :K] = gen(N, K, seed=2, scale=0.05)
    bias = gen(N, seed=3)
    Ad, Bd = A.to(DEV, dtype), B.to(DEV, dtype)
    ldc = (N + 7) // 8 * 8
    out32 = torch.zeros(M, ldc, device=DEV)
    out16 = torch.zeros(M, ldc, device=DEV, dtype=dtype)
    ops.gemm(Ad, Bd, M, N, K, out32=out32, out16=out16, bias=bias.to(DEV), tile=tile)
    ref = Ad.double().cpu()[:, :K] @ Bd.double().cpu()[:, :K].T + bias.double()
    assert rel_err(out32[:, :N], ref) < (1e-5 if dtype == torch.float32 else 2e-3)
    assert rel_err(out16[:, :N].float(), ref) < tol(dtype)
    assert float(out32[:, N:].abs().max() if ldc > N else 0.0) == 0.0      # padding untouched


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [64, 128])
def test_gemm_dgrad_wgrad(dtype, tile):
    T, N, K = 100, 768, 3072                   # Y[T,N] = X[T,K] W[N,K]^T
    X = gen(T, K, seed=4).to(DEV, dtype)
    W = gen(N, K, seed=5, scale=0.05).to(DEV, dtype)
    dY = gen(T, N, seed=6).to(DEV, dtype)
    res = gen(T, K, seed=7).to(DEV)
    # dgrad dX[T,K] = dY[T,N] . W[N,K]  (B T-major) + residual
    dX = torch.zeros(T, K, device=DEV)
    ops.gemm(dY, W, T, K, N, trans_b=True, out32=dX, residual=res, tile=tile)
    ref = dY.double().cpu() @ W.double().cpu() + res.double().cpu()
    assert rel_err(dX, ref) < (1e-5 if dtype == torch.float32 else 2e-3)
    # wgrad dW[N,K] = dY^T[N,T] . X[T,K]  (both T-major), with bias grad and accumulate
    dW = torch.ones(N, K, device=DEV)
    db = torch.ones(N, device=DEV)
    ops.gemm(dY, X, N, K, T, trans_a=True, trans_b=True, out32=dW, dbias=db, accumulate=True, tile=tile)
    ref = dY.double().cpu().T @ X.double().cpu() + 1.0
    assert rel_err(dW, ref) < (1e-5 if dtype == torch.float32 else 2e-3)
    assert rel_err(db, dY.double().cpu().sum(0) + 1.0) < (1e-5 if dtype == torch.float32 else 2e-3)
    # T-major A with K-major B (not used by the model but part of the ABI)
    C = torch.zeros(N, T, device=DEV)
    ops.gemm(dY, X, N, T, 64, trans_a=True, trans_b=False, out32=C, tile=tile)
    ref = dY.double().cpu()[:64].T @ X.double().cpu()[:, :64].T
    assert rel_err(C, ref) < (1e-5 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_gelu_epilogues_and_splitk(dtype):
    T, N, K = 192, 3072, 768
    X = gen(T, K, seed=8).to(DEV, dtype)
    W = gen(N, K, seed=9, scale=0.05).to(DEV, dtype)
    b = gen(N, seed=10).to(DEV)
    u = torch.zeros(T, N, device=DEV, dtype=dtype)
    f = torch.zeros(T, N, device=DEV, dtype=dtype)
    ops.gemm(X, W, T, N, K, out16=f, bias=b, aux=u, gelu="fwd")
    uref = X.double().cpu() @ W.double().cpu().T + b.double().cpu()
    assert rel_err(u.float(), uref) < tol(dtype)
    assert rel_err(f.float(), O.gelu(uref)) < tol(dtype)
    # gelu backward epilogue: dU = (dF . W2) * gelu'(u)
    dZ = gen(T, 768, seed=11).to(DEV, dtype)
    W2 = gen(768, N, seed=12, scale=0.05).to(DEV, dtype)
    dU = torch.zeros(T, N, device=DEV, dtype=dtype)
    ops.gemm(dZ, W2, T, N, 768, trans_b=True, out16=dU, aux=u, gelu="bwd")
    uu = u.double().cpu().requires_grad_(True)
    O.gelu(uu).backward(dZ.double().cpu() @ W2.double().cpu())
    assert rel_err(dU.float(), uu.grad) < tol(dtype)
    # split-K (fp32 atomics into a zeroed buffer), bias + residual applied exactly once
    Y = torch.zeros(T, 768, device=DEV)
    res = gen(T, 768, seed=13).to(DEV)
    b2 = gen(768, seed=14).to(DEV)
    ops.gemm(f, W2, T, 768, N, out32=Y, bias=b2, residual=res, ksplit=4)
    ref = f.double().cpu() @ W2.double().cpu().T + b2.double().cpu() + res.double().cpu()
    assert rel_err(Y, ref) < (1e-5 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T", [192, 100, 700])
def test_gemm_group_wgrads(dtype, T):
    """One launch for the four weight-gradient GEMMs of a layer == four separate launches (bit-identical) == reference;
    mixed shapes, ragged contraction length, per-member accumulate / bias-gradient flags, a split-K member."""
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]       # (rows of dY^T, cols of X)
    descs, outs, refs, dbs, singles, keep = [], [], [], [], [], []
    for k, (N, K) in enumerate(shapes):
        dY = gen(T, N, seed=20 + k).to(DEV, dtype)
        X = gen(T, K, seed=30 + k).to(DEV, dtype)
        keep += [dY, X]                                    # descriptors hold raw pointers
        acc = k == 1
        dW = torch.full((N, K), 1.0 if acc else 0.0, device=DEV)
        db = torch.zeros(N, device=DEV) if k in (1, 3) else None
        ks = 2 if (k == 2 and T > 256) else 1
        descs.append(ops.gemm_desc(dY, X, N, K, T, trans_a=True, trans_b=True, out32=dW, dbias=db, dbias_atomic=db is not None,
                                   accumulate=acc, ksplit=ks))
        outs.append(dW)
        dbs.append((db, dY))
        refs.append(dY.double().cpu().T @ X.double().cpu() + (1.0 if acc else 0.0))
        dW1 = torch.full((N, K), 1.0 if acc else 0.0, device=DEV)
        ops.gemm(dY, X, N, K, T, trans_a=True, trans_b=True, out32=dW1, accumulate=acc, ksplit=ks)
        singles.append(dW1)
    ops.gemm_group(descs)
    for k, (dW, ref) in enumerate(zip(outs, refs)):
        assert rel_err(dW, ref) < (1e-5 if dtype == torch.float32 else 2e-3), k
        if not (k == 2 and T > 256):                       # atomics of the split-K member may reorder
            assert torch.equal(dW, singles[k]), k
    for db, dY in dbs:
        if db is not None:
            assert rel_err(db, dY.double().cpu().sum(0)) < (1e-5 if dtype == torch.float32 else 2e-3)
    # the same launch on a capped grid (univl_gemm_group_limited: the workgroups walk the tiles): identical results
    for k, dW in enumerate(outs):
        dW.fill_(1.0 if k == 1 else 0.0)
    for db, _ in dbs:
        if db is not None:
            db.zero_()
    ops.gemm_group(descs, max_blocks=96)
    for k, dW in enumerate(outs):
        if not (k == 2 and T > 256):
            assert torch.equal(dW, singles[k]), ("capped grid", k)
    with pytest.raises(RuntimeError):
        ops.gemm_group(descs + descs[:1])                  # more than GEMM_GROUP_MAX members
    mixed = [descs[0], ops.gemm_desc(dbs[0][1], dbs[0][1], 4, 4, 64, out32=torch.zeros(4, 4, device=DEV))]
    with pytest.raises(RuntimeError):
        ops.gemm_group(mixed)                              # different operand layouts


GEMM_VARIANTS = [(64, 2, 4), (64, 2, 8), (128, 2, 8)]   # UnivlGemm.tile, .stages, .waves (reference: 128 / 2 / 4)


@pytest.mark.parametrize("M,N,K", [(700, 1000, 768), (6144, 768, 3072), (300, 264, 100), (256, 128, 64), (260, 130, 200),
                                    (1024, 2304, 1096)])
def test_gemm_tile_and_wave_variants_forward_dgrad(M, N, K):
    """The 64 tile and the 8-wave workgroups against the 128-tile / 4-wave kernel: BIT-identical (every output element contracts the
    same 32-deep chunks in the same order in every variant), and that kernel against fp64.  Shapes: ragged rows / columns, short
    contractions (K = 64, 100), a partial last K tile (K = 200, 1096), the MFMA-bound shape of bs 128 (6144 x 768 x 3072)."""
    dtype = torch.bfloat16
    Kp = (K + 7) // 8 * 8
    A = torch.zeros(M, Kp); A[:, :K] = gen(M, K, seed=1)
    B = torch.zeros(N, Kp); B[:, :K] = gen(N, K, seed=2, scale=0.05)
    bias = gen(N, seed=3).to(DEV)
    Ad, Bd = A.to(DEV, dtype), B.to(DEV, dtype)
    ldc = (N + 7) // 8 * 8
    base = torch.zeros(M, ldc, device=DEV)
    base16 = torch.zeros(M, ldc, device=DEV, dtype=dtype)
    ops.gemm(Ad, Bd, M, N, K, out32=base, out16=base16, bias=bias, tile=128, stages=2, waves=4)
    ref = Ad.double().cpu()[:, :K] @ Bd.double().cpu()[:, :K].T + bias.double().cpu()
    assert rel_err(base[:, :N], ref) < 2e-3
    # dgrad layout: dX[M, Kp] = dY[M, N] . W[N, Kp]   (B T-major) + fp32 residual
    Np = (N + 7) // 8 * 8
    dY = torch.zeros(M, Np); dY[:, :N] = gen(M, N, seed=4)
    dYd = dY.to(DEV, dtype)
    Wd = torch.zeros(Np, Kp, device=DEV, dtype=dtype); Wd[:N] = Bd
    res = gen(M, Kp, seed=5).to(DEV)
    dbase = torch.zeros(M, Kp, device=DEV)
    ops.gemm(dYd, Wd, M, Kp, N, trans_b=True, out32=dbase, residual=res, tile=128, stages=2, waves=4)
    ref = dYd.double().cpu()[:, :N] @ Wd.double().cpu()[:N] + res.double().cpu()
    assert rel_err(dbase, ref) < 2e-3
    for tile, stages, waves in GEMM_VARIANTS:
        for rep in range(2):                               # a race between the stages would come and go
            out = torch.zeros(M, ldc, device=DEV)
            o16 = torch.zeros(M, ldc, device=DEV, dtype=dtype)
            ops.gemm(Ad, Bd, M, N, K, out32=out, out16=o16, bias=bias, tile=tile, stages=stages, waves=waves)
            assert torch.equal(out, base), ("forward", tile, stages, waves, rep, float((out - base).abs().max()))
            assert torch.equal(o16, base16), ("forward bf16", tile, stages, waves, rep)
            out = torch.zeros(M, Kp, device=DEV)
            ops.gemm(dYd, Wd, M, Kp, N, trans_b=True, out32=out, residual=res, tile=tile, stages=stages, waves=waves)
            assert torch.equal(out, dbase), ("dgrad", tile, stages, waves, rep, float((out - dbase).abs().max()))


@pytest.mark.parametrize("M,N,K", [(6144, 768, 3072), (700, 1000, 768), (300, 264, 100), (130, 70, 64)])
def test_gemm_half_width_tiles_forward_dgrad(M, N, K):
    """UnivlGemm.tile = 12864 / 64128 (128 x 64 and 64 x 128 tiles on 4 or 8 waves; chosen by slot fill from ~2000 rows on): BIT-identical to the
    128 tile (same 64-deep K steps, same chunk order per output element) for the two layouts they serve -- forward
    (K-major x K-major, bias, fp32 + bf16 outputs) and dgrad (K-major x T-major, fp32 residual); a weight-gradient
    descriptor (T-major A) falls back to the 128 tile."""
    dtype = torch.bfloat16
    Kp, Np, ldc = (K + 7) // 8 * 8, (N + 7) // 8 * 8, (N + 7) // 8 * 8
    A = torch.zeros(M, Kp); A[:, :K] = gen(M, K, seed=1)
    B = torch.zeros(N, Kp); B[:, :K] = gen(N, K, seed=2, scale=0.05)
    bias = gen(N, seed=3).to(DEV)
    Ad, Bd = A.to(DEV, dtype), B.to(DEV, dtype)
    base = torch.zeros(M, ldc, device=DEV)
    base16 = torch.zeros(M, ldc, device=DEV, dtype=dtype)
    ops.gemm(Ad, Bd, M, N, K, out32=base, out16=base16, bias=bias, tile=128, stages=2, waves=4)
    assert rel_err(base[:, :N], Ad.double().cpu()[:, :K] @ Bd.double().cpu()[:, :K].T + bias.double().cpu()) < 2e-3
    dY = torch.zeros(M, Np); dY[:, :N] = gen(M, N, seed=4)
    dYd = dY.to(DEV, dtype)
    Wd = torch.zeros(Np, Kp, device=DEV, dtype=dtype); Wd[:N] = Bd
    res = gen(M, Kp, seed=5).to(DEV)
    dbase = torch.zeros(M, Kp, device=DEV)
    ops.gemm(dYd, Wd, M, Kp, N, trans_b=True, out32=dbase, residual=res, tile=128, stages=2, waves=4)
    for tile in (12864, 64128):
        for waves in (8, 4):
            for rep in range(2):
                out = torch.zeros(M, ldc, device=DEV)
                o16 = torch.zeros(M, ldc, device=DEV, dtype=dtype)
                ops.gemm(Ad, Bd, M, N, K, out32=out, out16=o16, bias=bias, tile=tile, waves=waves)
                assert torch.equal(out, base), ("forward", tile, waves, rep, float((out - base).abs().max()))
                assert torch.equal(o16, base16), ("forward bf16", tile, waves, rep)
                out = torch.zeros(M, Kp, device=DEV)
                ops.gemm(dYd, Wd, M, Kp, N, trans_b=True, out32=out, residual=res, tile=tile, waves=waves)
                assert torch.equal(out, dbase), ("dgrad", tile, waves, rep, float((out - dbase).abs().max()))
        # T-major A: served by the 128 tile
        wg = torch.zeros(N, Kp, device=DEV)
        wb = torch.zeros(N, Kp, device=DEV)
        ops.gemm(dYd, Ad, N, Kp, M, trans_a=True, trans_b=True, out32=wb, tile=128, stages=2, waves=4)
        ops.gemm(dYd, Ad, N, Kp, M, trans_a=True, trans_b=True, out32=wg, tile=tile)
        assert torch.equal(wg, wb), ("wgrad fallback", tile)


@pytest.mark.parametrize("T", [192, 1000, 6144])
def test_gemm_tile_and_wave_variants_wgrad(T):
    """Weight-gradient layout (both operands T-major) with the fused bias gradient, accumulate, the per-tensor sum of squares
    (three tensors of 768 rows in one launch) and the grouped launch: every variant against the double-buffered 128 tile."""
    dtype = torch.bfloat16
    dY = gen(T, 2304, seed=1).to(DEV, dtype)
    X = gen(T, 768, seed=2).to(DEV, dtype)
    X2 = gen(T, 3072, seed=3).to(DEV, dtype)
    dY768 = dY[:, :768].contiguous()
    stride = (768 // 64) * (768 // 64) * 4

    def run(tile, stages, waves, group):
        out = torch.full((2304, 768), 0.5, device=DEV)
        db = torch.full((2304,), 0.25, device=DEV)
        part = torch.zeros(3 * stride, device=DEV)
        kw = dict(trans_a=True, trans_b=True, out32=out, dbias=db, accumulate=True, sumsq=part, sumsq_rows=768,
                  sumsq_stride=stride, tile=tile, stages=stages, waves=waves)
        if group:
            o2 = torch.zeros(768, 3072, device=DEV)
            d2 = ops.gemm_desc(dY768, X2, 768, 3072, T, trans_a=True, trans_b=True, out32=o2, tile=tile, stages=stages, waves=waves)
            ops.gemm_group([ops.gemm_desc(dY, X, 2304, 768, T, **kw), d2])
            return out, db, part.view(3, stride).sum(1), o2
        ops.gemm(dY, X, 2304, 768, T, **kw)
        return out, db, part.view(3, stride).sum(1), None

    b_out, b_db, b_ss, _ = run(128, 2, 4, False)
    assert rel_err(b_out, dY.double().cpu().T @ X.double().cpu() + 0.5) < 2e-3
    assert rel_err(b_ss, (b_out.double() ** 2).view(3, -1).sum(1).cpu()) < 1e-5
    b_o2 = torch.zeros(768, 3072, device=DEV)
    ops.gemm(dY768, X2, 768, 3072, T, trans_a=True, trans_b=True, out32=b_o2, tile=128, stages=2, waves=4)
    assert rel_err(b_o2, dY768.double().cpu().T @ X2.double().cpu()) < 2e-3
    for tile, stages, waves in GEMM_VARIANTS:
        plain = torch.zeros(2304, 768, device=DEV)         # no sum of squares: the 64 tile keeps its 8 waves
        ops.gemm(dY, X, 2304, 768, T, trans_a=True, trans_b=True, out32=plain, tile=tile, stages=stages, waves=waves)
        assert rel_err(plain + 0.5, b_out) < 1e-6, ("plain", tile, stages, waves)
        for group in (False, True):
            out, db, ss, o2 = run(tile, stages, waves, group)
            assert torch.equal(out, b_out), (tile, stages, waves, group)
            assert rel_err(db, b_db) < 1e-5                # 64- / 128- / 256-row tiles sum the bias gradient in different orders
            assert rel_err(ss, b_ss) < 1e-5
            if o2 is not None:
                assert torch.equal(o2, b_o2), ("group member 2", tile, stages, waves)


# ---- the 256 x 256 "8-phase" body (csrc/gemm256.h): every layout, epilogue and launch form against fp64 on the host
G256_SHAPES = [(256, 256, 128), (512, 768, 256), (768, 512, 640), (1536, 2304, 768), (3072, 768, 3072)]


def _ref64(A, B, ta, tb):
    a, b = A.double().cpu(), B.double().cpu()
    return (a.T if ta else a) @ (b if tb else b.T)


@pytest.mark.parametrize("M,N,K", G256_SHAPES)
@pytest.mark.parametrize("layout", ["fwd", "dgrad", "wgrad"])
def test_gemm256_layouts_against_fp64(M, N, K, layout):
    """C = A_op . B_op^T on the 256 tile: forward (K-major, K-major), dgrad (K-major, T-major), wgrad (T-major, T-major); K = 128 runs
    only the two tail tiles of the loop, 256 one steady trip + tail, 640 / 768 / 3072 several.  Operands are random and asymmetric (a
    transposed operand or output cannot pass).  fp32 output: bf16 products summed in fp32, so the error against fp64 is round-off
    class; the same launch repeated must be bit-identical (a race between the DMA stream and the fragment reads shows up as run-to-run
    differences long before it shows up as a large error)."""
    dtype = torch.bfloat16
    ta, tb = layout == "wgrad", layout != "fwd"
    A = gen(*((K, M) if ta else (M, K)), seed=11).to(DEV, dtype)
    B = gen(*((K, N) if tb else (N, K)), seed=12).to(DEV, dtype)
    ref = _ref64(A, B, ta, tb)
    out = torch.empty(M, N, device=DEV)
    out16 = torch.empty(M, N, device=DEV, dtype=dtype)
    ops.gemm(A, B, M, N, K, trans_a=ta, trans_b=tb, out32=out, out16=out16, tile=256)
    assert rel_err(out, ref) < 2e-5, layout
    assert rel_err(out16, ref) < 6e-3, layout
    # the same product on the older tiles agrees to fp32 round-off (different summation order: 32x32x16 against 16x16x32 chunks)
    old = torch.empty(M, N, device=DEV)
    ops.gemm(A, B, M, N, K, trans_a=ta, trans_b=tb, out32=old, tile=128)
    assert rel_err(out, old) < 2e-5
    first = out.clone()
    for _ in range(10):
        out.fill_(-1.0)
        ops.gemm(A, B, M, N, K, trans_a=ta, trans_b=tb, out32=out, tile=256)
        assert torch.equal(out, first), layout


def test_gemm256_epilogues_split_k_and_fallbacks():
    dtype = torch.bfloat16
    M, N, K = 1536, 3072, 768
    X = gen(M, K, seed=1).to(DEV, dtype)
    W = gen(N, K, seed=2, scale=0.05).to(DEV, dtype)
    bias = gen(N, seed=3).to(DEV)
    pre = X.double().cpu() @ W.double().cpu().T + bias.double().cpu()
    # FFN1 forward: bias + erf-GELU, saving the pre-activation (module_bert.py:226-235, until_module.py:28-33)
    f = torch.empty(M, N, device=DEV, dtype=dtype)
    u = torch.empty(M, N, device=DEV, dtype=dtype)
    ops.gemm(X, W, M, N, K, out16=f, bias=bias, aux=u, gelu="fwd", tile=256)
    gel = pre * 0.5 * (1.0 + torch.erf(pre / math.sqrt(2.0)))
    assert rel_err(u, pre) < 6e-3
    assert rel_err(f, gel) < 6e-3
    # the branch-free erf of this body against libm's on the older tile: same bf16 results up to one rounding of a few elements
    f_old = torch.empty_like(f)
    u_old = torch.empty_like(u)
    ops.gemm(X, W, M, N, K, out16=f_old, bias=bias, aux=u_old, gelu="fwd", tile=128)
    assert torch.equal(u, u_old) or rel_err(u, u_old) < 4e-3
    assert float((f.float() - f_old.float()).abs().max()) <= 2.0 ** -7 * float(f_old.float().abs().max())
    assert float((f != f_old).float().mean()) < 2e-2
    # FFN2 dgrad: dY . W2 with GELU' of the saved pre-activation (bf16 out)
    dY = gen(M, K, seed=4).to(DEV, dtype)                     # [T, 768]
    W2 = gen(K, N, seed=5, scale=0.05).to(DEV, dtype)         # nn.Linear(3072 -> 768) weight [768, 3072]: T-major B of the dgrad
    du = torch.empty(M, N, device=DEV, dtype=dtype)
    ops.gemm(dY, W2, M, N, K, trans_b=True, out16=du, aux=u, gelu="bwd", tile=256)
    uf = u.double().cpu()
    gp = 0.5 * (1.0 + torch.erf(uf / math.sqrt(2.0))) + uf * torch.exp(-0.5 * uf * uf) / math.sqrt(2.0 * math.pi)
    assert rel_err(du, (dY.double().cpu() @ W2.double().cpu()) * gp) < 6e-3
    # FFN1 dgrad: fp32 output + fp32 residual, unsplit and in three K slices (atomics into a zeroed output), alpha
    dU = gen(M, N, seed=6).to(DEV, dtype)
    R = gen(M, K, seed=7).to(DEV)
    ref = dU.double().cpu() @ W.double().cpu() * 0.5 + R.double().cpu()
    for ks in (1, 3, 2):
        dx = torch.zeros(M, K, device=DEV)
        ops.gemm(dU, W, M, K, N, trans_b=True, out32=dx, residual=R, ksplit=ks, alpha=0.5, tile=256)
        assert rel_err(dx, ref) < 2e-5, ks
    # accumulate + non-unit leading dimension of the output (a column block of a wider buffer)
    wide = torch.full((M, 2 * K), 0.25, device=DEV)
    ops.gemm(dU, W, M, K, N, trans_b=True, out32=wide[:, K:], accumulate=True, alpha=0.5, tile=256)
    assert rel_err(wide[:, K:], ref - R.double().cpu() + 0.25) < 2e-5
    assert float((wide[:, :K] - 0.25).abs().max()) == 0.0
    # products the body does not carry fall back to the older tiles (same results as asking for them)
    for (m, n, k, kw) in [(1500, 768, 768, {}), (1536, 700, 768, {}), (1536, 768, 704, {}), (1536, 768, 768, dict(ksplit=4))]:
        a = gen(m, k, seed=8).to(DEV, dtype)
        b = gen(n, k, seed=9).to(DEV, dtype)
        o1, o2 = torch.zeros(m, n, device=DEV), torch.zeros(m, n, device=DEV)
        ops.gemm(a, b, m, n, k, out32=o1, tile=256, **kw)
        ops.gemm(a, b, m, n, k, out32=o2, tile=128, **kw)
        assert rel_err(o1, a.double().cpu() @ b.double().cpu().T) < 2e-5
        if not kw:
            assert torch.equal(o1, o2)


def test_gemm256_gelu_and_gelu_grad_against_exact_erf_over_every_bf16_input():
    """gemm256.h evaluates erf by a branch-free Abramowitz-Stegun 7.1.26 form (__expf, rcp) instead of erff, so GELU / GELU'
    depend on which tile a shape selects.  Pin that form directly: the 256 body's GELU epilogues on EVERY bf16 pre-activation with
    |x| <= 8 (33 282 values, an identity product puts them into the accumulators exactly), and on fp32 pre-activations (the same grid plus
    an fp32 bias), against the exact erf in fp64.  The form's error is ABSOLUTE (|erf error| <= 1.5e-7, i.e. 7.5e-8 on Phi): where
    |gelu| is large against it the bf16 result must be the correctly rounded one up to one ulp on a small fraction of inputs; in the
    negative tail (x < -3.5, |gelu| < 1e-3) the bound is 2e-7 |x| absolute -- measured first contact: up to 9 bf16 ulps of a 1e-6
    result there, which the older tiles' erff does not show and no gate of a training step can see."""
    dtype = torch.bfloat16
    M = N = K = 256
    pos = torch.arange(0, 0x4100 + 1, dtype=torch.int32)                              # +0 .. 8.0 as bf16 bit patterns
    bits = torch.cat([pos, pos | 0x8000]).to(torch.int16)
    grid = torch.zeros(M * N, dtype=torch.int16)
    grid[:bits.numel()] = bits
    G = grid.view(torch.bfloat16).view(N, M).to(DEV)                                  # W[n, m]
    X = torch.eye(M, K, dtype=torch.float32).to(DEV, dtype)
    sqrt2 = math.sqrt(2.0)

    def exact(u):
        u = u.double().cpu()
        return u * 0.5 * (1.0 + torch.erf(u / sqrt2)), 0.5 * (1.0 + torch.erf(u / sqrt2)) + u * torch.exp(-0.5 * u * u) / math.sqrt(2.0 * math.pi)

    def check(out, ref64, x64, what):
        got = out.double().cpu()
        err = (got - ref64).abs()
        tol = 2.0 ** -8 * ref64.abs() + 2e-7 * x64.abs().clamp(min=1.0)              # one bf16 rounding + the form's absolute error
        bad = err > tol
        assert not bool(bad.any()), (what, int(bad.sum()), float(x64[bad][0]), float(got[bad][0]), float(ref64[bad][0]))
        # where the result is large against the absolute error: correctly rounded, up to one ulp on a small fraction
        want = ref64.to(torch.float32).to(torch.bfloat16)
        big = ref64.abs() > 1e-3
        a = out.cpu().view(torch.int16).to(torch.int32)[big]
        b = want.view(torch.int16).to(torch.int32)[big]
        a = torch.where(a < 0, -(a & 0x7fff), a)                                       # sign-magnitude -> ordered integers
        b = torch.where(b < 0, -(b & 0x7fff), b)
        d = (a - b).abs()
        assert int(d.max()) <= 1, (what, int(d.max()))
        assert float((d > 0).float().mean()) < 1e-2, (what, float((d > 0).float().mean()))

    for bias in (None, (torch.rand(N, generator=torch.Generator().manual_seed(5)) * 2 - 1).to(DEV)):
        f = torch.empty(M, N, device=DEV, dtype=dtype)
        u = torch.empty(M, N, device=DEV, dtype=dtype)
        ops.gemm(X, G, M, N, K, out16=f, aux=u, bias=bias, gelu="fwd", tile=256)
        pre = G.float().T + (bias if bias is not None else 0.0)                       # pre[m, n] = W[n, m] (+ bias[n]): exact in fp32
        gel, _ = exact(pre)
        check(f, gel, pre.double().cpu(), "gelu")
        assert torch.equal(u.float().cpu(), pre.to(torch.bfloat16).float().cpu())     # the saved pre-activation: correctly rounded
        # GELU' of the SAVED (bf16) pre-activation times an upstream gradient of exactly 1
        W2 = torch.ones(K, N, dtype=torch.float32).to(DEV, dtype)                     # one-hot rows of dY against an all-ones W2: every product 1
        dY = torch.zeros(M, K, device=DEV, dtype=dtype)
        dY[:, 0] = 1.0
        du = torch.empty(M, N, device=DEV, dtype=dtype)
        ops.gemm(dY, W2, M, N, K, trans_b=True, out16=du, aux=u, gelu="bwd", tile=256)
        _, gp = exact(u.float())
        check(du, gp, u.double().cpu(), "gelu'")


@pytest.mark.parametrize("T", [1536, 6144])
def test_gemm256_weight_gradient_group_with_bias_gradients_and_norms(T):
    """A layer's four weight gradients as ONE grouped launch on the 256 tile (what the backward plan issues at thousands of tokens):
    beta = 0 and accumulate members, the fused q/k/v member with three gradient-norm tensors, bias gradients taken by the column-sum
    roles in front of the tiles."""
    dtype = torch.bfloat16
    shapes = [("qkv", 2304, 768), ("o", 768, 768), ("ffn1", 3072, 768), ("ffn2", 768, 3072)]
    dY = {n_: gen(T, N, seed=20 + i).to(DEV, dtype) for i, (n_, N, K) in enumerate(shapes)}
    X = {n_: gen(T, K, seed=30 + i).to(DEV, dtype) for i, (n_, N, K) in enumerate(shapes)}
    dW = {n_: torch.full((N, K), 0.5, device=DEV) for n_, N, K in shapes}
    db = {n_: torch.zeros(N, device=DEV) for n_, N, K in shapes}
    stride = (768 // 64) * (768 // 64) * 4
    part = torch.zeros(3 * stride, device=DEV)
    descs = []
    for n_, N, K in shapes:
        kw = dict(trans_a=True, trans_b=True, out32=dW[n_], accumulate=(n_ == "o"))
        if n_ in ("qkv", "ffn1"):
            kw["dbias"] = db[n_]
        if n_ == "qkv":
            kw.update(sumsq=part, sumsq_rows=768, sumsq_stride=stride)
        descs.append(ops.gemm_desc(dY[n_], X[n_], N, K, T, **kw))
    ops.gemm_group(descs)
    for n_, N, K in shapes:
        ref = dY[n_].double().cpu().T @ X[n_].double().cpu() + (0.5 if n_ == "o" else 0.0)
        assert rel_err(dW[n_], ref) < 2e-5, n_
        if n_ in ("qkv", "ffn1"):
            assert rel_err(db[n_], dY[n_].double().cpu().sum(0)) < 1e-5, n_
    assert rel_err(part.view(3, stride).sum(1), (dW["qkv"].double() ** 2).view(3, -1).sum(1).cpu()) < 1e-5
    # ... and it IS the 256 body: a forced 128-tile group gives the same numbers only to round-off, not bit for bit
    dW2 = {n_: torch.full((N, K), 0.5, device=DEV) for n_, N, K in shapes}
    ops.gemm_group([ops.gemm_desc(dY[n_], X[n_], N, K, T, trans_a=True, trans_b=True, out32=dW2[n_], accumulate=(n_ == "o"), tile=128,
                                  stages=2, waves=4) for n_, N, K in shapes])
    assert all(rel_err(dW[n_], dW2[n_]) < 2e-5 for n_, _, _ in shapes)
    assert not all(torch.equal(dW[n_], dW2[n_]) for n_, _, _ in shapes)


def test_gemm256_under_concurrent_hbm_traffic_is_bit_stable():
    """The loop keeps three half-tiles of LDS-DMA in flight across its barriers and reads a stage one phase after the counted wait that
    retires it (gemm256.h).  A fragment read that overtakes its DMA returns the previous K tile's bytes -- only when the DMA is slow.
    So: a side stream saturates HBM with copies of UNEVEN size while 40 launches per layout run; every launch must reproduce the quiet
    run bit for bit (checked on every word)."""
    dtype = torch.bfloat16
    M, N, K = 1536, 2304, 768
    side = torch.cuda.Stream()
    big = [torch.empty(s, device=DEV, dtype=torch.uint8) for s in (1 << 28, 1 << 28)]
    small = [torch.empty(s, device=DEV, dtype=torch.uint8) for s in (1 << 20, 1 << 20)]
    for layout in ("fwd", "dgrad", "wgrad"):
        ta, tb = layout == "wgrad", layout != "fwd"
        A = gen(*((K, M) if ta else (M, K)), seed=41).to(DEV, dtype)
        B = gen(*((K, N) if tb else (N, K)), seed=42).to(DEV, dtype)
        quiet = torch.empty(M, N, device=DEV)
        ops.gemm(A, B, M, N, K, trans_a=ta, trans_b=tb, out32=quiet, tile=256)
        assert rel_err(quiet, _ref64(A, B, ta, tb)) < 2e-5
        torch.cuda.synchronize()
        outs = [torch.empty(M, N, device=DEV) for _ in range(40)]
        with torch.cuda.stream(side):
            for r in range(60):
                big[1].copy_(big[0])
                if r % 3 == 0:
                    small[1].copy_(small[0])
        for o in outs:
            ops.gemm(A, B, M, N, K, trans_a=ta, trans_b=tb, out32=o, tile=256)
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            assert torch.equal(o, quiet), (layout, i)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_fused_sum_of_squares(dtype):
    """UnivlGemm.sumsq: the wgrad epilogue accumulates the per-tensor sum of squares of what it stores (single tensor,
    fused q/k/v group with one accumulator per 768 rows, accumulate mode = squares of the FINAL values)."""
    T = 200
    dY = gen(T, 2304, seed=1).to(DEV, dtype)
    X = gen(T, 768, seed=2).to(DEV, dtype)
    out = torch.full((2304, 768), 0.5, device=DEV)
    stride = (768 // 64) * (768 // 64) * 4
    part = torch.zeros(3 * stride + 8, device=DEV)
    for tile in (64, 128):
        part.zero_()
        out.fill_(0.5)
        ops.gemm(dY, X, 2304, 768, T, trans_a=True, trans_b=True, out32=out, accumulate=True, sumsq=part, sumsq_rows=768,
                 sumsq_stride=stride, tile=tile)
        ref = (out.double() ** 2).view(3, -1).sum(1).cpu()
        assert float(part[3 * stride:].abs().max()) == 0.0
        assert rel_err(part[:3 * stride].view(3, stride).sum(1), ref) < 1e-5
        seg = torch.tensor([4, 0, 2], dtype=torch.int32, device=DEV)
        start = torch.tensor([0, stride, 2 * stride], dtype=torch.int32, device=DEV)
        count = torch.full((3,), stride, dtype=torch.int32, device=DEV)
        res = torch.full((5,), -1.0, device=DEV)
        ops.sumsq_finish(part, seg, start, count, res)
        assert rel_err(res[[4, 0, 2]], ref) < 1e-5 and float(res[1]) == -1.0 and float(res[3]) == -1.0
    one = torch.zeros(stride, device=DEV)
    o2 = torch.zeros(768, 768, device=DEV)
    d0 = ops.gemm_desc(dY[:, :768].contiguous(), X, 768, 768, T, trans_a=True, trans_b=True, out32=o2, sumsq=one)
    ops.gemm_group([d0, ops.gemm_desc(dY, X, 2304, 768, T, trans_a=True, trans_b=True, out32=out)])
    assert rel_err(one.sum().reshape(1), (o2.double() ** 2).sum().cpu().reshape(1)) < 1e-5
    with pytest.raises(RuntimeError):
        ops.gemm(dY, X, 2304, 768, T, trans_a=True, trans_b=True, out32=out, sumsq=part, ksplit=2)


def test_gemm_argument_errors():
    A = torch.zeros(8, 256, device=DEV)
    B = torch.zeros(8, 256, device=DEV)
    out = torch.zeros(8, 8, device=DEV)
    with pytest.raises(RuntimeError):
        ops.gemm(A[:, 1:], B[:, 1:], 8, 8, 8, out32=out)          # pointer not 16-byte aligned
    with pytest.raises(RuntimeError):
        ops.gemm(A, B, 8, 8, 256, out16=out, ksplit=2)            # split-K needs fp32 output
    with pytest.raises(RuntimeError):
        ops.gemm(A, B, 8, 8, 256, out32=out, gelu="fwd")          # GELU epilogue without aux
    with pytest.raises(RuntimeError):
        ops.gemm(A.cpu(), B.cpu(), 8, 8, 256, out32=out)          # CPU tensors: no fallback


# --------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,rows", [(768, 192), (768, 61), (1024, 100)])
def test_layernorm_fwd_bwd(dtype, N, rows):
    dt = ops.dtype_code(dtype)
    x, res = gen(rows, N, seed=1), gen(rows, N, seed=2)
    period = 7
    pos = gen(period, N, seed=3)
    gamma, beta = 1 + 0.1 * gen(N, seed=4), 0.1 * gen(N, seed=5)
    dout = gen(rows, N, seed=6)
    xd = x.to(DEV)
    y = torch.empty(rows, N, device=DEV); stats = torch.empty(rows, 2, device=DEV)
    out32 = torch.empty(rows, N, device=DEV); out16 = torch.empty(rows, N, device=DEV, dtype=dtype)
    ops.layernorm_fwd(dtype=dt, rows=rows, N=N, x=xd, residual=res.to(DEV), pos=pos.to(DEV), pos_period=period,
                      gamma=gamma.to(DEV), beta=beta.to(DEV), y=y, stats=stats, out32=out32, out16=out16)
    xr = x.double().requires_grad_(True); rr = res.double().requires_grad_(True); pr = pos.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True); br = beta.double().requires_grad_(True)
    rowpos = pr[torch.arange(rows) % period]
    ref = O.layer_norm(xr + rr + rowpos, gr, br)
    assert rel_err(out32, ref) < 1e-5
    assert rel_err(out16.float(), ref) < tol(dtype)
    ref.backward(dout.double())
    dx32 = torch.empty(rows, N, device=DEV); dxd16 = torch.empty(rows, N, device=DEV, dtype=dtype)
    dg = torch.zeros(N, device=DEV); db = torch.zeros(N, device=DEV); dbias = torch.zeros(N, device=DEV)
    dpos = torch.zeros(period, N, device=DEV)
    ops.layernorm_bwd(dtype=dt, rows=rows, N=N, gamma=gamma.to(DEV), y=y, stats=stats, dout=dout.to(DEV), dx32=dx32,
                      dxd16=dxd16, dgamma=dg, dbeta=db, dbias=dbias, dpos=dpos, pos_period=period)
    assert rel_err(dx32, xr.grad) < 1e-4
    assert rel_err(dxd16.float(), xr.grad) < tol(dtype)
    assert rel_err(dg, gr.grad) < 1e-4 and rel_err(db, br.grad) < 1e-4
    assert rel_err(dbias, xr.grad.sum(0)) < 1e-4
    assert rel_err(dpos, pr.grad) < 1e-4


def test_layernorm_f64_input_and_zero_rows():
    """NormalizeVideo (modeling.py:88-92): float64 in, all-zero padded frames must give exactly `bias`."""
    rows, N = 50, 1024
    x = gen(rows, N, seed=1).double()
    x[10:20] = 0.0
    gamma, beta = 1 + 0.1 * gen(N, seed=2), 0.1 * gen(N, seed=3)
    out = torch.empty(rows, N, device=DEV)
    ops.layernorm_fwd(dtype=_lib.DT_F32, rows=rows, N=N, x=x.to(DEV), x_f64=True, gamma=gamma.to(DEV), beta=beta.to(DEV), out32=out)
    ref = O.layer_norm(x.float(), gamma, beta)
    assert rel_err(out, ref) < 1e-5
    assert torch.equal(out[10:20].cpu(), beta.expand(10, N))


def test_layernorm_dropout_mask_consistency():
    """dropout masks regenerated in the backward equal the forward's (both p_pre and p_post); keep-rate ~ 1-p."""
    rows, N, p = 64, 768, 0.1
    x = gen(rows, N, seed=1).to(DEV)
    ones, zeros = torch.ones(N, device=DEV), torch.zeros(N, device=DEV)
    y = torch.empty(rows, N, device=DEV); stats = torch.empty(rows, 2, device=DEV); out = torch.empty(rows, N, device=DEV)
    kw = dict(dtype=_lib.DT_F32, rows=rows, N=N, gamma=ones, beta=zeros, p_pre=p, p_post=p, seed=5, off_pre=11, off_post=12)
    ops.layernorm_fwd(x=x, y=y, stats=stats, out32=out, **kw)
    kept_pre = (y != 0).float().mean().item()
    kept_post = (out != 0).float().mean().item()
    assert abs(kept_pre - 0.9) < 0.01 and abs(kept_post - 0.9) < 0.01
    assert rel_err(y[y != 0], (x / 0.9)[y != 0]) < 1e-6
    dout = torch.ones(rows, N, device=DEV)
    dxd = torch.empty(rows, N, device=DEV); dg = torch.zeros(N, device=DEV); db = torch.zeros(N, device=DEV)
    ops.layernorm_bwd(y=y, stats=stats, dout=dout, dxd32=dxd, dgamma=dg, dbeta=db, **kw)
    # dbeta = column sums of the post-dropout-masked upstream gradient: equals count of kept outputs / 0.9
    assert rel_err(db, (out != 0).float().sum(0) / 0.9) < 1e-5
    assert torch.equal(dxd == 0, y == 0) or ((dxd == 0) & (y != 0)).float().mean().item() < 1e-3


# ----------------------------------------------------------------------- product + LayerNorm in one launch (K8 / K10)
@pytest.mark.parametrize("M,K,ksplit,p_drop", [(192, 768, 2, 0.0), (192, 3072, 8, 0.1), (768, 768, 1, 0.1), (768, 3072, 3, 0.0),
                                                (100, 768, 2, 0.1), (1024, 768, 1, 0.0), (64, 3072, 8, 0.1), (576, 768, 2, 0.1), (624, 3072, 3, 0.0)])
def test_gemm_ln_fold_matches_the_two_launches(M, K, ksplit, p_drop):
    """univl_gemm_ln (gemm.hip: ln_fold): the LayerNorm behind an attention-output / FFN2 product finished by the product's own launch.
    Same arithmetic per row as univl_layernorm_fwd on the same fp32 sums: bit-identical to the two launches when the product is not
    split (plain stores), within fp32 summation-order noise when its slices meet in atomics; 40 back-to-back launches per case as a
    race screen (a workgroup normalising a row block before every contribution landed would show up as a wrong row), and the arrival
    counters must be zero again after every launch."""
    import ctypes as C
    import univl_amd
    was = univl_amd.deterministic()
    univl_amd.set_deterministic(False)
    try:
        N = 768
        bf = torch.bfloat16
        a = gen(M, K, seed=1).to(DEV, bf)
        w = gen(N, K, seed=2, scale=K ** -0.5).to(DEV, bf)
        bias = gen(N, seed=3).to(DEV)
        res = gen(M, N, seed=4).to(DEV)
        gm, bt = (1.0 + 0.1 * gen(N, seed=5)).to(DEV), gen(N, seed=6).to(DEV)

        def bufs():
            return dict(x=torch.zeros(M, N, device=DEV), stats=torch.zeros(M, 2, device=DEV), out32=torch.zeros(M, N, device=DEV),
                        out16=torch.zeros(M, N, device=DEV, dtype=bf))

        def descs(b):
            g = ops.gemm_desc(a, w, M, N, K, out32=b["x"], bias=bias, ksplit=ksplit)
            ln = ops.layernorm_desc(ops.dtype_code(bf), M, N, x=b["x"], residual=res, gamma=gm, beta=bt, y=b["x"], stats=b["stats"],
                                    out32=b["out32"], out16=b["out16"], p_pre=p_drop, seed=7, off_pre=3 << 40)
            return g, ln

        ref = bufs()
        g, ln = descs(ref)
        _lib.check(_lib.lib().univl_gemm(C.byref(g), None), "gemm")
        _lib.check(_lib.lib().univl_layernorm_fwd(C.byref(ln), None), "layernorm_fwd")
        ctr = torch.zeros(2 * ((M + 63) // 64), dtype=torch.int32, device=DEV)
        got = bufs()
        g2, ln2 = descs(got)
        assert ops.gemm_ln(g2, ln2, ctr, dry_run=True)
        for it in range(40):
            got["x"].zero_()
            assert ops.gemm_ln(g2, ln2, ctr)
            assert int(ctr.abs().sum()) == 0, it
            for k in ("x", "stats", "out32", "out16"):
                if ksplit == 1:
                    assert torch.equal(got[k], ref[k]), (it, k)
                else:
                    assert rel_err(got[k].float(), ref[k].float()) < (1e-2 if k == "out16" else 2e-5), (it, k)
        # deterministic mode: refused (callers enqueue the two launches)
        univl_amd.set_deterministic(True)
        assert not ops.gemm_ln(g2, ln2, ctr, dry_run=True)
    finally:
        univl_amd.set_deterministic(was)


@pytest.mark.parametrize("T,K_out,ks,p_drop", [(192, 3072, 8, 0.1), (192, 2304, 6, 0.0), (100, 3072, 8, 0.1), (320, 768, 2, 0.1), (64, 2304, 6, 0.0),
                                               # round 5: the rectangular dgrad body (384+ tokens, slices up to 1536 deep)
                                               (384, 3072, 2, 0.1), (512, 2304, 2, 0.0), (448, 3072, 8, 0.1), (512, 3072, 4, 0.1), (1024, 2304, 3, 0.0),
                                               # round 6: the plans fold up to 640 tokens
                                               (576, 3072, 3, 0.1), (624, 2304, 3, 0.0)])
def test_gemm_pair_ln_fold_matches_the_two_launches(T, K_out, ks, p_drop):
    """univl_gemm_pair_ln (gemm.hip: ln_fold_bwd): the LayerNorm BACKWARD fed by a pair launch's dgrad product, finished by the dgrad's
    last workgroups per 64-row block -- dx32 / dxd16 rows and the dgamma / dbeta / dbias column sums against univl_gemm_pair +
    univl_layernorm_bwd on the same inputs (fp32 summation-order noise only: the dgrad slices meet in atomics, the column sums are
    grouped by 8 rows instead of 4), the weight gradient bit for bit; 40 launches per case as a race screen, counters back at zero."""
    import univl_amd
    was = univl_amd.deterministic()
    univl_amd.set_deterministic(False)
    try:
        H = 768
        bf = torch.bfloat16
        dY = gen(T, K_out, seed=1).to(DEV, bf)
        W = gen(K_out, H, seed=2, scale=0.05).to(DEV, bf)
        X = gen(T, H, seed=3).to(DEV, bf)
        res = gen(T, H, seed=5).to(DEV)
        gm = (1.0 + 0.1 * gen(H, seed=6)).to(DEV)
        y = gen(T, H, seed=7).to(DEV)
        stats = torch.stack([y.mean(1), 1.0 / (y.var(1, unbiased=False) + 1e-12).sqrt()], 1).contiguous()

        def bufs():
            return dict(da=torch.zeros(T, H, device=DEV), dx32=torch.zeros(T, H, device=DEV), dxd16=torch.zeros(T, H, device=DEV, dtype=bf),
                        dgamma=torch.zeros(H, device=DEV), dbeta=torch.zeros(H, device=DEV), dbias=torch.zeros(H, device=DEV),
                        dW=torch.zeros(K_out, H, device=DEV), db=torch.zeros(K_out, device=DEV))

        def descs(b):
            dg = ops.gemm_desc(dY, W, T, H, K_out, trans_b=True, out32=b["da"], residual=res, ksplit=ks)
            wg = ops.gemm_desc(dY, X, K_out, H, T, trans_a=True, trans_b=True, out32=b["dW"], dbias=b["db"])
            ln = ops.layernorm_desc(ops.dtype_code(bf), T, H, gamma=gm, y=y, stats=stats, dout=b["da"], dx32=b["dx32"], dxd16=b["dxd16"],
                                    dgamma=b["dgamma"], dbeta=b["dbeta"], dbias=b["dbias"], p_pre=p_drop, seed=9, off_pre=2 << 40)
            return dg, wg, ln

        ref = bufs()
        dg, wg, ln = descs(ref)
        assert ops.gemm_pair(dg, wg)
        _lib.check(_lib.lib().univl_layernorm_bwd(ops._BYREF(ln), ops._stream()), "layernorm_bwd")
        torch.cuda.synchronize()
        ctr = torch.zeros(2 * ((T + 63) // 64), dtype=torch.int32, device=DEV)
        got = bufs()
        dg2, wg2, ln2 = descs(got)
        assert ops.gemm_pair_ln(dg2, wg2, ln2, ctr, dry_run=True)
        for it in range(40):
            for k in ("da", "dgamma", "dbeta", "dbias", "dW", "db"):
                got[k].zero_()
            assert ops.gemm_pair_ln(dg2, wg2, ln2, ctr)
            assert int(ctr.abs().sum()) == 0, it
            assert torch.equal(got["dW"], ref["dW"]), it
            for k, t in (("da", 2e-5), ("dx32", 1e-4), ("dxd16", 1e-2), ("dgamma", 1e-4), ("dbeta", 1e-4), ("dbias", 1e-4), ("db", 1e-5)):
                assert rel_err(got[k].float(), ref[k].float()) < t, (it, k, rel_err(got[k].float(), ref[k].float()))
        univl_amd.set_deterministic(True)
        assert not ops.gemm_pair_ln(dg2, wg2, ln2, ctr, dry_run=True)
    finally:
        univl_amd.set_deterministic(was)


def _cu_masked_stream(every=8):
    """A HIP stream restricted to every `every`-th compute unit of the 256 (hipExtStreamCreateWithCUMask; on a multi-XCD device in SPX
    mode consecutive mask bits go round-robin over the XCDs, so bits 0, 8, 16, ... are the 32 units of ONE XCD).  None if unavailable."""
    import ctypes as C
    try:
        hip = C.CDLL("libamdhip64.so")
        st = C.c_void_p()
        word = sum(1 << b for b in range(0, 32, every))
        mask = (C.c_uint32 * 8)(*([word] * 8))
        if hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, mask) != 0 or not st.value:
            return None
        return torch.cuda.ExternalStream(st.value)
    except Exception:   # noqa: BLE001
        return None


@pytest.mark.parametrize("load", ["hbm_bursts", "one_xcd_saturated"])
def test_layernorm_folds_under_concurrent_hbm_traffic(load):
    """The folds' hand-over has no fence: contributions are fp32 atomics, "done" is s_waitcnt vmcnt(0) before the ticket, the rows come
    back through agent-scope loads.  Stress form of the race screens above: 400 forward and 400 backward fold launches at the 4-pair
    shapes
      hbm_bursts:         while a second stream saturates HBM with 256 MB copies (what the riding optimizer chunks and the other encoder
                          branch do to these launches in the step), every 4th launch compared with the two-launch reference;
      one_xcd_saturated:  UNEVEN load (round 5) -- a CU-masked stream keeps the 32 compute units of one XCD busy with streaming kernels
                          for the whole test, so the fold's workgroups on that XCD run late and the finishing workgroups are more often
                          ones that arrived early elsewhere; EVERY output word of EVERY launch is compared (x, stats, out32, out16;
                          dx32, dxd16, dgamma, dbeta, dbias, dW, db)."""
    import univl_amd
    was = univl_amd.deterministic()
    univl_amd.set_deterministic(False)
    try:
        bf, H, T, I = torch.bfloat16, 768, 192, 3072
        a = gen(T, I, seed=1).to(DEV, bf)
        w = gen(H, I, seed=2, scale=I ** -0.5).to(DEV, bf)
        bias, res = gen(H, seed=3).to(DEV), gen(T, H, seed=4).to(DEV)
        gm, bt = (1.0 + 0.1 * gen(H, seed=5)).to(DEV), gen(H, seed=6).to(DEV)
        dY = gen(T, I, seed=7).to(DEV, bf)
        W1 = gen(I, H, seed=8, scale=0.05).to(DEV, bf)
        X = gen(T, H, seed=9).to(DEV, bf)
        y = gen(T, H, seed=10).to(DEV)
        stats = torch.stack([y.mean(1), 1.0 / (y.var(1, unbiased=False) + 1e-12).sqrt()], 1).contiguous()

        def fwd_bufs():
            return dict(x=torch.zeros(T, H, device=DEV), stats=torch.zeros(T, 2, device=DEV), out32=torch.zeros(T, H, device=DEV),
                        out16=torch.zeros(T, H, device=DEV, dtype=bf))

        def fwd_descs(b):
            return (ops.gemm_desc(a, w, T, H, I, out32=b["x"], bias=bias, ksplit=8),
                    ops.layernorm_desc(ops.dtype_code(bf), T, H, x=b["x"], residual=res, gamma=gm, beta=bt, y=b["x"], stats=b["stats"],
                                       out32=b["out32"], out16=b["out16"], p_pre=0.1, seed=7, off_pre=3 << 40))

        def bwd_bufs():
            return dict(da=torch.zeros(T, H, device=DEV), dx32=torch.zeros(T, H, device=DEV), dxd16=torch.zeros(T, H, device=DEV, dtype=bf),
                        dgamma=torch.zeros(H, device=DEV), dbeta=torch.zeros(H, device=DEV), dbias=torch.zeros(H, device=DEV),
                        dW=torch.zeros(I, H, device=DEV), db=torch.zeros(I, device=DEV))

        def bwd_descs(b):
            return (ops.gemm_desc(dY, W1, T, H, I, trans_b=True, out32=b["da"], residual=res, ksplit=8),
                    ops.gemm_desc(dY, X, I, H, T, trans_a=True, trans_b=True, out32=b["dW"], dbias=b["db"]),
                    ops.layernorm_desc(ops.dtype_code(bf), T, H, gamma=gm, y=y, stats=stats, dout=b["da"], dx32=b["dx32"], dxd16=b["dxd16"],
                                       dgamma=b["dgamma"], dbeta=b["dbeta"], dbias=b["dbias"], p_pre=0.1, seed=9, off_pre=2 << 40))

        rf = fwd_bufs()
        g, ln = fwd_descs(rf)
        _lib.check(_lib.lib().univl_gemm(ops._BYREF(g), ops._stream()), "gemm")
        _lib.check(_lib.lib().univl_layernorm_fwd(ops._BYREF(ln), ops._stream()), "layernorm_fwd")
        rb = bwd_bufs()
        dg, wg, lb = bwd_descs(rb)
        assert ops.gemm_pair(dg, wg)
        _lib.check(_lib.lib().univl_layernorm_bwd(ops._BYREF(lb), ops._stream()), "layernorm_bwd")
        torch.cuda.synchronize()
        gf, gb = fwd_bufs(), bwd_bufs()
        g2, ln2 = fwd_descs(gf)
        dg2, wg2, lb2 = bwd_descs(gb)
        cf = torch.zeros(2 * ((T + 63) // 64), dtype=torch.int32, device=DEV)
        cb = torch.zeros_like(cf)
        big_a, big_b = torch.empty(64 << 20, device=DEV), torch.empty(64 << 20, device=DEV)      # 256 MB each
        uneven = load == "one_xcd_saturated"
        side = _cu_masked_stream() if uneven else torch.cuda.Stream()
        if side is None:
            pytest.skip("hipExtStreamCreateWithCUMask unavailable")
        side.wait_stream(torch.cuda.current_stream())
        worst = dict(out32=0.0, dx32=0.0, dgamma=0.0)
        for it in range(400):
            if uneven:
                with torch.cuda.stream(side):            # 32 compute units stream 0.5 GB per iteration: never idle while the folds run
                    big_b.copy_(big_a)
                    big_a.mul_(1.0)
            elif it % 8 == 0:
                with torch.cuda.stream(side):
                    for _ in range(6):
                        big_b.copy_(big_a)                                                            # ~3 GB of HBM traffic per burst
            gf["x"].zero_()
            assert ops.gemm_ln(g2, ln2, cf)
            for k in ("da", "dgamma", "dbeta", "dbias", "dW", "db"):
                gb[k].zero_()
            assert ops.gemm_pair_ln(dg2, wg2, lb2, cb)
            if uneven:
                assert int(cf.abs().sum()) == 0 and int(cb.abs().sum()) == 0, it
                for k in ("x", "stats", "out32", "out16"):
                    assert rel_err(gf[k].float(), rf[k].float()) < (1e-2 if k == "out16" else 2e-5), (it, k)
                for k in ("dx32", "dxd16", "dgamma", "dbeta", "dbias", "db"):
                    assert rel_err(gb[k].float(), rb[k].float()) < (1e-2 if k == "dxd16" else 1e-4), (it, k)
                assert torch.equal(gb["dW"], rb["dW"]), it
                worst["out32"] = max(worst["out32"], rel_err(gf["out32"], rf["out32"]))
                worst["dx32"] = max(worst["dx32"], rel_err(gb["dx32"], rb["dx32"]))
                worst["dgamma"] = max(worst["dgamma"], rel_err(gb["dgamma"], rb["dgamma"]))
            elif it % 4 == 3:
                worst["out32"] = max(worst["out32"], rel_err(gf["out32"], rf["out32"]))
                worst["dx32"] = max(worst["dx32"], rel_err(gb["dx32"], rb["dx32"]))
                worst["dgamma"] = max(worst["dgamma"], rel_err(gb["dgamma"], rb["dgamma"]))
                assert int(cf.abs().sum()) == 0 and int(cb.abs().sum()) == 0, it
                assert worst["out32"] < 2e-5 and worst["dx32"] < 1e-4 and worst["dgamma"] < 1e-4, (it, worst)
                assert torch.equal(gb["dW"], rb["dW"]), it
        side.synchronize()
        print("[fold stress, %s] worst relative errors over 400 + 400 launches under load:" % load, worst)
    finally:
        univl_amd.set_deterministic(was)


# --------------------------------------------------------------------------------------------- attention
ATTN_CASES = [(2, 48, 48, False), (3, 20, 20, False), (2, 12, 12, False), (2, 96, 96, False), (2, 40, 56, False),
              (1, 128, 224, False), (2, 24, 24, True), (1, 128, 128, True), (1, 224, 224, False)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Sq,Sk,causal", ATTN_CASES)
def test_attention_fwd_bwd(dtype, B, Sq, Sk, causal):
    H, D = 12, 64
    dt = ops.dtype_code(dtype)
    qkv_q = gen(B * Sq, H * D, seed=1).to(DEV, dtype)
    kv = gen(B * Sk, 2 * H * D, seed=2).to(DEV, dtype)
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(1, Sk + 1, (B,), generator=g)
    lens[0] = Sk
    mask = (torch.arange(Sk)[None] < lens[:, None]).long()
    if B > 1 and not causal:
        mask[1] = 0                                     # a fully masked row (reference: finite, not NaN)
    out = torch.zeros(B * Sq, H * D, device=DEV, dtype=dtype)
    lse = torch.zeros(B, H, Sq, device=DEV)
    args = (dt, B, H, Sq, Sk, qkv_q, H * D, (kv, 0), 2 * H * D, (kv, H * D), 2 * H * D, out, H * D, lse)
    ops.attention_fwd(*args, key_mask=mask.to(DEV), causal=causal)
    q = qkv_q.double().cpu().view(B, Sq, H * D).requires_grad_(True)
    k = kv.double().cpu()[:, :H * D].reshape(B, Sk, H * D).requires_grad_(True)
    v = kv.double().cpu()[:, H * D:].reshape(B, Sk, H * D).requires_grad_(True)
    add = O.extended_mask(mask, torch.float64)
    if causal:
        sub = torch.triu(torch.ones(Sq, Sk, dtype=torch.float64), diagonal=1)
        add = ((1.0 - mask[:, None, None, :].double()) + sub[None, None]).gt(0).double() * -10000.0
    ref = O.attention_core(q, k, v, add, H)
    assert rel_err(out.float().view(B, Sq, -1), ref) < tol(dtype)
    dout = gen(B * Sq, H * D, seed=4).to(DEV, dtype)
    ref.backward(dout.double().cpu().view(B, Sq, -1))
    dq = torch.zeros_like(qkv_q); dkv = torch.zeros_like(kv)
    ops.attention_bwd(*args, key_mask=mask.to(DEV), causal=causal, dout=dout, lddo=H * D, dq=dq, lddq=H * D,
                      dk=(dkv, 0), lddk=2 * H * D, dv=(dkv, H * D), lddv=2 * H * D)
    t = tol(dtype) * (2 if dtype == torch.bfloat16 else 1)
    assert rel_err(dq.float().view(B, Sq, -1), q.grad) < t
    assert rel_err(dkv.float()[:, :H * D].reshape(B, Sk, -1), k.grad) < t
    assert rel_err(dkv.float()[:, H * D:].reshape(B, Sk, -1), v.grad) < t


@pytest.mark.parametrize("B,S,p_drop,masked", [(4, 48, 0.1, False), (4, 48, 0.0, True), (3, 20, 0.1, True), (2, 64, 0.0, False), (5, 33, 0.1, True)])
def test_attention_bwd_with_the_output_dgrad_inside_equals_the_two_launches(B, S, p_drop, masked):
    """univl_attention_bwd_fused (round 5): every workgroup of the attention backward first multiplies the 64 x 64 block of
    dctx = dY . W_o that belongs to its (batch row, head) and works on it from LDS.  Same chunk order per output element, same bf16
    rounding of dctx as the dgrad's epilogue: dq / dk / dv are BIT-IDENTICAL to univl_gemm + univl_attention_bwd, with ragged key
    masks, a fully masked row, dropout regenerated from the seed, sequences that are not multiples of 16 -- and so is the weight
    gradient that rides in the launch (with its gradient-norm partials) to the product alone."""
    dtype, H, D = torch.bfloat16, 12, 64
    dt = ops.dtype_code(dtype)
    T, HD = B * S, H * D
    qkv = gen(T, 3 * HD, seed=1).to(DEV, dtype)
    g = torch.Generator().manual_seed(7)
    lens = torch.randint(1, S + 1, (B,), generator=g)
    lens[0] = S
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    if B > 1:
        mask[1] = 0
    km = mask.to(DEV) if masked else None
    ctx = torch.zeros(T, HD, device=DEV, dtype=dtype)
    lse = torch.zeros(B, H, S, device=DEV)
    seed = torch.full((1,), 1234, dtype=torch.int64, device=DEV)
    args = (dt, B, H, S, S, (qkv, 0), 3 * HD, (qkv, HD), 3 * HD, (qkv, 2 * HD), 3 * HD, ctx, HD, lse)
    kw = dict(key_mask=km, p_drop=p_drop, offset=5 << 40, seed_dev=seed)
    ops.attention_fwd(*args, **kw)
    dY = gen(T, HD, seed=2).to(DEV, dtype)                         # gradient wrt the attention-output projection's output
    Wo = gen(HD, HD, seed=3, scale=0.05).to(DEV, dtype)            # nn.Linear(768, 768).weight [out, in]
    # ---- the two launches
    dctx = torch.zeros(T, HD, device=DEV, dtype=dtype)
    ops.gemm(dY, Wo, T, HD, HD, trans_b=True, out16=dctx)
    dqkv = torch.zeros(T, 3 * HD, device=DEV, dtype=dtype)
    bw = dict(dout=dctx, lddo=HD, dq=(dqkv, 0), lddq=3 * HD, dk=(dqkv, HD), lddk=3 * HD, dv=(dqkv, 2 * HD), lddv=3 * HD)
    ops.attention_bwd(*args, **kw, **bw)
    assert float(dqkv.float().abs().max()) > 0
    gW = torch.zeros(HD, HD, device=DEV)
    stride = (HD // 64) * (HD // 64) * 4
    part = torch.zeros(stride, device=DEV)
    ops.gemm(dY, ctx, HD, HD, T, trans_a=True, trans_b=True, out32=gW, sumsq=part, sumsq_stride=stride)
    # ---- one launch: dctx never exists in global memory (the buffer named as dout stays untouched)
    dctx2 = torch.full((T, HD), 7.0, device=DEV, dtype=dtype)
    dqkv2 = torch.zeros(T, 3 * HD, device=DEV, dtype=dtype)
    bw2 = dict(dout=dctx2, lddo=HD, dq=(dqkv2, 0), lddq=3 * HD, dk=(dqkv2, HD), lddk=3 * HD, dv=(dqkv2, 2 * HD), lddv=3 * HD)
    at = ops.attention_desc(*args, **kw, **bw2)
    od = ops.gemm_desc(dY, Wo, T, HD, HD, trans_b=True, out16=dctx2)
    gW2 = torch.zeros(HD, HD, device=DEV)
    part2 = torch.zeros(stride, device=DEV)
    ow = ops.gemm_desc(dY, ctx, HD, HD, T, trans_a=True, trans_b=True, out32=gW2, sumsq=part2, sumsq_stride=stride)
    assert ops.attention_bwd_fused(at, od, ow)
    assert torch.equal(dqkv2, dqkv), float((dqkv2.float() - dqkv.float()).abs().max())
    assert torch.equal(gW2, gW)
    assert rel_err(part2.sum().reshape(1), (gW.double() ** 2).sum().cpu().reshape(1)) < 1e-5
    assert float((dctx2.float() - 7.0).abs().max()) == 0.0
    # ... and without a riding weight gradient; repeated launches reproduce the bits (the product's stages and the attention images
    # share LDS: a missing barrier between them shows up as run-to-run differences)
    for _ in range(5):
        dqkv2.zero_()
        assert ops.attention_bwd_fused(at, od, None)
        assert torch.equal(dqkv2, dqkv)
    # sequences beyond 64 positions, fp32, a split product: not carried
    at_long = ops.attention_desc(dt, 1, H, 80, 80, (qkv, 0), 3 * HD, (qkv, HD), 3 * HD, (qkv, 2 * HD), 3 * HD, ctx, HD, lse, **bw2)
    assert not ops.attention_bwd_fused(at_long, ops.gemm_desc(dY, Wo, 80, HD, HD, trans_b=True, out16=dctx2), None, dry_run=True)


@pytest.mark.parametrize("B,S,p_drop,masked", [(4, 48, 0.1, False), (4, 48, 0.0, True), (3, 20, 0.1, True), (2, 64, 0.0, False), (5, 33, 0.1, True),
                                               (4, 128, 0.1, False), (4, 96, 0.0, True), (3, 100, 0.1, True), (2, 65, 0.0, True), (6, 112, 0.1, False)])
def test_attention_fwd_with_the_qkv_projection_inside_equals_the_two_launches(B, S, p_drop, masked):
    """univl_attention_fwd_fused (round 5): the workgroup of a (batch row, head) multiplies its 64 x 192 block of q | k | v, stores it
    and attends on it from LDS.  The qkv buffer, the attention output and the log-sum-exp are BIT-IDENTICAL to univl_gemm +
    univl_attention_fwd (ragged key masks, a fully masked row, dropout, sequences that are not multiples of 16); rows of the qkv buffer
    that belong to no sequence position are not touched.  Round 6: sequences of 65 .. 128 positions run as TWO workgroups per (batch row,
    head) -- one per block of 64 queries, each multiplying the whole sequence's q | k | v -- with the same bits."""
    dtype, H, D = torch.bfloat16, 12, 64
    dt = ops.dtype_code(dtype)
    T, HD = B * S, H * D
    x = gen(T, HD, seed=1).to(DEV, dtype)
    W = gen(3 * HD, HD, seed=2, scale=0.05).to(DEV, dtype)
    bias = gen(3 * HD, seed=3).to(DEV)
    g = torch.Generator().manual_seed(7)
    lens = torch.randint(1, S + 1, (B,), generator=g)
    lens[0] = S
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    if B > 1:
        mask[1] = 0
    km = mask.to(DEV) if masked else None
    seed = torch.full((1,), 4321, dtype=torch.int64, device=DEV)
    kw = dict(key_mask=km, p_drop=p_drop, offset=3 << 40, seed_dev=seed)

    def run(fused):
        qkv = torch.full((T + 2, 3 * HD), 5.0, device=DEV, dtype=dtype)[:T]      # (two guard rows behind the buffer)
        ctx = torch.zeros(T, HD, device=DEV, dtype=dtype)
        lse = torch.zeros(B, H, S, device=DEV)
        args = (dt, B, H, S, S, (qkv, 0), 3 * HD, (qkv, HD), 3 * HD, (qkv, 2 * HD), 3 * HD, ctx, HD, lse)
        if fused:
            assert ops.attention_fwd_fused(ops.attention_desc(*args, **kw), ops.gemm_desc(x, W, T, 3 * HD, HD, out16=qkv, bias=bias))
        else:
            ops.gemm(x, W, T, 3 * HD, HD, out16=qkv, bias=bias)
            ops.attention_fwd(*args, **kw)
        torch.cuda.synchronize()
        return qkv.clone(), ctx, lse

    q0, c0, l0 = run(False)
    assert rel_err(q0.float(), x.double().cpu() @ W.double().cpu().T + bias.double().cpu()) < 6e-3
    for _ in range(4):
        q1, c1, l1 = run(True)
        assert torch.equal(q1, q0), float((q1.float() - q0.float()).abs().max())
        assert torch.equal(c1, c0), float((c1.float() - c0.float()).abs().max())
        assert torch.equal(l1, l0)
    # not carried: more than 128 positions, a causal mask
    qkv = torch.zeros(320, 3 * HD, device=DEV, dtype=dtype)
    ctx = torch.zeros(320, HD, device=DEV, dtype=dtype)
    lse = torch.zeros(2, H, 160, device=DEV)
    x320 = torch.zeros(320, HD, device=DEV, dtype=dtype)
    at = ops.attention_desc(dt, 2, H, 160, 160, (qkv, 0), 3 * HD, (qkv, HD), 3 * HD, (qkv, 2 * HD), 3 * HD, ctx, HD, lse)
    assert not ops.attention_fwd_fused(at, ops.gemm_desc(x320, W, 320, 3 * HD, HD, out16=qkv), dry_run=True)
    at = ops.attention_desc(dt, 2, H, 160, 160, (qkv, 0), 3 * HD, (qkv, HD), 3 * HD, (qkv, 2 * HD), 3 * HD, ctx, HD, lse, causal=True)
    assert not ops.attention_fwd_fused(at, ops.gemm_desc(x320, W, 320, 3 * HD, HD, out16=qkv), dry_run=True)


# ------------------------------------------------------------------------------------------ operand pairs (round 6)
def _pair(x32):
    """hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits in two bf16 tensors (include/univl_hip.h: UnivlGemm.A_lo)."""
    hi = x32.to(torch.bfloat16)
    return hi, (x32 - hi.float()).to(torch.bfloat16)


def _pair_ref(xh, xl, wh, wl):
    """what the launch computes: A.B + A.B_lo + A_lo.B (lo.lo dropped), in fp64"""
    xh, wh = xh.double().cpu(), wh.double().cpu()
    r = xh @ wh.T
    if wl is not None:
        r = r + xh @ wl.double().cpu().T
    if xl is not None:
        r = r + xl.double().cpu() @ wh.T
    return r


@pytest.mark.parametrize("M,N,K,ksplit,tile", [(192, 768, 768, 1, 0), (192, 768, 768, 5, 0), (192, 768, 3072, 8, 0), (100, 2304, 768, 1, 0),
                                               (700, 1000, 768, 3, 0), (768, 3072, 768, 1, 64128), (768, 768, 3072, 1, 128), (48, 768, 1024, 4, 0)])
@pytest.mark.parametrize("which", ["b", "a", "ab"])
def test_gemm_operand_pairs_forward(M, N, K, ksplit, tile, which):
    """A product whose operands are bf16 PAIRS walks the contraction once per term (A.B | A.B_lo | A_lo.B) into the same fp32
    accumulators: equal to the three products in fp64 to fp32 rounding, and -- with both operands paired -- to the product of the
    fp32 operands to 2^-16-ish, against 2^-8-ish for plain bf16.  Slices of a split contraction may start inside any term and cross
    term boundaries (ksplit 3 / 5 / 8 over two or three terms)."""
    x32, w32 = gen(M, K, seed=1).to(DEV), gen(N, K, seed=2, scale=0.05).to(DEV)
    bias = gen(N, seed=3).to(DEV)
    (xh, xl), (wh, wl) = _pair(x32), _pair(w32)
    a_lo = xl if "a" in which else None
    b_lo = wl if "b" in which else None
    Y = torch.zeros(M, N, device=DEV)
    ops.gemm(xh, wh, M, N, K, out32=Y, bias=bias, ksplit=ksplit, tile=tile, a_lo=a_lo, b_lo=b_lo)
    ref = _pair_ref(xh, a_lo, wh, b_lo) + bias.double().cpu()
    assert rel_err(Y, ref) < 3e-6
    exact = x32.double().cpu() @ w32.double().cpu().T + bias.double().cpu()
    Y0 = torch.zeros(M, N, device=DEV)
    ops.gemm(xh, wh, M, N, K, out32=Y0, bias=bias, ksplit=ksplit, tile=tile)
    e0, e1 = rel_err(Y0, exact), rel_err(Y, exact)
    assert e1 < (3e-5 if which == "ab" else e0), (e0, e1)


@pytest.mark.parametrize("T,N,K", [(192, 3072, 768), (60, 768, 3072)])
def test_gemm_operand_pairs_dgrad_gelu_and_paired_output(T, N, K):
    """T-major B with a lo half (the dgrad form), the GELU epilogue on a paired product, and the paired bf16 OUTPUT (C16_lo):
    out16 + out16_lo carries the fp32 result to ~2^-16."""
    x32, w32 = gen(T, K, seed=4).to(DEV), gen(N, K, seed=5, scale=0.05).to(DEV)
    b = gen(N, seed=6).to(DEV)
    (xh, xl), (wh, wl) = _pair(x32), _pair(w32)
    u = torch.zeros(T, N, device=DEV, dtype=torch.bfloat16)
    f, fl = torch.zeros_like(u), torch.zeros_like(u)
    ops.gemm(xh, wh, T, N, K, out16=f, out16_lo=fl, bias=b, aux=u, gelu="fwd", a_lo=xl, b_lo=wl)
    uref = x32.double().cpu() @ w32.double().cpu().T + b.double().cpu()
    assert rel_err(u.float(), uref) < 6e-3                         # the saved pre-activation is one bf16
    assert rel_err(f.float() + fl.float(), O.gelu(uref)) < 5e-5    # the activation leaves as a pair
    assert rel_err(f.float(), O.gelu(uref)) > 5e-4
    # dgrad: dX = dY . W with W = hi + lo read T-major
    dy32 = gen(T, N, seed=7).to(DEV)
    dyh, dyl = _pair(dy32)
    for a_lo, b_lo in ((None, wl), (dyl, wl)):
        dX = torch.zeros(T, K, device=DEV)
        ops.gemm(dyh, wh, T, K, N, trans_b=True, out32=dX, a_lo=a_lo, b_lo=b_lo, ksplit=2)
        ref = dyh.double().cpu() @ (wh.double().cpu() + wl.double().cpu())
        if a_lo is not None:
            ref = ref + dyl.double().cpu() @ wh.double().cpu()
        assert rel_err(dX, ref) < 3e-6


def test_gemm_operand_pairs_refusals():
    x, w = gen(64, 96, seed=1).to(DEV, torch.bfloat16), gen(64, 96, seed=2).to(DEV, torch.bfloat16)
    Y = torch.zeros(64, 64, device=DEV)
    with pytest.raises(RuntimeError):                      # K = 96 is not a multiple of the K step (128)
        ops.gemm(x, w, 64, 64, 96, out32=Y, b_lo=w)
    x32, w32 = gen(256, 256, seed=1).to(DEV), gen(256, 256, seed=2).to(DEV)
    with pytest.raises(RuntimeError):                      # fp32 mode has no pairs
        ops.gemm(x32, w32, 256, 256, 256, out32=torch.zeros(256, 256, device=DEV), b_lo=w32)
    # the 256 x 256 body never carries pairs: the same descriptor runs on the older tiles and is still correct
    xh, xl = _pair(gen(1536, 768, seed=3).to(DEV))
    wh, wl = _pair(gen(2304, 768, seed=4, scale=0.05).to(DEV))
    q = torch.zeros(1536, 2304, device=DEV, dtype=torch.bfloat16)
    ops.gemm(xh, wh, 1536, 2304, 768, out16=q, a_lo=xl, b_lo=wl, tile=256)
    assert rel_err(q.float(), _pair_ref(xh, xl, wh, wl)) < 5e-3


@pytest.mark.parametrize("B,S,p_drop,masked", [(4, 48, 0.1, False), (3, 20, 0.0, True), (2, 64, 0.1, False), (4, 128, 0.1, True), (3, 96, 0.0, False)])
@pytest.mark.parametrize("which", ["b", "ab"])
def test_attention_fwd_fused_with_operand_pairs_equals_the_two_launches(B, S, p_drop, masked, which):
    """The q | k | v projection inside the attention launch walks the same terms in the same order as univl_gemm with the same lo
    halves: qkv, context, context lo half and log-sum-exp are bit-identical to the two launches."""
    dtype, H, D = torch.bfloat16, 12, 64
    dt = ops.dtype_code(dtype)
    T, HD = B * S, H * D
    xh, xl = _pair(gen(T, HD, seed=1).to(DEV))
    wh, wl = _pair(gen(3 * HD, HD, seed=2, scale=0.05).to(DEV))
    bias = gen(3 * HD, seed=3).to(DEV)
    lens = torch.randint(1, S + 1, (B,), generator=torch.Generator().manual_seed(7))
    lens[0] = S
    km = (torch.arange(S)[None] < lens[:, None]).long().to(DEV) if masked else None
    seed = torch.full((1,), 4321, dtype=torch.int64, device=DEV)
    kw = dict(key_mask=km, p_drop=p_drop, offset=3 << 40, seed_dev=seed)
    a_lo = xl if "a" in which else None

    def run(fused):
        qkv = torch.zeros(T, 3 * HD, device=DEV, dtype=dtype)
        ctx, ctx_lo = torch.zeros(T, HD, device=DEV, dtype=dtype), torch.zeros(T, HD, device=DEV, dtype=dtype)
        lse = torch.zeros(B, H, S, device=DEV)
        args = (dt, B, H, S, S, (qkv, 0), 3 * HD, (qkv, HD), 3 * HD, (qkv, 2 * HD), 3 * HD, ctx, HD, lse)
        gd = ops.gemm_desc(xh, wh, T, 3 * HD, HD, out16=qkv, bias=bias, a_lo=a_lo, b_lo=wl)
        if fused:
            assert ops.attention_fwd_fused(ops.attention_desc(*args, out_lo=ctx_lo, **kw), gd)
        else:
            _lib.check(_lib.lib().univl_gemm(ops._BYREF(gd), ops._stream()), "gemm")
            ops.attention_fwd(*args, out_lo=ctx_lo, **kw)
        torch.cuda.synchronize()
        return qkv, ctx, ctx_lo, lse

    q0, c0, cl0, l0 = run(False)
    assert rel_err(q0.float(), _pair_ref(xh, a_lo, wh, wl) + bias.double().cpu()) < 6e-3
    assert float(cl0.float().abs().max()) > 0 and float((cl0.float().abs() > c0.float().abs() / 128 + 1e-30).sum()) == 0
    q1, c1, cl1, l1 = run(True)
    assert torch.equal(q1, q0) and torch.equal(c1, c0) and torch.equal(cl1, cl0) and torch.equal(l1, l0)


def test_layernorm_embedding_cast_and_adam_write_the_lo_half():
    """Every producer of a paired bf16 tensor: lo == bf16(fp32 value - hi) exactly."""
    dt = ops.dtype_code(torch.bfloat16)
    rows, N = 100, 768
    x, res = gen(rows, N, seed=1).to(DEV), gen(rows, N, seed=2).to(DEV)
    ga, be = (1 + 0.1 * gen(N, seed=3)).to(DEV), (0.1 * gen(N, seed=4)).to(DEV)
    y, st = torch.zeros(rows, N, device=DEV), torch.zeros(rows, 2, device=DEV)
    o32 = torch.zeros(rows, N, device=DEV)
    o16, olo = torch.zeros(rows, N, device=DEV, dtype=torch.bfloat16), torch.zeros(rows, N, device=DEV, dtype=torch.bfloat16)
    ops.layernorm_fwd(dtype=dt, rows=rows, N=N, x=x, residual=res, gamma=ga, beta=be, y=y, stats=st, out32=o32, out16=o16, out16_lo=olo)
    assert torch.equal(o16, o32.to(torch.bfloat16)) and torch.equal(olo, (o32 - o16.float()).to(torch.bfloat16))
    # text embeddings
    B, S, V = 3, 20, 500
    ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(5)).to(DEV)
    word, pos = gen(V, 768, seed=6).to(DEV), gen(64, 768, seed=7).to(DEV)
    e32 = torch.zeros(B * S, 768, device=DEV)
    e16, elo = torch.zeros(B * S, 768, device=DEV, dtype=torch.bfloat16), torch.zeros(B * S, 768, device=DEV, dtype=torch.bfloat16)
    ops.embed_text_fwd(dt, B, S, ids, word, pos, ga, be, y=torch.zeros(B * S, 768, device=DEV), stats=torch.zeros(B * S, 2, device=DEV),
                       out32=e32, out16=e16, out16_lo=elo)
    assert torch.equal(e16, e32.to(torch.bfloat16)) and torch.equal(elo, (e32 - e16.float()).to(torch.bfloat16))
    # the weight shadow pair
    p = gen(100003, seed=8).to(DEV)
    hi, lo = torch.zeros(100003, device=DEV, dtype=torch.bfloat16), torch.zeros(100003, device=DEV, dtype=torch.bfloat16)
    ops.cast_bf16_pair(p, hi, lo)
    assert torch.equal(hi, p.to(torch.bfloat16)) and torch.equal(lo, (p - hi.float()).to(torch.bfloat16))
    lo2 = torch.zeros_like(lo)
    ops.cast_bf16_pair(p, None, lo2)
    assert torch.equal(lo2, lo)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_maximum_sequence_and_limit(dtype):
    """Largest sequence the single-pass kernels hold in LDS (384 bf16 / 256 fp32) and the loud failure one past it."""
    H, D = 12, 64
    S = 384 if dtype == torch.bfloat16 else 256
    dt = ops.dtype_code(dtype)
    qkv = gen(S, 3 * H * D, seed=1).to(DEV, dtype)
    out = torch.zeros(S, H * D, device=DEV, dtype=dtype); lse = torch.zeros(H * S, device=DEV)
    a = (dt, 1, H, S, S, (qkv, 0), 3 * H * D, (qkv, H * D), 3 * H * D, (qkv, 2 * H * D), 3 * H * D)
    ops.attention_fwd(*a, out, H * D, lse)
    x = qkv.double().cpu()
    q, k, v = (x[:, i * H * D:(i + 1) * H * D].reshape(1, S, H * D).requires_grad_(True) for i in range(3))
    ref = O.attention_core(q, k, v, torch.zeros(1, 1, 1, S, dtype=torch.float64), H)
    assert rel_err(out.float().view(1, S, -1), ref) < tol(dtype)
    dout = gen(S, H * D, seed=2).to(DEV, dtype)
    ref.backward(dout.double().cpu().view(1, S, -1))
    dqkv = torch.zeros_like(qkv)
    ops.attention_bwd(*a, out, H * D, lse, dout=dout, lddo=H * D, dq=(dqkv, 0), lddq=3 * H * D, dk=(dqkv, H * D), lddk=3 * H * D,
                      dv=(dqkv, 2 * H * D), lddv=3 * H * D)
    t = tol(dtype) * (2 if dtype == torch.bfloat16 else 1)
    for i, g_ in enumerate((q.grad, k.grad, v.grad)):
        assert rel_err(dqkv.float()[:, i * H * D:(i + 1) * H * D].reshape(1, S, -1), g_) < t
    big = torch.zeros(S + 8, 3 * H * D, device=DEV, dtype=dtype)
    with pytest.raises(RuntimeError):
        ops.attention_fwd(dt, 1, H, S + 8, S + 8, (big, 0), 3 * H * D, (big, H * D), 3 * H * D, (big, 2 * H * D), 3 * H * D,
                          torch.zeros(S + 8, H * D, device=DEV, dtype=dtype), H * D, torch.zeros(H * (S + 8), device=DEV))


def test_attention_dropout_statistics():
    """p_drop = 0.1: E[out] matches the no-dropout output within sampling noise and fwd/bwd masks agree
    (checked through dV = P_drop^T dO with dO = 1, whose column sums equal sum of dropped probabilities)."""
    B, H, S, D = 2, 12, 48, 64
    qkv = gen(B * S, 3 * H * D, seed=1).to(DEV)
    out0 = torch.zeros(B * S, H * D, device=DEV); out1 = torch.zeros_like(out0); lse = torch.zeros(B, H, S, device=DEV)
    a = (_lib.DT_F32, B, H, S, S, (qkv, 0), 3 * H * D, (qkv, H * D), 3 * H * D, (qkv, 2 * H * D), 3 * H * D)
    ops.attention_fwd(*a, out0, H * D, lse)
    ops.attention_fwd(*a, out1, H * D, lse, p_drop=0.1, seed=7, offset=3)
    assert not torch.equal(out0, out1)
    assert abs(float((out1 - out0).mean())) < 0.01
    out2 = torch.zeros_like(out0)
    ops.attention_fwd(*a, out2, H * D, lse, p_drop=0.1, seed=7, offset=3)
    assert torch.equal(out1, out2)                      # same (seed, offset) -> same mask


# -------------------------------------------------------------------------------------------- embeddings
@pytest.mark.parametrize("dtype", DTYPES)
def test_embed_text_fwd_bwd(dtype):
    B, S, V = 3, 20, 500
    dt = ops.dtype_code(dtype)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, V, (B, S), generator=g)
    ids[0, :4] = 7                                        # repeated ids: scatter-add collisions
    tts = torch.randint(0, 2, (B, S), generator=g)
    word, pos, typ = gen(V, 768, seed=2, scale=0.02), gen(64, 768, seed=3, scale=0.02), gen(2, 768, seed=4, scale=0.02)
    gamma, beta = 1 + 0.1 * gen(768, seed=5), 0.1 * gen(768, seed=6)
    y = torch.empty(B * S, 768, device=DEV); stats = torch.empty(B * S, 2, device=DEV)
    out32 = torch.empty(B * S, 768, device=DEV); out16 = torch.empty(B * S, 768, device=DEV, dtype=dtype)
    a = (dt, B, S, ids.to(DEV), word.to(DEV), pos.to(DEV), gamma.to(DEV), beta.to(DEV))
    ops.embed_text_fwd(*a, type_ids=tts.to(DEV), type_emb=typ.to(DEV), y=y, stats=stats, out32=out32, out16=out16)
    wr, pr, tr = word.double().requires_grad_(True), pos.double().requires_grad_(True), typ.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = O.layer_norm(wr[ids] + pr[torch.arange(S)][None] + tr[tts], gr, br)
    assert rel_err(out32.view(B, S, -1), ref) < 1e-5
    assert rel_err(out16.float().view(B, S, -1), ref) < tol(dtype)
    dout = gen(B * S, 768, seed=7)
    ref.backward(dout.double().view(B, S, -1))
    dw, dp, dty = torch.zeros(V, 768, device=DEV), torch.zeros(64, 768, device=DEV), torch.zeros(2, 768, device=DEV)
    dg, db = torch.zeros(768, device=DEV), torch.zeros(768, device=DEV)
    ops.embed_text_bwd(*a, type_ids=tts.to(DEV), type_emb=typ.to(DEV), y=y, stats=stats, dout=dout.to(DEV), dword=dw,
                       dpos=dp, dtype_emb=dty, dgamma=dg, dbeta=db)
    for got, want in ((dw, wr.grad), (dp, pr.grad), (dty, tr.grad), (dg, gr.grad), (db, br.grad)):
        assert rel_err(got, want) < 1e-4


# ------------------------------------------------------------------------------------- pooling and losses
@pytest.mark.parametrize("B,S", [(5, 20), (3, 300), (4, 48)])
def test_pool_fwd_bwd(B, S):
    x = gen(B, S, 768, seed=1)
    g = torch.Generator().manual_seed(2)
    mask = (torch.arange(S)[None] < torch.randint(2, S + 1, (B, 1), generator=g)).long()
    vmask = mask.clone(); vmask[2] = 0                    # a video with zero valid frames (guarded count)
    x[1, -1] = float("inf")                               # garbage at a padded position must not leak (row 1 is masked there)
    mask[1, -1] = 0; vmask[1, -1] = 0
    xr = torch.nan_to_num(x, posinf=0.0).double().requires_grad_(True)
    t_ref, v_ref = O.mean_pooling_for_similarity(xr, xr, mask, vmask)
    t_ref, v_ref = torch.nn.functional.normalize(t_ref, dim=-1), torch.nn.functional.normalize(v_ref, dim=-1)
    dout = gen(B, 768, seed=3)
    for skip_first, m, ref in ((True, mask, t_ref), (False, vmask, v_ref)):
        mean = torch.empty(B, 768, device=DEV); out = torch.empty(B, 768, device=DEV); dx = torch.empty(B, S, 768, device=DEV)
        ops.pool_fwd(B, S, x.to(DEV), m.to(DEV), skip_first=skip_first, normalize=True, mean=mean, out=out)
        assert rel_err(out, ref) < 1e-5
        xr.grad = None
        ref.backward(dout.double(), retain_graph=True)
        ops.pool_bwd(B, S, x.to(DEV), m.to(DEV), skip_first=skip_first, normalize=True, mean=mean, out=out, dout=dout.to(DEV), dx=dx)
        assert rel_err(dx, xr.grad) < 1e-5
        base = gen(B, S, 768, seed=9).to(DEV)
        dx2 = base.clone()
        ops.pool_bwd(B, S, x.to(DEV), m.to(DEV), skip_first=skip_first, normalize=True, mean=mean, out=out, dout=dout.to(DEV), dx=dx2,
                     accumulate=True)
        assert rel_err(dx2 - base, xr.grad) < 1e-4


@pytest.mark.parametrize("rows,period,n", [(6144, 48, 768), (50, 7, 768), (3, 5, 1024), (128, 128, 768)])
def test_rows_gather_sum(rows, period, n):
    """univl_rows_gather_sum: out[s] += sum of rows[s::period] -- the position-table gradient from per-token rows."""
    x = gen(rows, n, seed=1)
    base = gen(period, n, seed=2)
    out = base.clone().to(DEV)
    ops.rows_gather_sum(x.to(DEV), period, out)
    ref = base.double().clone()
    for s_ in range(period):
        ref[s_] += x[s_::period].double().sum(0)
    assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("B,W,F", [(4, 48, 48), (6, 20, 300), (16, 48, 12)])
def test_pool_pair_launches_and_similarity_backward_folded_in(B, W, F):
    """univl_pool_pair_fwd / _bwd: the text and the video pooling of one similarity head in one launch each way; the backward takes
    its upstream gradient from d loss / d sim (UnivlPool.dsim: torch.matmul(text, video.t()) of modeling.py:389 and the upstream
    factor folded in).  Against the separate kernels (scale, two fp32 GEMMs, two pool backwards)."""
    xt, xv = gen(B, W, 768, seed=1).to(DEV), gen(B, F, 768, seed=2).to(DEV)
    g = torch.Generator().manual_seed(3)
    mt = (torch.arange(W)[None] < torch.randint(2, W + 1, (B, 1), generator=g)).long().to(DEV)
    mv = (torch.arange(F)[None] < torch.randint(0, F + 1, (B, 1), generator=g)).long().to(DEV)
    ldp = (B + 3) // 4 * 4
    dsim = torch.zeros(ldp, ldp, device=DEV); dsim[:B, :B] = gen(B, B, seed=4).to(DEV)
    gout = torch.tensor([0.5], device=DEV)
    e = lambda *s_: torch.empty(*s_, device=DEV)
    # reference: separate launches
    tm, tn, vm, vn = e(B, 768), torch.zeros(ldp, 768, device=DEV), e(B, 768), torch.zeros(ldp, 768, device=DEV)
    ops.pool_fwd(B, W, xt, mt, skip_first=True, normalize=True, mean=tm, out=tn)
    ops.pool_fwd(B, F, xv, mv, skip_first=False, normalize=True, mean=vm, out=vn)
    ds = dsim.clone()
    ops.scale_by_device_scalar(ds, gout)
    dtn, dvn = e(B, 768), e(B, 768)
    ops.gemm(ds, vn, B, 768, B, trans_b=True, out32=dtn)
    ops.gemm(ds, tn, B, 768, B, trans_a=True, trans_b=True, out32=dvn)
    base_t, base_v = gen(B, W, 768, seed=5).to(DEV), gen(B, F, 768, seed=6).to(DEV)
    dxt, dxv = base_t.clone(), base_v.clone()
    ops.pool_bwd(B, W, xt, mt, skip_first=True, normalize=True, mean=tm, out=tn, dout=dtn, dx=dxt, accumulate=True)
    ops.pool_bwd(B, F, xv, mv, skip_first=False, normalize=True, mean=vm, out=vn, dout=dvn, dx=dxv, accumulate=True)
    # pair launches
    tm2, tn2, vm2, vn2 = e(B, 768), torch.zeros(ldp, 768, device=DEV), e(B, 768), torch.zeros(ldp, 768, device=DEV)
    ops.pool_pair_fwd(ops.pool_desc(B, W, xt, mt, skip_first=True, normalize=True, mean=tm2, out=tn2),
                      ops.pool_desc(B, F, xv, mv, skip_first=False, normalize=True, mean=vm2, out=vn2))
    assert torch.equal(tn2, tn) and torch.equal(vn2, vn) and torch.equal(tm2, tm) and torch.equal(vm2, vm)
    dxt2, dxv2 = base_t.clone(), base_v.clone()
    ops.pool_pair_bwd(ops.pool_desc(B, W, xt, mt, skip_first=True, normalize=True, mean=tm2, out=tn2, dx=dxt2, accumulate=True,
                                    dsim=dsim, other=vn2, n_other=B, transpose=False, gscale=gout),
                      ops.pool_desc(B, F, xv, mv, skip_first=False, normalize=True, mean=vm2, out=vn2, dx=dxv2, accumulate=True,
                                    dsim=dsim, other=tn2, n_other=B, transpose=True, gscale=gout))
    # (dx - base cancels ~2 digits of the fp32 sums on both sides: the gate of the accumulate form, as in test_pool_fwd_bwd)
    assert rel_err(dxt2 - base_t, dxt - base_t) < 1e-4 and rel_err(dxv2 - base_v, dxv - base_v) < 1e-4


@pytest.mark.parametrize("M,N,K", [(4, 4, 768), (16, 16, 768), (32, 7, 1023), (1, 1, 5)])
def test_gemm_small_fp32_dot_path(M, N, K):
    """B x B similarity products (modeling.py:389) take the one-workgroup-per-output path."""
    Kp = (K + 3) // 4 * 4
    A = torch.zeros(M, Kp); A[:, :K] = gen(M, K, seed=1)
    B_ = torch.zeros(N, Kp); B_[:, :K] = gen(N, K, seed=2)
    out = torch.full((M, 40), 2.0, device=DEV)
    ops.gemm(A.to(DEV), B_.to(DEV), M, N, K, out32=out, alpha=0.5)
    ref = 0.5 * (A.double()[:, :K] @ B_.double()[:, :K].T)
    assert rel_err(out[:, :N], ref) < 1e-5
    assert float((out[:, N:] - 2.0).abs().max()) == 0.0
    ops.gemm(A.to(DEV), B_.to(DEV), M, N, K, out32=out, alpha=0.5, accumulate=True)
    assert rel_err(out[:, :N], 2 * ref) < 1e-5


@pytest.mark.parametrize("n", [4, 16, 37])
def test_maxmargin_and_crossen_losses(n):
    sim = gen(n, n, seed=1, scale=0.5)
    for name in ("maxmargin", "crossen"):
        s = sim.double().requires_grad_(True)
        ref = O.max_margin_ranking_loss(s, 0.1, 1, n, 1, 0.5) if name == "maxmargin" else O.cross_en(s)
        ref.backward()
        loss = torch.zeros(1, device=DEV); ds = torch.zeros(n, n, device=DEV)
        if name == "maxmargin":
            ops.maxmargin_loss(sim.to(DEV), 0.1, None, loss, ds)
        else:
            ops.crossen_loss(sim.to(DEV), loss, ds)
        assert abs(float(loss) - float(ref)) < 1e-5 * max(1, abs(float(ref)))
        assert rel_err(ds, s.grad) < 1e-5


@pytest.mark.parametrize("bs,npair", [(2, 3), (4, 1), (5, 2)])
def test_milnce_loss(bs, npair):
    n = bs * npair
    sim = gen(n, n, seed=2, scale=2.0)
    s = sim.double().requires_grad_(True)
    ref = O.mil_nce_loss(s, bs, npair)
    ref.backward()
    loss = torch.zeros(1, device=DEV); ds = torch.zeros(n, n, device=DEV)
    ops.milnce_loss(sim.to(DEV), bs, npair, loss, ds)
    assert abs(float(loss) - float(ref)) < 1e-5 * max(1, abs(float(ref)))
    assert rel_err(ds, s.grad) < 1e-5


def test_maxmargin_weighted():
    """negative_weighting with n_pair > 1 (until_module.py:238-243, 249-250)."""
    bs, npair = 3, 2
    n = bs * npair
    sim = gen(n, n, seed=3, scale=0.5)
    s = sim.double().requires_grad_(True)
    ref = O.max_margin_ranking_loss(s, 0.1, 1, bs, npair, 0.5)
    ref.backward()
    easy = 0.5
    alpha = easy / ((bs - 1) * (1 - easy))
    mm = np.kron((1 - alpha) * np.eye(bs) + alpha, np.ones((npair, npair))) * (bs * (1 - easy))
    w = torch.tensor(mm, dtype=torch.float32)
    loss = torch.zeros(1, device=DEV); ds = torch.zeros(n, n, device=DEV)
    ops.maxmargin_loss(sim.to(DEV), 0.1, w.to(DEV), loss, ds)
    assert abs(float(loss) - float(ref)) < 1e-5
    assert rel_err(ds, s.grad) < 1e-5


# ------------------------------------------------------------ cross encoder / classifier / pretraining helpers
def test_pair_concat_and_postype():
    """torch.cat((seq, vis), 1) over (text row, video row) pairs + concat_mask (modeling.py:315-325, 352-366) and the
    position + token-type table of CrossEmbeddings (module_cross.py:123-138); backward scatter-adds."""
    Bt, Bv, W, F = 3, 2, 5, 7
    seq, vis = gen(Bt, W, 768, seed=1), gen(Bv, F, 768, seed=2)
    am = (torch.arange(W)[None] < torch.tensor([[5], [2], [3]])).long()
    vm = (torch.arange(F)[None] < torch.tensor([[7], [1]])).long()
    pairs = [(i, j) for i in range(Bt) for j in range(Bv)]
    tidx = torch.tensor([a for a, _ in pairs], dtype=torch.int32, device=DEV)
    vidx = torch.tensor([b for _, b in pairs], dtype=torch.int32, device=DEV)
    P, S = len(pairs), W + F
    out = torch.zeros(P * S, 768, device=DEV); om = torch.zeros(P, S, dtype=torch.int64, device=DEV)
    ops.pair_concat_fwd(seq.to(DEV), vis.to(DEV), am.to(DEV), vm.to(DEV), tidx, vidx, P, W, F, out, om)
    ref = torch.stack([torch.cat((seq[a], vis[b]), 0) for a, b in pairs])
    refm = torch.stack([torch.cat((am[a], vm[b]), 0) for a, b in pairs])
    assert torch.equal(out.view(P, S, 768).cpu(), ref) and torch.equal(om.cpu(), refm)
    d = gen(P, S, 768, seed=3)
    dseq, dvis = torch.zeros(Bt * W, 768, device=DEV), torch.zeros(Bv * F, 768, device=DEV)
    ops.pair_concat_bwd(d.to(DEV).view(P * S, 768), tidx, vidx, P, W, F, dseq, dvis)
    rs = torch.zeros(Bt, W, 768, dtype=torch.float64); rv = torch.zeros(Bv, F, 768, dtype=torch.float64)
    for k, (a, b) in enumerate(pairs):
        rs[a] += d[k, :W].double(); rv[b] += d[k, W:].double()
    assert rel_err(dseq.view(Bt, W, 768), rs) < 1e-6 and rel_err(dvis.view(Bv, F, 768), rv) < 1e-6
    pos, typ = gen(32, 768, seed=4), gen(2, 768, seed=5)
    tab = torch.zeros(S, 768, device=DEV)
    ops.postype_fwd(pos.to(DEV), typ.to(DEV), W, S, tab)
    reft = pos[:S] + typ[(torch.arange(S) >= W).long()]
    assert torch.equal(tab.cpu(), reft)
    dpos, dtyp = torch.zeros(32, 768, device=DEV), torch.zeros(2, 768, device=DEV)
    dt_ = gen(S, 768, seed=6)
    ops.postype_bwd(dt_.to(DEV), W, S, dpos, dtyp)
    assert rel_err(dpos[:S], dt_) < 1e-6 and float(dpos[S:].abs().max()) == 0.0
    assert rel_err(dtyp, torch.stack([dt_[:W].sum(0), dt_[W:].sum(0)])) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
def test_tanh_gelu_simdense_colsum(dtype):
    x = gen(37, 768, seed=1)
    y = torch.empty(37, 768, device=DEV)
    ops.tanh_fwd(x.to(DEV), y)
    assert rel_err(y, torch.tanh(x.double())) < 1e-6
    dy = gen(37, 768, seed=2)
    dx = torch.empty(37, 768, device=DEV, dtype=dtype)
    ops.tanh_bwd(dy.to(DEV), y, dx)
    assert rel_err(dx.float(), dy.double() * (1 - torch.tanh(x.double()) ** 2)) < tol(dtype)
    u = gen(37, 768, seed=3).to(DEV, dtype)
    du = torch.empty(37, 768, device=DEV, dtype=dtype)
    ops.gelu_bwd(dy.to(DEV), u, du)
    uu = u.double().cpu().requires_grad_(True)
    O.gelu(uu).backward(dy.double())
    assert rel_err(du.float(), uu.grad) < tol(dtype)
    w, b = gen(768, seed=4, scale=0.05), gen(1, seed=5)
    s = torch.empty(37, device=DEV)
    ops.simdense_fwd(x.to(DEV), w.to(DEV), b.to(DEV), s)
    assert rel_err(s, x.double() @ w.double() + b.double()) < 1e-5
    ds = gen(37, seed=6)
    dxs = torch.empty(37, 768, device=DEV); dw = torch.zeros(768, device=DEV); db = torch.zeros(1, device=DEV)
    ops.simdense_bwd(ds.to(DEV), x.to(DEV), w.to(DEV), dxs, dw, db)
    assert rel_err(dxs, ds.double()[:, None] * w.double()[None]) < 1e-6
    assert rel_err(dw, ds.double() @ x.double()) < 1e-5 and abs(float(db) - float(ds.sum())) < 1e-4
    m = gen(100, 1024, seed=7).to(DEV, dtype)
    cs = torch.ones(1024, device=DEV)
    ops.colsum(m, cs)
    assert rel_err(cs, m.double().cpu().sum(0) + 1) < 1e-5
    big = gen(777, 1040, seed=9).to(DEV, dtype)            # an aligned strided view, rows not a multiple of anything, and a misaligned view
    for lo, hi in ((8, 1008), (3, 1003)):
        view = big[:, lo:hi]
        cs = torch.full((hi - lo,), 2.0, device=DEV)
        ops.colsum(view, cs)
        assert rel_err(cs, view.double().cpu().sum(0) + 2) < 1e-5, (lo, hi)
    z = gen(50, 64, seed=8).to(DEV, dtype)
    ref = z.double().cpu() * 0.25
    ops.scale_ct(z, torch.tensor([0.25], device=DEV))
    assert rel_err(z.float(), ref) < tol(dtype)


@pytest.mark.parametrize("V", [1000, 1002])             # 30522 % 4 == 2: the vectorised row passes end in a scalar tail
@pytest.mark.parametrize("dtype", DTYPES)
def test_ce_loss_with_ignore_index(dtype, V):
    """CrossEntropyLoss(ignore_index=-1) (modeling.py:168): padding label 0 is NOT ignored, only -1 is."""
    T = 50
    ld = 1008
    logits = torch.zeros(T, ld); logits[:, :V] = gen(T, V, seed=1, scale=3.0)
    g = torch.Generator().manual_seed(2)
    labels = torch.randint(0, V, (T,), generator=g)
    labels[::5] = -1
    labels[1] = 0
    labels[2] = V - 1                                     # the label in the scalar tail
    ld_ = torch.zeros(T, ld, device=DEV, dtype=dtype)
    loss = torch.zeros(1, device=DEV); scr = torch.zeros(2, device=DEV)
    ops.ce_loss(logits.to(DEV), labels.to(DEV), V, scr, loss, ld_)
    lr = logits[:, :V].double().requires_grad_(True)
    ref = O.cross_entropy_ignore(lr, labels)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * float(ref)
    assert rel_err(ld_[:, :V].float(), lr.grad) < tol(dtype)
    assert float(ld_[:, V:].abs().max()) == 0.0 and float(ld_[::5].abs().max()) == 0.0
    labels[:] = -1                                        # nothing to average over: NaN, like torch
    ops.ce_loss(logits.to(DEV), labels.to(DEV), V, scr, loss, ld_)
    assert math.isnan(float(loss))


@pytest.mark.parametrize("rows,V", [(50, 1000), (192, 1002), (288, 30522)])       # 30522 = 238 full column tiles + 58 columns
@pytest.mark.parametrize("dtype", DTYPES)
def test_vocab_ce_online_log_softmax_matches_the_materialised_path(dtype, rows, V):
    """K16 (univl_vocab_ce_fwd / _bwd): h . E^T + bias -> CrossEntropyLoss(ignore_index=-1) (module_bert.py:327-330 + modeling.py:253)
    with the statistics taken in the product's epilogue.  Against fp64 on the SAME (rounded) operands: loss, n_valid, every row's
    log-sum-exp, dlogits = (softmax - onehot) * gout / n_valid, zero rows where the label is ignored, untouched padding columns, NaN when
    no row counts; and against the two-step path (univl_gemm -> univl_ce_loss -> scale) of the same library."""
    K = 768
    ldv = (V + 7) // 8 * 8
    x = gen(rows, K, seed=1).to(DEV, dtype)
    table = (gen(V, K, seed=2, scale=0.06)).to(DEV, dtype)
    bias = gen(V, seed=3, scale=0.5).to(DEV)
    g = torch.Generator().manual_seed(4)
    labels = torch.randint(0, V, (rows,), generator=g)
    labels[::5] = -1
    labels[1], labels[2] = 0, V - 1
    labels_d = labels.to(DEV)
    dl = torch.full((rows, ldv), 7.0, device=DEV, dtype=dtype)
    d, buf = ops.vocab_ce_desc(x, table, bias, labels_d, dl, V)
    gout = torch.tensor([0.37], device=DEV)
    ops.vocab_ce_fwd(d)
    d.gout = gout.data_ptr()
    ops.vocab_ce_bwd(d)
    lr = (x.double().cpu() @ table.double().cpu().T + bias.double().cpu()).requires_grad_(True)
    ref = O.cross_entropy_ignore(lr, labels)
    (ref * 0.37).backward()
    ot = 2e-6 if dtype == torch.float32 else 2e-6             # same operands, fp32 accumulation on both sides
    assert abs(float(buf["loss"]) - float(ref)) < max(ot, 3e-6) * max(1.0, abs(float(ref))), (float(buf["loss"]), float(ref))
    assert float(buf["scratch"][0]) == float((labels != -1).sum())
    assert rel_err(buf["lse"], torch.logsumexp(lr.detach(), 1)) < 1e-6
    assert rel_err(dl[:, :V].float(), lr.grad) < tol(dtype)
    assert float(dl[::5, :V].float().abs().max()) == 0.0
    if ldv > V:
        assert float((dl[:, V:].float() - 7.0).abs().max()) == 0.0
    # the two-step path of the same library on the same operands
    logits = torch.zeros(rows, ldv, device=DEV)
    ops.gemm(x, table, rows, V, K, out32=logits, bias=bias)
    dl2 = torch.zeros(rows, ldv, device=DEV, dtype=dtype)
    loss2, scr2 = torch.zeros(1, device=DEV), torch.zeros(2, device=DEV)
    ops.ce_loss(logits, labels_d, V, scr2, loss2, dl2)
    ops.scale_ct(dl2, gout)
    assert abs(float(buf["loss"]) - float(loss2)) < 3e-6 * max(1.0, abs(float(loss2)))
    # (bf16: the two-step path rounds twice -- after the 1 / n_valid of the loss kernel and after the scale pass -- K16 once)
    assert rel_err(dl[:, :V].float(), dl2[:, :V].float().double().cpu()) < (2e-5 if dtype == torch.float32 else 1e-2)
    # bit-reproducible: fixed-order folds everywhere
    first = (buf["loss"].clone(), buf["lse"].clone(), dl.clone())
    d.gout = None
    ops.vocab_ce_fwd(d)
    d.gout = gout.data_ptr()
    ops.vocab_ce_bwd(d)
    assert torch.equal(buf["loss"], first[0]) and torch.equal(buf["lse"], first[1]) and torch.equal(dl, first[2])
    labels_d.fill_(-1)                                    # nothing to average over: NaN, like torch; dlogits all zero
    ops.vocab_ce_fwd(d)
    ops.vocab_ce_bwd(d)
    assert math.isnan(float(buf["loss"])) and float(dl[:, :V].float().abs().max()) == 0.0


def test_mfm_nce_loss():
    """_calculate_mfm_loss tail (modeling.py:285-297): pair mask * -1e8, diagonal log-softmax, masked-position mean."""
    B, F = 3, 8
    n = B * F
    logits = gen(n, n, seed=1, scale=2.0)
    vmask = (torch.arange(F)[None] < torch.tensor([[8], [3], [5]])).long().reshape(-1)
    lab = torch.full((n,), -1, dtype=torch.int64)
    lab[[1, 4, 9, 17, 20]] = torch.tensor([1, 4, 1, 1, 4])
    x = logits.double().requires_grad_(True)
    vm = vmask.double()
    masked = x + (1.0 - vm.view(-1, 1) @ vm.view(1, -1)) * -1e8
    nce = -torch.diag(torch.log_softmax(masked, dim=-1))
    ref = nce[lab != -1].mean()
    ref.backward()
    buf = torch.zeros(n, n, device=DEV); buf.copy_(logits)
    loss = torch.zeros(1, device=DEV); scr = torch.zeros(2, device=DEV)
    ops.mfm_nce_loss(buf, vmask.to(DEV), lab.to(DEV), scr, loss, buf)          # gradient written in place
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
    assert rel_err(buf, x.grad) < 1e-5


# ------------------------------------------------------------------------------- decoding / evaluation helpers
def test_gather_rows_and_log_softmax_rows():
    R, T, W2 = 7, 6, 1536
    src = torch.randn(R, T, W2, generator=torch.Generator().manual_seed(1)).to(DEV, torch.bfloat16)
    dst = torch.zeros_like(src)
    idx = torch.tensor([3, 3, 0, 6, 1, 2, 5], dtype=torch.int32, device=DEV)
    ops.gather_rows(src, dst, idx, R, T * W2 * 2, 4 * W2 * 2)          # positions [0, 4) of the parent rows
    assert torch.equal(dst[:, :4], src[idx.long(), :4]) and float(dst[:, 4:].abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        ops.gather_rows(src, dst, idx, R, T * W2 * 2, 4 * W2 * 2 + 8)  # not a multiple of 16 bytes
    x = gen(5, 30528, seed=2, scale=3.0).to(DEV)
    ref = torch.log_softmax(x[:, :30522].double().cpu(), dim=-1)
    tail = x[:, 30522:].clone()
    ops.log_softmax_rows(x, 30522)
    assert float((x[:, :30522].double().cpu() - ref).abs().max()) < 1e-5
    assert torch.equal(x[:, 30522:], tail)                              # padding columns untouched


def test_rank_counts():
    x = gen(33, 33, seed=4).to(DEV)
    x[5, 7] = x[5, 5]
    gt, eq = ops.rank_counts(x)
    d = x.diag()[:, None]
    assert torch.equal(gt.long(), (x > d).sum(1)) and torch.equal(eq.long(), (x == d).sum(1))


def test_attention_kv_cache_strides():
    """Sq = 1 queries over a key/value cache of capacity Tmax > Sk (batch strides) == dense attention on the prefix."""
    dt = ops.dtype_code(torch.float32)
    B, NH, Tmax, t = 3, 12, 10, 6
    cache = gen(B, Tmax, 2 * 768, seed=5).to(DEV)
    q = gen(B, 768, seed=6).to(DEV)
    out = torch.zeros(B, 768, device=DEV); lse = torch.zeros(B * NH, device=DEV)
    ops.attention_fwd(dt, B, NH, 1, t, q, 768, (cache, 0), 2 * 768, (cache, 768), 2 * 768, out, 768, lse,
                      bsk=Tmax * 2 * 768, bsv=Tmax * 2 * 768)
    k = cache[:, :t, :768].reshape(B, t, NH, 64).permute(0, 2, 1, 3).double().cpu()
    v = cache[:, :t, 768:].reshape(B, t, NH, 64).permute(0, 2, 1, 3).double().cpu()
    qq = q.reshape(B, 1, NH, 64).permute(0, 2, 1, 3).double().cpu()
    ref = (torch.softmax(qq @ k.transpose(-1, -2) / 8.0, -1) @ v).permute(0, 2, 1, 3).reshape(B, 768)
    assert rel_err(out, ref) < 1e-5


# ------------------------------------------------------------------- sparse word-table bookkeeping / merged clears
def test_zero_many_and_word_row_bookkeeping():
    """univl_zero_many (several buffers, one launch) and the listed-rows clear / append / sum-of-squares of the word table
    (univl_rows_zero / _append / _sumsq), incl. duplicates, list reset, accumulation and overflow."""
    g = torch.Generator().manual_seed(3)
    bufs = [torch.randn(n, generator=g).to(DEV) for n in (4, 1000, 64 * 768, 123456)]
    bufs.append(torch.randn(16, generator=g).to(DEV).to(torch.bfloat16)[:8])                  # 16 bytes
    guard = torch.randn(1000, generator=g).to(DEV)
    keep = guard.clone()
    ops.zero_many(bufs)
    assert all(float(b.float().abs().max()) == 0.0 for b in bufs) and torch.equal(guard, keep)
    V, N, cap = 3000, 768, 64
    table = torch.randn(V, N, generator=g).to(DEV)
    ref = table.clone()
    lst = torch.zeros(cap, dtype=torch.int64, device=DEV)
    meta = torch.zeros(2, dtype=torch.int32, device=DEV)
    ids1 = torch.tensor([5, 17, 5, 2999, 0, 17], device=DEV)
    ops.rows_append(ids1, lst, meta, reset=True)
    assert meta.tolist() == [6, 0] and lst[:6].tolist() == ids1.tolist()
    out = torch.zeros(1, device=DEV)
    ops.rows_sumsq(table, lst, meta, out)                    # each listed row once
    uniq = torch.tensor([5, 17, 2999, 0])
    assert abs(float(out) - float((ref[uniq].double() ** 2).sum())) < 1e-3 * float(out)
    ids2 = torch.tensor([7, 5], device=DEV)
    ops.rows_append(ids2, lst, meta, reset=False)            # gradient accumulation: the list grows
    assert meta.tolist() == [8, 0]
    ops.rows_zero(table, lst, meta)
    z = torch.tensor([5, 17, 2999, 0, 7])
    assert float(table[z].abs().max()) == 0.0
    mask = torch.ones(V, dtype=torch.bool); mask[z] = False
    assert torch.equal(table[mask].cpu(), ref[mask].cpu())   # nothing else touched
    ops.rows_append(ids2, lst, meta, reset=True)
    assert meta.tolist() == [2, 0]
    big = torch.arange(cap, device=DEV)
    ops.rows_append(big, lst, meta, reset=False)             # does not fit: every row counts as listed from now on
    assert meta.tolist()[1] == 1
    table.copy_(ref)
    out.zero_()
    ops.rows_sumsq(table, lst, meta, out)
    assert abs(float(out) - float((ref.double() ** 2).sum())) < 1e-3 * float(out)
    ops.rows_zero(table, lst, meta)
    assert float(table.abs().max()) == 0.0


def test_copy_many_bit_exact():
    """univl_copy_many: several device-to-device copies in one launch (input staging / loss hand-off of the training step) --
    bit-exact for aligned buffers (16-byte words), misaligned views and sizes that are not a multiple of 16, with the bytes
    around every destination untouched."""
    g = torch.Generator().manual_seed(3)
    sizes = [4, 1536, 1537 * 8, 48 * 1024 * 8 * 4, 24, 16, 1, 3 * 1024 * 1024 + 5]
    pairs, checks = [], []
    for k, nb in enumerate(sizes):
        src_base = torch.randint(0, 256, (nb + 64,), generator=g, dtype=torch.uint8).to(DEV)
        dst_base = torch.full((nb + 64,), 0xEE, dtype=torch.uint8, device=DEV)
        so, do = ((0, 0), (8, 0), (0, 4), (16, 16), (3, 5), (0, 0), (1, 2), (32, 32))[k]
        src, dst = src_base[so:so + nb], dst_base[do:do + nb]
        pairs.append((dst, src))
        checks.append((dst_base, do, nb, src.clone()))
    ops.copy_many(pairs)
    torch.cuda.synchronize()
    for dst_base, do, nb, want in checks:
        assert torch.equal(dst_base[do:do + nb], want)
        assert bool((dst_base[:do] == 0xEE).all()) and bool((dst_base[do + nb:] == 0xEE).all())
    # typed tensors, as UniVL.forward stages them
    ids = torch.randint(0, 30522, (4, 48), generator=g).to(DEV)
    video = torch.randn(4 * 48, 1024, generator=g, dtype=torch.float64).to(DEV)
    loss = torch.randn(1, generator=g).to(DEV)
    d_ids, d_video, d_loss = torch.zeros_like(ids), torch.zeros_like(video), torch.empty((), device=DEV)
    ops.copy_many([(d_ids, ids), (d_video, video), (d_loss, loss)])
    assert torch.equal(d_ids, ids) and torch.equal(d_video, video) and float(d_loss) == float(loss)


@pytest.mark.parametrize("T,K_out,K_in,ks", [(192, 768, 768, 2), (192, 3072, 768, 8), (192, 768, 3072, 1), (192, 2304, 768, 6),
                                            (320, 768, 768, 1), (100, 768, 3072, 1),
                                            # from 384 tokens on: the rectangular form (64 x 128 dgrad tiles, 128 x 64 weight-gradient tiles)
                                            (384, 768, 768, 1), (768, 3072, 768, 3), (768, 768, 3072, 1), (768, 2304, 768, 3),
                                            (400, 768, 3072, 1), (1000, 1024, 512, 1)])
def test_gemm_pair_equals_separate_launches(T, K_out, K_in, ks):
    """univl_gemm_pair (UNIVL_WGRAD_RIDE, default on): the dgrad product dX = dY . W and the weight-gradient product
    dW = dY^T . X of one nn.Linear backward in ONE launch give what the two separate launches give -- bit for bit for the
    weight gradient and the non-split dgrad; split-K dgrads, the fused gradient norm and the bias gradient (column-sum workgroups
    of their own in the pair launch, an LDS walk inside the product otherwise: two fixed but different summation orders) within fp32
    summation-order noise -- with the dgrad epilogues of the step (GELU' / residual / split-K) and beta = 1 accumulation."""
    bf = torch.bfloat16
    dY = gen(T, K_out, seed=1).to(DEV, bf)                     # upstream gradient [tokens, out features]
    W = gen(K_out, K_in, seed=2, scale=0.05).to(DEV, bf)       # nn.Linear weight [out, in]
    X = gen(T, K_in, seed=3).to(DEV, bf)                       # saved input activation
    gelu = K_in == 3072 and ks == 1                            # FFN2-dgrad shape: GELU' epilogue, bf16 output
    u = gen(T, K_in, seed=4).to(DEV, bf) if gelu else None
    res = None if gelu else gen(T, K_in, seed=5).to(DEV)

    def run(pair):
        dX32 = None if gelu else torch.zeros(T, K_in, device=DEV)
        dX16 = torch.zeros(T, K_in, device=DEV, dtype=bf) if gelu else None
        dW = gen(K_out, K_in, seed=6).to(DEV)                  # accumulate into an existing gradient (beta = 1)
        db = torch.zeros(K_out, device=DEV)
        part = torch.zeros(K_out * K_in // 1024, device=DEV)
        dg = ops.gemm_desc(dY, W, T, K_in, K_out, trans_b=True, out32=dX32, out16=dX16, aux=u, gelu="bwd" if gelu else None,
                           residual=res, ksplit=ks)
        wg = ops.gemm_desc(dY, X, K_out, K_in, T, trans_a=True, trans_b=True, out32=dW, accumulate=True, dbias=db,
                           sumsq=part, sumsq_rows=0, sumsq_stride=part.numel())
        if pair:
            assert ops.gemm_pair(dg, wg), "the C side refused a bf16 64-tile pair"
        else:
            _lib.check(_lib.lib().univl_gemm(ops._BYREF(dg), ops._stream()), "gemm")
            _lib.check(_lib.lib().univl_gemm(ops._BYREF(wg), ops._stream()), "gemm")
        torch.cuda.synchronize()
        return (dX16 if gelu else dX32).float().cpu(), dW.cpu(), db.cpu(), float(part.double().sum())

    x1, w1, b1, s1 = run(True)
    x0, w0, b0, s0 = run(False)
    ref_w = gen(K_out, K_in, seed=6).double() + dY.double().cpu().T @ X.double().cpu()
    assert rel_err(w0, ref_w) < 1e-2 and rel_err(w1, ref_w) < 1e-2
    assert torch.equal(w1, w0), float((w1 - w0).abs().max())
    assert rel_err(b1, b0) < 1e-5 and rel_err(b1, dY.double().cpu().sum(0)) < 1e-5
    if ks == 1:
        assert torch.equal(x1, x0)
    else:
        assert rel_err(x1, x0) < 1e-5
    assert abs(s1 - s0) <= 1e-5 * abs(s0) and s0 > 0
