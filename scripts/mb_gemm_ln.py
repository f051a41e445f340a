"""(product + LayerNorm) as one launch (univl_gemm_ln) against the two launches, under hipGraph replay: chains of 100 dependent pairs.

    python scripts/mb_gemm_ln.py

Shapes: the attention-output product (K = 768) and FFN2 (K = 3072) at 192 / 768 tokens with the split-K factors the plans use."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univl_amd import ops, _lib  # noqa: E402

dev, bf = "cuda", torch.bfloat16


def chain(f, reps=100):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            f()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def case(M, K, ks):
    N = 768
    a = torch.randn(M, K, device=dev).to(bf)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
    bias, res = torch.randn(N, device=dev), torch.randn(M, N, device=dev)
    gm, bt = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    x, st, o32 = torch.zeros(M, N, device=dev), torch.zeros(M, 2, device=dev), torch.zeros(M, N, device=dev)
    o16 = torch.zeros(M, N, device=dev, dtype=bf)
    ctr = torch.zeros(2 * ((M + 63) // 64), dtype=torch.int32, device=dev)
    g = ops.gemm_desc(a, w, M, N, K, out32=x, bias=bias, ksplit=ks)
    ln = ops.layernorm_desc(_lib.DT_BF16, M, N, x=x, residual=res, gamma=gm, beta=bt, y=x, stats=st, out32=o32, out16=o16, p_pre=0.1, seed=1,
                            off_pre=3 << 40)
    L = _lib.lib()

    def two():
        x.zero_()
        _lib.check(L.univl_gemm(C.byref(g), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm")
        _lib.check(L.univl_layernorm_fwd(C.byref(ln), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "ln")

    def one():
        x.zero_()
        assert ops.gemm_ln(g, ln, ctr)

    def zero_only():
        x.zero_()

    t2, t1, t0 = chain(two), chain(one), chain(zero_only)
    print("M %4d K %4d ks %d   two launches %6.2f us   one launch %6.2f us   (the arena clear both include: %5.2f us)" % (M, K, ks, t2 - t0, t1 - t0, t0), flush=True)


if __name__ == "__main__":
    for M, K, ks in [(192, 768, 2), (192, 3072, 8), (768, 768, 1), (768, 3072, 3), (1024, 768, 1), (1024, 3072, 3)]:
        case(M, K, ks)
