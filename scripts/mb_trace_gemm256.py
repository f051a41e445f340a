"""Phase trace of the 256 x 256 product body (csrc/gemm256.h) -- entry, first K tile landed, end of the K loop, end of the epilogue per
workgroup (the -DUNIVL_TRACE build; see scripts/mb_trace_gemm.py for the method and the columns).

    python univl_amd/build.py --trace
    python scripts/mb_trace_gemm256.py [--rows 6144]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mb_trace_gemm import run, DEV, bf  # noqa: E402  (sets UNIVL_LIB to the trace build)

import torch  # noqa: E402

from univl_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="6144")
    ap.add_argument("--tiles", default="256,128")
    a = ap.parse_args()
    H, I = 768, 3072
    for M in [int(x) for x in a.rows.split(",")]:
        print("---- rows M = %d" % M)

        def pool(n, k):
            cnt = max(2, int(300e6 // (n * k * 2)) + 1)
            return [torch.randn(n, k, device=DEV).to(bf) * 0.02 for _ in range(cnt)]
        x = torch.randn(M, H, device=DEV).to(bf)
        f = torch.randn(M, I, device=DEV).to(bf)
        bias3, biasH, biasI = (torch.randn(n, device=DEV) for n in (3 * H, H, I))
        qkv = torch.empty(M, 3 * H, device=DEV, dtype=bf)
        y32 = torch.zeros(M, H, device=DEV)
        u = torch.empty(M, I, device=DEV, dtype=bf)
        fo = torch.empty(M, I, device=DEV, dtype=bf)
        dxd = torch.randn(M, H, device=DEV).to(bf)
        du = torch.empty(M, I, device=DEV, dtype=bf)
        dqkv = torch.randn(M, 3 * H, device=DEV).to(bf)
        dx32 = torch.zeros(M, H, device=DEV)
        gW2 = torch.empty(H, I, device=DEV)
        gWq = torch.empty(3 * H, H, device=DEV)
        touch = lambda t: t.mul_(1.0)
        Wq, W1, W2 = pool(3 * H, H), pool(I, H), pool(H, I)
        for tile in [int(t) for t in a.tiles.split(",")]:
            kw = dict(tile=tile)
            run("fwd QKV N2304 K768 bf16 [%d]" % tile, lambda i, st: (touch(x), st(), ops.gemm(x, Wq[i % len(Wq)], M, 3 * H, H, out16=qkv, bias=bias3, **kw)), rot=24)
            run("fwd FFN1 N3072 K768 gelu [%d]" % tile, lambda i, st: (touch(x), st(), ops.gemm(x, W1[i % len(W1)], M, I, H, out16=fo, bias=biasI, aux=u, gelu="fwd", **kw)), rot=24)
            run("fwd FFN2 N768 K3072 f32 [%d]" % tile, lambda i, st: (touch(f), st(), ops.gemm(f, W2[i % len(W2)], M, H, I, out32=y32, bias=biasH, **kw)), rot=24)
            run("dgrad FFN2 N3072 K768 gelu' [%d]" % tile, lambda i, st: (touch(dxd), st(), ops.gemm(dxd, W2[i % len(W2)], M, I, H, trans_b=True, out16=du, aux=u, gelu="bwd", **kw)), rot=24)
            run("dgrad QKV N768 K2304 f32 [%d]" % tile, lambda i, st: (touch(dqkv), st(), ops.gemm(dqkv, Wq[i % len(Wq)], M, H, 3 * H, trans_b=True, out32=dx32, **kw)), rot=24)
            run("wgrad FFN2 [768x3072] K=%d [%d]" % (M, tile), lambda i, st: (touch(dxd), st(), ops.gemm(dxd, f, H, I, M, trans_a=True, trans_b=True, out32=gW2, **kw)), rot=1)
            run("wgrad QKV [2304x768] K=%d [%d]" % (M, tile), lambda i, st: (touch(dqkv), st(), ops.gemm(dqkv, x, 3 * H, H, M, trans_a=True, trans_b=True, out32=gWq, **kw)), rot=1)


if __name__ == "__main__":
    main()
