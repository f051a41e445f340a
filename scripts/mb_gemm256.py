"""Micro-benchmark of the 256 x 256 "8-phase" product body (univl_amd/csrc/gemm256.h) against the older tiles, shape by shape, for the
dense contractions of one encoder layer at 32 / 64 / 128 pairs per GPU (1536 / 3072 / 6144 tokens): forward, dgrad, the single
weight-gradient products and a layer's grouped weight gradients.  Decides `G256_MIN_ROWS` / `G256_MIN_WG` in gemm.hip and the
split-K depth of the N = 768 products in engine.EncoderStack.

    python scripts/mb_gemm256.py [--rows 1536,3072,6144] [--out gpurun_out/mb_gemm256.json]

Timing as in scripts/mb_gemm_variants.py: 24 launches in one hipGraph, operands rotating so that consecutive launches miss L2."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from univl_amd import ops  # noqa: E402
from mb_gemm_variants import time_graph, LINEAR, NA, NW, REPS  # noqa: E402

DEV = "cuda"
OLD = [(128, 2, 8), (64128, 2, 8), (12864, 2, 8), (64, 2, 8)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="1536,3072,6144")
    ap.add_argument("--out", default="gpurun_out/mb_gemm256.json")
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad,group")
    a = ap.parse_args()
    bf = torch.bfloat16
    results = []
    for M in [int(x) for x in a.rows.split(",") if x]:
        print("tokens = %d    us per launch | TFLOP/s" % M)
        for name, N, K in LINEAR:
            X = [torch.randn(M, K, device=DEV).to(bf) for _ in range(NA)]
            W = [(torch.randn(N, K, device=DEV) * 0.05).to(bf) for _ in range(NW)]
            dY = [torch.randn(M, N, device=DEV).to(bf) for _ in range(NA)]
            Y16 = torch.zeros(M, N, device=DEV, dtype=bf)
            Y32 = torch.zeros(M, N, device=DEV)
            U16 = torch.zeros(M, N, device=DEV, dtype=bf)
            dX = torch.zeros(M, K, device=DEV)
            dX16 = torch.zeros(M, K, device=DEV, dtype=bf)
            dW = torch.zeros(N, K, device=DEV)
            bias = torch.zeros(N, device=DEV)
            flops = 2.0 * M * N * K
            for kind in [k for k in a.kinds.split(",") if k != "group"]:
                row = dict(rows=M, linear=name, kind=kind, N=N, K=K, us={})
                cases = [("256/ks1", dict(tile=256))]
                wide_out = (N if kind == "fwd" else K) > 768          # the products with H-wide outputs can be split (fp32 out)
                if kind != "wgrad" and not wide_out:
                    cases += [("256/ks2", dict(tile=256, ksplit=2)), ("256/ks3", dict(tile=256, ksplit=3))]
                cases += [("%d/%d/%d" % v, dict(tile=v[0], stages=v[1], waves=v[2])) for v in OLD if not (kind == "wgrad" and v[0] > 256)]
                for key, kw in cases:
                    if kind == "fwd":
                        # what the plans run: QKV / FFN1 -> bf16 (+ GELU for FFN1), attention output / FFN2 -> fp32
                        if name == "ffn1":
                            fn = lambda i, kw=kw: ops.gemm(X[i % NA], W[i % NW], M, N, K, out16=Y16, bias=bias, aux=U16, gelu="fwd", **kw)
                        elif name == "qkv":
                            fn = lambda i, kw=kw: ops.gemm(X[i % NA], W[i % NW], M, N, K, out16=Y16, bias=bias, **kw)
                        else:
                            fn = lambda i, kw=kw: ops.gemm(X[i % NA], W[i % NW], M, N, K, out32=Y32, bias=bias, **kw)
                    elif kind == "dgrad":
                        if name == "ffn2":       # dY [T,768] . W2 [768,3072] -> du bf16 with GELU'
                            fn = lambda i, kw=kw: ops.gemm(dY[i % NA], W[i % NW], M, K, N, trans_b=True, out16=dX16, aux=X[i % NA], gelu="bwd", **kw)
                        elif name == "attn_out":
                            fn = lambda i, kw=kw: ops.gemm(dY[i % NA], W[i % NW], M, K, N, trans_b=True, out16=dX16, **kw)
                        else:
                            fn = lambda i, kw=kw: ops.gemm(dY[i % NA], W[i % NW], M, K, N, trans_b=True, out32=dX, **kw)
                    else:
                        fn = lambda i, kw=kw: ops.gemm(dY[i % NA], X[i % NA], N, K, M, trans_a=True, trans_b=True, out32=dW, **kw)
                    try:
                        us = time_graph(fn)
                    except RuntimeError as e:
                        print("   %s %s %s: %s" % (name, kind, key, str(e)[:90]))
                        continue
                    row["us"][key] = round(us, 2)
                old_best = min((v for k, v in row["us"].items() if not k.startswith("256")), default=None)
                new_best = min((v for k, v in row["us"].items() if k.startswith("256")), default=None)
                row["old_best"], row["new_best"] = old_best, new_best
                results.append(row)
                print("  %-8s %-5s out %4d contraction %4d  " % (name, kind, N if kind == "fwd" else K, K if kind == "fwd" else N) +
                      "  ".join("%s %6.1f|%4.0f" % (k, v, flops / v * 1e-6) for k, v in row["us"].items()) +
                      "   => new/old %.2f" % (new_best / old_best if old_best and new_best else float("nan")))
            del X, W, dY
        if "group" in a.kinds.split(","):
            T = M
            dYs = {n_: [torch.randn(T, N, device=DEV).to(bf) for _ in range(2)] for n_, N, K in LINEAR}
            Xs = {n_: [torch.randn(T, K, device=DEV).to(bf) for _ in range(2)] for n_, N, K in LINEAR}
            dWs = {n_: torch.zeros(N, K, device=DEV) for n_, N, K in LINEAR}
            dbs = {n_: torch.zeros(N, device=DEV) for n_, N, K in LINEAR}
            flops = sum(2.0 * T * N * K for _, N, K in LINEAR)
            row = dict(rows=T, linear="layer", kind="wgrad_group", us={})
            for key, kw in [("256", dict(tile=256)), ("128/2/4", dict(tile=128, stages=2, waves=4)), ("128/2/8", dict(tile=128, stages=2, waves=8)),
                            ("64/2/4", dict(tile=64, stages=2, waves=4))]:
                def fn(i, kw=kw):
                    ops.gemm_group([ops.gemm_desc(dYs[n_][i % 2], Xs[n_][i % 2], N, K, T, trans_a=True, trans_b=True, out32=dWs[n_],
                                                  dbias=dbs[n_] if n_ in ("qkv", "ffn1") else None, **kw) for n_, N, K in LINEAR])
                try:
                    row["us"][key] = round(time_graph(fn), 2)
                except RuntimeError as e:
                    print("   group %s: %s" % (key, str(e)[:90]))
            results.append(row)
            print("  layer's grouped weight gradients:  " + "  ".join("%s %7.1f|%4.0f" % (k, v, flops / v * 1e-6) for k, v in row["us"].items()))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
