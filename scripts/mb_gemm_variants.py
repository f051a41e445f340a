"""Micro-benchmark behind the GEMM tile / pipeline-depth choice (univl_amd/csrc/gemm.hip `choose`): every dense contraction
shape of one encoder layer -- forward (K-major x K-major), dgrad (K-major x T-major), wgrad (T-major x T-major) -- at the
token counts of 4 / 16 / 128 pairs per GPU (M = 192 / 768 / 6144 rows), timed for every (tile, stages, waves) variant of the bf16 kernel.

    python scripts/mb_gemm_variants.py [--rows 192,768,6144] [--out gpurun_out/mb_gemm_variants.json]

Each timing: 24 launches of the SAME problem captured in one hipGraph and replayed (no host launch cost in the number); the
weight operand rotates over 8 buffers and the activation operand over 4, so that consecutive launches do not find their
operands in L2 (in the training step every GEMM reads a different layer's weights).  Prints microseconds per launch and
TFLOP/s, and the best variant per shape.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univl_amd import ops  # noqa: E402

DEV = "cuda"
VARIANTS = [(64, 2, 4), (64, 3, 4), (64, 2, 8), (64, 3, 8), (128, 2, 4), (128, 3, 4), (128, 2, 8), (128, 3, 8), (256, 2, 8), (256, 3, 8)]   # tile, stages, waves
REPS, NW, NA = 24, 8, 4
# one encoder layer, hidden 768: (name, out columns, contraction) of the forward products; dgrad swaps them; wgrad contracts tokens
LINEAR = [("qkv", 2304, 768), ("attn_out", 768, 768), ("ffn1", 3072, 768), ("ffn2", 768, 3072)]


def time_graph(fn):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(REPS):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000.0 / REPS)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="192,768,6144")
    ap.add_argument("--out", default="gpurun_out/mb_gemm_variants.json")
    a = ap.parse_args()
    bf = torch.bfloat16
    results = []
    for M in [int(x) for x in a.rows.split(",")]:
        variants = [v for v in VARIANTS if not (M < 512 and v[0] == 256)]
        print("rows M = %d   (us per launch | TFLOP/s)   variants (tile,stages,waves): %s" % (M, variants))
        for name, N, K in LINEAR:
            X = [torch.randn(M, K, device=DEV).to(bf) for _ in range(NA)]           # layer input
            W = [(torch.randn(N, K, device=DEV) * 0.05).to(bf) for _ in range(NW)]  # nn.Linear weight [out, in]
            dY = [torch.randn(M, N, device=DEV).to(bf) for _ in range(NA)]
            Y16 = torch.zeros(M, N, device=DEV, dtype=bf)
            dX = torch.zeros(M, K, device=DEV)
            dW = torch.zeros(N, K, device=DEV)
            db = torch.zeros(N, device=DEV)
            bias = torch.zeros(N, device=DEV)
            flops = 2.0 * M * N * K
            for kind in ("fwd", "dgrad", "wgrad"):
                row = dict(rows=M, linear=name, kind=kind, N=N, K=K, us={})
                for tile, stages, waves in variants:
                    if kind == "fwd":
                        fn = lambda i: ops.gemm(X[i % NA], W[i % NW], M, N, K, out16=Y16, bias=bias, tile=tile, stages=stages, waves=waves)
                    elif kind == "dgrad":
                        fn = lambda i: ops.gemm(dY[i % NA], W[i % NW], M, K, N, trans_b=True, out32=dX, tile=tile, stages=stages, waves=waves)
                    else:
                        fn = lambda i: ops.gemm(dY[i % NA], X[i % NA], N, K, M, trans_a=True, trans_b=True, out32=dW, dbias=db,
                                                tile=tile, stages=stages, waves=waves)
                    try:
                        us = time_graph(fn)
                    except RuntimeError as e:                      # a variant the library refuses for this shape
                        print("   %s %s tile %d stages %d waves %d: %s" % (name, kind, tile, stages, waves, str(e)[:80]))
                        continue
                    row["us"]["%d/%d/%d" % (tile, stages, waves)] = round(us, 2)
                best = min(row["us"], key=row["us"].get)
                row["best"] = best
                row["tflops_best"] = round(flops / row["us"][best] * 1e-6, 1)
                results.append(row)
                print("  %-8s %-5s N=%4d K=%4d  " % (name, kind, N, K) +
                      "  ".join("%s %7.1f|%5.0f" % (k, v, flops / v * 1e-6) for k, v in row["us"].items()) + "   best " + best)
            del X, W, dY
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
