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
VARIANTS = [(64, 2, 4), (64, 2, 8), (128, 2, 4), (128, 2, 8), (12864, 2, 8), (12864, 2, 4), (64128, 2, 8), (64128, 2, 4)]   # tile, stages, waves (12864 = 128 x 64, 64128 = 64 x 128: K-major A only)
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
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad")
    ap.add_argument("--group-dbias", type=int, default=1, help="0: the grouped weight gradients without the fused bias gradients")
    ap.add_argument("--group-rows", type=int, default=0, help="also time one layer's four weight gradients as a grouped launch over this many tokens")
    ap.add_argument("--variants", default="", help="e.g. 128/2/8,12864/2/8 (default: all)")
    ap.add_argument("--check", type=int, default=1, help="compare every variant's result with torch.matmul once")
    a = ap.parse_args()
    global VARIANTS
    if a.variants:
        VARIANTS = [tuple(int(x) for x in v.split("/")) for v in a.variants.split(",")]
    bf = torch.bfloat16
    results = []
    for M in [int(x) for x in a.rows.split(",") if x]:
        variants = list(VARIANTS)
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
            for kind in a.kinds.split(","):
                row = dict(rows=M, linear=name, kind=kind, N=N, K=K, us={}, err={})
                for tile, stages, waves in variants:
                    if kind == "wgrad" and tile > 256:
                        continue
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
                    key = "%d/%d/%d" % (tile, stages, waves)
                    row["us"][key] = round(us, 2)
                    if a.check:                                    # the last launch of the replay was fn(REPS - 1)
                        j = REPS - 1
                        if kind == "fwd":
                            got, ref = Y16.float(), X[j % NA].float() @ W[j % NW].float().t()
                        elif kind == "dgrad":
                            got, ref = dX, dY[j % NA].float() @ W[j % NW].float()
                        else:
                            got, ref = dW, dY[j % NA].float().t() @ X[j % NA].float()
                        row["err"][key] = float((got - ref).norm() / ref.norm())
                        if row["err"][key] > (6e-3 if kind == "fwd" else 1e-5 * 50):
                            print("   !! %s %s %s: relative error %.3e" % (name, kind, key, row["err"][key]))
                best = min(row["us"], key=row["us"].get)
                row["best"] = best
                row["tflops_best"] = round(flops / row["us"][best] * 1e-6, 1)
                results.append(row)
                print("  %-8s %-5s N=%4d K=%4d  " % (name, kind, N, K) +
                      "  ".join("%s %7.1f|%5.0f" % (k, v, flops / v * 1e-6) for k, v in row["us"].items()) + "   best " + best)
            del X, W, dY
    if a.group_rows:
        # a layer's four weight-gradient products as ONE grouped launch (what the backward plan issues), contraction over T tokens
        T = a.group_rows
        print("grouped weight gradients of one layer, T = %d tokens, fused bias gradients: %d" % (T, a.group_dbias))
        dYs = {n_: [torch.randn(T, N, device=DEV).to(bf) for _ in range(2)] for n_, N, K in LINEAR}
        Xs = {n_: [torch.randn(T, K, device=DEV).to(bf) for _ in range(2)] for n_, N, K in LINEAR}
        dWs = {n_: torch.zeros(N, K, device=DEV) for n_, N, K in LINEAR}
        dbs = {n_: torch.zeros(N, device=DEV) for n_, N, K in LINEAR}
        flops = sum(2.0 * T * N * K for _, N, K in LINEAR)
        row = dict(rows=T, linear="layer", kind="wgrad_group", us={}, err={})
        for tile, stages, waves in [(0, 0, 0)] + [v for v in VARIANTS if v[0] <= 128]:
            def fn(i, tile=tile, stages=stages, waves=waves):
                ops.gemm_group([ops.gemm_desc(dYs[n_][i % 2], Xs[n_][i % 2], N, K, T, trans_a=True, trans_b=True, out32=dWs[n_],
                                              dbias=dbs[n_] if (a.group_dbias and n_ in ("qkv", "ffn1")) else None, tile=tile, stages=stages, waves=waves)
                                for n_, N, K in LINEAR])
            try:
                us = time_graph(fn)
            except RuntimeError as e:
                print("   group tile %d stages %d waves %d: %s" % (tile, stages, waves, str(e)[:80]))
                continue
            key = "%d/%d/%d" % (tile, stages, waves)
            row["us"][key] = round(us, 2)
            j = (REPS - 1) % 2
            row["err"][key] = max(float((dWs[n_] - dYs[n_][j].float().t() @ Xs[n_][j].float()).norm() /
                                        (dYs[n_][j].float().t() @ Xs[n_][j].float()).norm()) for n_, N, K in LINEAR)
        results.append(row)
        print("  " + "  ".join("%s %7.1f|%5.0f (err %.1e)" % (k, v, flops / v * 1e-6, row["err"][k]) for k, v in row["us"].items()))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
