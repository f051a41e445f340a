"""VERDICT r5 next 2 asked for a one-layer prototype of a persistent / merged launch with a kill criterion (>= 10 % over the separate
launches, alternating on one box).  This measures it for the FFN half of an encoder layer's forward:

    three launches (the plans):  univl_gemm_ln(attention-output product + LayerNorm)  ->  univl_gemm(FFN1 + GELU)  ->  univl_gemm_ln(FFN2 + LayerNorm)
    one launch (prototype):      univl_proto_layer_ffn: the same tiles as roles of ONE grid, later products waiting on row-block flags
                                 (csrc/gemm.hip, #ifdef UNIVL_PROTO; results NOT valid: no write-through hand-off -- a lower bound on time)

as a chain of 12 dependent layers (each layer's input = the previous layer's output, its own weights: HBM-cold as in the step), captured
into one hipGraph and replayed; no optimizer chunks riding (the bare chain).  Needs the prototype build:

    python univl_amd/build.py --variant proto -DUNIVL_PROTO
    UNIVL_LIB=univl_amd/lib/libunivl_hip_proto.so python scripts/mb_layer_proto.py [rows ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univl_amd import ops, _lib  # noqa: E402

dev, bf = "cuda", torch.bfloat16
H, I, L = 768, 3072, 12


def build(M, ks_o, ks_f2):
    nb = (M + 63) // 64
    layers = []
    x16 = torch.randn(M, H, device=dev).to(bf)
    x32 = torch.randn(M, H, device=dev)
    zero_me = []
    for l in range(L):
        wo = (torch.randn(H, H, device=dev) * H ** -0.5).to(bf)
        w1 = (torch.randn(I, H, device=dev) * H ** -0.5).to(bf)
        w2 = (torch.randn(H, I, device=dev) * I ** -0.5).to(bf)
        bo, b1, b2 = torch.zeros(H, device=dev), torch.zeros(I, device=dev), torch.zeros(H, device=dev)
        g1, be1, g2, be2 = torch.ones(H, device=dev), torch.zeros(H, device=dev), torch.ones(H, device=dev), torch.zeros(H, device=dev)
        y1, y2 = torch.zeros(M, H, device=dev), torch.zeros(M, H, device=dev)
        st1, st2 = torch.zeros(M, 2, device=dev), torch.zeros(M, 2, device=dev)
        a32, a16 = torch.zeros(M, H, device=dev), torch.zeros(M, H, device=dev, dtype=bf)
        o32, o16 = torch.zeros(M, H, device=dev), torch.zeros(M, H, device=dev, dtype=bf)
        u, f = torch.zeros(M, I, device=dev, dtype=bf), torch.zeros(M, I, device=dev, dtype=bf)
        ctr1, ctr2 = torch.zeros(2 * nb, dtype=torch.int32, device=dev), torch.zeros(2 * nb, dtype=torch.int32, device=dev)
        flags = torch.zeros(2 * nb, dtype=torch.int32, device=dev)
        od = ops.gemm_desc(x16, wo, M, H, H, out32=y1, bias=bo, ksplit=ks_o)
        l1 = ops.layernorm_desc(_lib.DT_BF16, M, H, x=y1, residual=x32, gamma=g1, beta=be1, y=y1, stats=st1, out32=a32, out16=a16)
        f1 = ops.gemm_desc(a16, w1, M, I, H, out16=f, bias=b1, aux=u, gelu="fwd")
        f2 = ops.gemm_desc(f, w2, M, H, I, out32=y2, bias=b2, ksplit=ks_f2)
        l2 = ops.layernorm_desc(_lib.DT_BF16, M, H, x=y2, residual=a32, gamma=g2, beta=be2, y=y2, stats=st2, out32=o32, out16=o16)
        layers.append(dict(od=od, l1=l1, f1=f1, f2=f2, l2=l2, ctr1=ctr1, ctr2=ctr2, flags=flags,
                           keep=[wo, w1, w2, bo, b1, b2, g1, be1, g2, be2, y1, y2, st1, st2, a32, a16, o32, o16, u, f, x16, x32]))
        zero_me += [y1, y2, flags]
        x16, x32 = o16, o32
    return layers, zero_me


def timeit(run, zero_me, reps=5):
    def step():
        for t in zero_me:
            t.zero_()
        run()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    gz = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gz):
        for t in zero_me:
            t.zero_()
    for gg in (g, gz):
        gg.replay()
    torch.cuda.synchronize()
    out = []
    for gg in (g, gz):
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                gg.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 10)
        out.append(best)
    return (out[0] - out[1]) / L          # us per layer, the clears subtracted


def main():
    Lb = _lib.lib()
    assert hasattr(Lb, "univl_proto_layer_ffn"), "load the prototype build: UNIVL_LIB=univl_amd/lib/libunivl_hip_proto.so"
    Lb.univl_proto_layer_ffn.argtypes = [C.c_void_p] * 9
    rows = [int(a) for a in sys.argv[1:]] or [192, 96, 48]
    for M in rows:
        tiles = ((M + 63) // 64) * 12
        ks = max(1, min(int(180 / tiles + 0.5), 3))          # the plans' split policy (K = 768: at most 3 slices of 256)
        ks2 = max(1, min(int(180 / tiles + 0.5), 12))
        layers, zero_me = build(M, ks, ks2)
        h = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def three():
            for d in layers:
                assert ops.gemm_ln(d["od"], d["l1"], d["ctr1"])
                _lib.check(Lb.univl_gemm(C.byref(d["f1"]), h()), "ffn1")
                assert ops.gemm_ln(d["f2"], d["l2"], d["ctr2"])

        def one():
            for d in layers:
                _lib.check(Lb.univl_proto_layer_ffn(C.byref(d["od"]), C.byref(d["l1"]), C.byref(d["f1"]), C.byref(d["f2"]), C.byref(d["l2"]),
                                                    C.c_void_p(d["ctr1"].data_ptr()), C.c_void_p(d["ctr2"].data_ptr()), C.c_void_p(d["flags"].data_ptr()), h()),
                           "proto")
        res = []
        for r in range(3):
            res.append((timeit(three, zero_me), timeit(one, zero_me)))
        t3 = sorted(x[0] for x in res)[1]
        t1 = sorted(x[1] for x in res)[1]
        print("rows %4d (split %d / %d)   three launches %6.2f us per layer   one launch %6.2f us   %+5.1f %%   [runs: %s]" %
              (M, ks, ks2, t3, t1, 100 * (t1 / t3 - 1), " ".join("%.1f/%.1f" % x for x in res)), flush=True)


if __name__ == "__main__":
    main()
