"""Where the microseconds of ONE GEMM-family launch go at a few hundred rows -- measured inside the kernel, no profiler attached.

    python univl_amd/build.py --trace                  # lib/libunivl_hip_trace.so  (-DUNIVL_TRACE, csrc/gemm.hip)
    python scripts/mb_trace_gemm.py [--rows 192,768]

Every workgroup's thread 0 writes the device wall clock (100 MHz) at entry (t0), when its first staged tile is visible (t1), after its
K loop (t2) and after its epilogue (t3); one-thread stamp kernels in front of and behind the launch (same stream, captured into a
hipGraph with it and replayed) give the kernel boundaries.  The weight operand rotates over > 256 MB of copies so that every replay
reads it from HBM, as the training step does (every layer's weights are read once per pass); the activation operand is written by a
small kernel right in front (L2-resident, as after the producing kernel of the step).  Printed per shape, medians over the replays, us:

    in      stamp0 -> first workgroup entry          (boundary in front of the launch)
    skew    first -> last workgroup entry
    tile1   entry -> first tile landed               (median / max over workgroups)
    kloop   first tile -> end of the K loop          (median / max)
    epi     K loop -> end of the epilogue            (median / max)
    span    first entry -> last exit
    out     last exit -> stamp1                      (boundary behind the launch)
    total   stamp0 -> stamp1
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("UNIVL_LIB", os.path.join(ROOT, "univl_amd", "lib", "libunivl_hip_trace.so"))

import torch  # noqa: E402

from univl_amd import _lib, ops  # noqa: E402

DEV = "cuda"
bf = torch.bfloat16


def med(x):
    x = sorted(x)
    return x[len(x) // 2]


def run(name, launch, split=0, reps=12, rot=1):
    """launch(i): enqueue the launch under test with operand set i (i in range(rot))."""
    L = _lib.lib()
    cap = 4096
    trace = torch.zeros(cap * 4, dtype=torch.int64, device=DEV)
    st = torch.zeros(2, dtype=torch.int64, device=DEV)
    L.univl_trace_set.argtypes = [C.c_void_p, C.c_int]
    assert L.univl_trace_set(C.c_void_p(trace.data_ptr()), cap) == 0
    graphs = []
    for i in range(rot):
        launch(i, lambda: None)                       # large-LDS opt-ins etc. outside the capture
    torch.cuda.synchronize()
    for i in range(rot):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            launch(i, lambda: ops.stamp(st[0:1]))      # launch() enqueues its producer kernel, calls the stamp, then the launch under test
            ops.stamp(st[1:2])
        graphs.append(g)
    rows = []
    for r in range(reps):
        trace.zero_()
        torch.cuda.synchronize()
        graphs[r % rot].replay()
        torch.cuda.synchronize()
        t = trace.view(cap, 4).cpu()
        s0, s1 = [int(x) for x in st.cpu()]
        live = t[:, 0] > 0
        if split:
            part = os.environ.get("MB_PART", "all")
            idx = torch.arange(cap)
            if part == "a":
                live = live & (idx < split)
            elif part == "b":
                live = live & (idx >= split)
        t = t[live].double()
        if t.numel() == 0:
            continue
        t0, t1, t2, t3 = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
        has1 = t1 > 0
        u = lambda x: float(x) / 100.0
        rows.append(dict(
            nwg=int(live.sum()), inn=u(t0.min() - s0), skew=u(t0.max() - t0.min()),
            tile1_med=u((t1 - t0)[has1].median()) if has1.any() else 0.0, tile1_max=u((t1 - t0)[has1].max()) if has1.any() else 0.0,
            kloop_med=u((t2 - torch.where(has1, t1, t0)).median()), kloop_max=u((t2 - torch.where(has1, t1, t0)).max()),
            epi_med=u((t3 - t2).median()), epi_max=u((t3 - t2).max()),
            wg_med=u((t3 - t0).median()), wg_max=u((t3 - t0).max()),
            span=u(t3.max() - t0.min()), out=u(s1 - t3.max()), total=u(s1 - s0)))
    rows = rows[2:] if len(rows) > 4 else rows
    if not rows:
        print("%-34s (no workgroups in this part)" % name)
        return None
    m = {k: med([r[k] for r in rows]) for k in rows[0]}
    print("%-34s wgs %4d | in %4.1f skew %4.1f | tile1 %4.1f/%4.1f kloop %4.1f/%4.1f epi %4.1f/%4.1f wg %4.1f/%4.1f | span %5.1f out %4.1f | total %5.1f"
          % (name, m["nwg"], m["inn"], m["skew"], m["tile1_med"], m["tile1_max"], m["kloop_med"], m["kloop_max"], m["epi_med"], m["epi_max"],
             m["wg_med"], m["wg_max"], m["span"], m["out"], m["total"]))
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="192,768")
    ap.add_argument("--warm", action="store_true",
                    help="ONE copy of every weight operand instead of > 256 MB of rotating copies: each replay finds the weights where the "
                         "previous replay left them (L2 / Infinity Cache) -- the upper bound of what prefetching the next launch's weights "
                         "across the dependency edge can buy")
    a = ap.parse_args()
    H, I = 768, 3072
    print("device:", torch.cuda.get_device_name(0), " lib:", os.environ["UNIVL_LIB"])
    for M in [int(x) for x in a.rows.split(",")]:
        tiles = ((M + 63) // 64) * (H // 64)
        splitk = tiles < 128

        def ks(K):
            if not splitk:
                return 1
            k = max(1, (K + 383) // 384)
            while k > 1 and tiles * k > 512:
                k -= 1
            return k
        print("---- rows M = %d (split-K of the N = 768 products: %s)" % (M, splitk))
        # rotating weights: > 256 MB per family so that each replay reads them from HBM
        def pool(n, k):
            cnt = 1 if a.warm else max(2, int(300e6 // (n * k * 2)) + 1)
            return [torch.randn(n, k, device=DEV).to(bf) * 0.02 for _ in range(cnt)]
        x = torch.randn(M, H, device=DEV).to(bf)
        f = torch.randn(M, I, device=DEV).to(bf)
        bias3, biasH, biasI = (torch.randn(n, device=DEV) for n in (3 * H, H, I))
        qkv = torch.empty(M, 3 * H, device=DEV, dtype=bf)
        y32 = torch.zeros(M, H, device=DEV)
        u = torch.empty(M, I, device=DEV, dtype=bf)
        fo = torch.empty(M, I, device=DEV, dtype=bf)
        res = torch.randn(M, H, device=DEV)
        touch = lambda t: t.mul_(1.0)                 # the producing kernel of the step: leaves the activation in L2
        Wq, Wo, W1, W2 = pool(3 * H, H), pool(H, H), pool(I, H), pool(H, I)
        R = len(Wq)
        run("fwd QKV  N2304 K768 ->bf16", lambda i, st: (touch(x), st(), ops.gemm(x, Wq[i % len(Wq)], M, 3 * H, H, out16=qkv, bias=bias3)), rot=min(R, 24))
        run("fwd O    N768 K768 ks%d ->f32" % ks(H), lambda i, st: (touch(x), st(), ops.gemm(x, Wo[i % len(Wo)], M, H, H, out32=y32, bias=biasH, ksplit=ks(H))), rot=24)
        run("fwd FFN1 N3072 K768 gelu", lambda i, st: (touch(x), st(), ops.gemm(x, W1[i % len(W1)], M, I, H, out16=fo, bias=biasI, aux=u, gelu="fwd")), rot=24)
        run("fwd FFN2 N768 K3072 ks%d ->f32" % ks(I), lambda i, st: (touch(f), st(), ops.gemm(f, W2[i % len(W2)], M, H, I, out32=y32, bias=biasH, ksplit=ks(I))), rot=24)
        # the same forward products with a non-temporal hint on the weight operand's LDS-DMA (measurement build: UNIVL_GEMM_NT_B is read per call)
        os.environ["UNIVL_GEMM_NT_B"] = "1"
        run("fwd QKV  N2304 K768 ->bf16  [nt B]", lambda i, st: (touch(x), st(), ops.gemm(x, Wq[i % len(Wq)], M, 3 * H, H, out16=qkv, bias=bias3)), rot=min(R, 24))
        run("fwd O    N768 K768 ks%d [nt B]" % ks(H), lambda i, st: (touch(x), st(), ops.gemm(x, Wo[i % len(Wo)], M, H, H, out32=y32, bias=biasH, ksplit=ks(H))), rot=24)
        run("fwd FFN1 N3072 K768 gelu [nt B]", lambda i, st: (touch(x), st(), ops.gemm(x, W1[i % len(W1)], M, I, H, out16=fo, bias=biasI, aux=u, gelu="fwd")), rot=24)
        run("fwd FFN2 N768 K3072 ks%d [nt B]" % ks(I), lambda i, st: (touch(f), st(), ops.gemm(f, W2[i % len(W2)], M, H, I, out32=y32, bias=biasH, ksplit=ks(I))), rot=24)
        os.environ["UNIVL_GEMM_NT_B"] = "0"
        # backward: dgrad alone, and the pair launches of the step
        dxd = torch.randn(M, H, device=DEV).to(bf)
        du = torch.empty(M, I, device=DEV, dtype=bf)
        dqkv = torch.randn(M, 3 * H, device=DEV).to(bf)
        gW2 = torch.empty(H, I, device=DEV)
        gW1 = torch.empty(I, H, device=DEV)
        gWq = torch.empty(3 * H, H, device=DEV)
        db1 = torch.zeros(I, device=DEV)
        dx32 = torch.zeros(M, H, device=DEV)
        run("dgrad FFN2 N3072 K768 gelu'", lambda i, st: (touch(dxd), st(), ops.gemm(dxd, W2[i % len(W2)], M, I, H, trans_b=True, out16=du, aux=u, gelu="bwd")), rot=24)
        run("wgrad FFN2 [768x3072] K=%d" % M, lambda i, st: (touch(dxd), st(), ops.gemm(dxd, f, H, I, M, trans_a=True, trans_b=True, out32=gW2)), rot=1)

        def pair_ffn2(i, st):
            touch(dxd)
            st()
            d = ops.gemm_desc(dxd, W2[i % len(W2)], M, I, H, trans_b=True, out16=du, aux=u, gelu="bwd")
            w = ops.gemm_desc(dxd, f, H, I, M, trans_a=True, trans_b=True, out32=gW2)
            assert ops.gemm_pair(d, w)

        def pair_ffn1(i, st):
            touch(du)
            st()
            d = ops.gemm_desc(du, W1[i % len(W1)], M, H, I, trans_b=True, out32=dx32, residual=res, ksplit=ks(I))
            w = ops.gemm_desc(du, x, I, H, M, trans_a=True, trans_b=True, out32=gW1, dbias=db1, dbias_atomic=True)
            assert ops.gemm_pair(d, w)

        def pair_qkv(i, st):
            touch(dqkv)
            st()
            d = ops.gemm_desc(dqkv, Wq[i % len(Wq)], M, H, 3 * H, trans_b=True, out32=dx32, residual=res, ksplit=ks(3 * H))
            w = ops.gemm_desc(dqkv, x, 3 * H, H, M, trans_a=True, trans_b=True, out32=gWq)
            assert ops.gemm_pair(d, w)

        pad8 = lambda n: (n + 7) // 8 * 8
        mt = (M + 63) // 64
        dbn = 128 if M >= 384 else 64                 # the pair launch's rectangular form: 64 x 128 dgrad tiles
        csn = lambda cols: pad8((cols + 63) // 64)    # bias-gradient role workgroups sit between the two products
        for part in ("all", "a", "b"):          # a: the dgrad workgroups (first ids), b: the weight-gradient workgroups
            os.environ["MB_PART"] = part
            run("pair FFN2 (dgrad + wgrad) [%s]" % part, pair_ffn2, split=pad8(mt * (I // dbn)), rot=24)
            run("pair FFN1 (dgrad ks%d + wgrad+dbias) [%s]" % (ks(I), part), pair_ffn1, split=pad8(mt * (H // dbn) * ks(I)) + csn(I), rot=24)
            run("pair QKV (dgrad ks%d + wgrad) [%s]" % (ks(3 * H), part), pair_qkv, split=pad8(mt * (H // dbn) * ks(3 * H)), rot=24)
        os.environ["MB_PART"] = "all"
        if M >= 512:
            # what the default plan does NOT do at this size: split-K of the deep N = 768 products, half-width tiles
            for k2 in (2, 3):
                run("fwd FFN2 N768 K3072 FORCED ks%d" % k2, lambda i, st, k2=k2: (touch(f), st(), ops.gemm(f, W2[i % len(W2)], M, H, I, out32=y32, bias=biasH, ksplit=k2)), rot=24)
            for tile in (64128, 12864, 128):
                run("fwd FFN1 N3072 K768 gelu tile %d" % tile, lambda i, st, tile=tile: (touch(x), st(), ops.gemm(x, W1[i % len(W1)], M, I, H, out16=fo, bias=biasI, aux=u, gelu="fwd", tile=tile)), rot=24)
                run("fwd QKV N2304 K768 tile %d" % tile, lambda i, st, tile=tile: (touch(x), st(), ops.gemm(x, Wq[i % len(Wq)], M, 3 * H, H, out16=qkv, bias=bias3, tile=tile)), rot=24)
                run("dgrad FFN2 N3072 K768 gelu' tile %d" % tile, lambda i, st, tile=tile: (touch(dxd), st(), ops.gemm(dxd, W2[i % len(W2)], M, I, H, trans_b=True, out16=du, aux=u, gelu="bwd", tile=tile)), rot=24)

            def pair_ffn1_ks(i, st, k2):
                touch(du)
                st()
                d = ops.gemm_desc(du, W1[i % len(W1)], M, H, I, trans_b=True, out32=dx32, residual=res, ksplit=k2)
                w = ops.gemm_desc(du, x, I, H, M, trans_a=True, trans_b=True, out32=gW1, dbias=db1, dbias_atomic=True)
                assert ops.gemm_pair(d, w)

            def pair_qkv_ks(i, st, k2):
                touch(dqkv)
                st()
                d = ops.gemm_desc(dqkv, Wq[i % len(Wq)], M, H, 3 * H, trans_b=True, out32=dx32, residual=res, ksplit=k2)
                w = ops.gemm_desc(dqkv, x, 3 * H, H, M, trans_a=True, trans_b=True, out32=gWq)
                assert ops.gemm_pair(d, w)
            for k2 in (2, 3):
                run("pair FFN1 dgrad FORCED ks%d [all]" % k2, lambda i, st, k2=k2: pair_ffn1_ks(i, st, k2), rot=24)
                run("pair QKV dgrad FORCED ks%d [all]" % k2, lambda i, st, k2=k2: pair_qkv_ks(i, st, k2), rot=24)
        # reference points: an empty boundary (two stamps back to back) and a trivial elementwise kernel between stamps
        run_empty()


def run_empty():
    st = torch.zeros(3, dtype=torch.int64, device=DEV)
    y = torch.zeros(192 * 768, device=DEV)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        ops.stamp(st[0:1])
        ops.stamp(st[1:2])
        y.mul_(1.0)
        ops.stamp(st[2:3])
    d1, d2 = [], []
    for _ in range(10):
        g.replay()
        torch.cuda.synchronize()
        a, b, c = [int(v) for v in st.cpu()]
        d1.append((b - a) / 100.0)
        d2.append((c - b) / 100.0)
    print("reference: stamp -> stamp %.1f us; stamp -> [192x768 fp32 elementwise] -> stamp %.1f us" % (med(d1), med(d2)))


if __name__ == "__main__":
    main()
