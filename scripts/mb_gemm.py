"""Micro-benchmark: per-launch time of the hot-path GEMM shapes at T=192, hot vs cold weights."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univl_amd import ops
dev = "cuda"
def bench(name, M, N, K, nbuf, reps=300, **kw):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    Ws = [torch.randn(N, K, device=dev).to(torch.bfloat16) * 0.05 for _ in range(nbuf)]
    out32 = torch.zeros(M, N, device=dev) if kw.get("fp32out", True) else None
    out16 = None if out32 is not None else torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    aux = torch.zeros(M, N, device=dev, dtype=torch.bfloat16) if kw.get("gelu") else None
    ks = kw.get("ksplit", 1)
    def run(i):
        ops.gemm(A, Ws[i % nbuf], M, N, K, out32=out32, out16=out16, aux=aux, gelu=kw.get("gelu"), ksplit=ks)
    for i in range(20): run(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(3): run(i)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for i in range(reps): run(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:28s} M={M} N={N} K={K} ks={ks} nbuf={nbuf:3d}: {us:6.2f} us/launch  ({2*M*N*K/us/1e6:7.1f} TF, W {N*K*2/us/1e3:6.1f} GB/s)", flush=True)
for nbuf in (1, 18, 80):
    bench("ffn1 fwd (gelu, bf16 out)", 192, 3072, 768, nbuf, gelu="fwd", fp32out=False)
    bench("qkv fwd (bf16 out)", 192, 2304, 768, nbuf, fp32out=False)
    bench("ffn2 fwd split", 192, 768, 3072, nbuf, ksplit=8)
    bench("ffn2 fwd nosplit", 192, 768, 3072, nbuf, ksplit=1)
    bench("outproj split", 192, 768, 768, nbuf, ksplit=2)
    bench("outproj nosplit", 192, 768, 768, nbuf, ksplit=1)
# trivial kernel floor: layernorm fwd on 192 rows
x = torch.randn(192, 768, device=dev); gm = torch.ones(768, device=dev); bt = torch.zeros(768, device=dev); o = torch.empty_like(x)
from univl_amd import _lib
def ln(): ops.layernorm_fwd(dtype=_lib.DT_F32, rows=192, N=768, x=x, gamma=gm, beta=bt, out32=o)
for _ in range(5): ln()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(300): ln()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print("layernorm fwd 192x768 dependent chain: %.2f us/launch" % (e0.elapsed_time(e1) * 1e3 / 300))
