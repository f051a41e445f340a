"""Driver for PMC passes over the small-M GEMMs (run under rocprofv3 --kernel-trace --pmc ...): 30 eager launches each of
the qkv / ffn1 / out-proj forward shapes at T = 192 with rotating (HBM-cold) weights."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univl_amd import ops
A = torch.randn(192, 768, device="cuda").to(torch.bfloat16)
for (N, gelu) in ((2304, None), (3072, "fwd"), (768, None)):
    Ws = [torch.randn(N, 768, device="cuda").to(torch.bfloat16) * 0.05 for _ in range(80)]
    out = torch.zeros(192, N, device="cuda", dtype=torch.bfloat16)
    aux = torch.zeros(192, N, device="cuda", dtype=torch.bfloat16) if gelu else None
    for i in range(30):
        ops.gemm(A, Ws[i], 192, N, 768, out16=out, aux=aux, gelu=gelu)
    torch.cuda.synchronize()
