"""Driver for the PMC passes over the 256 x 256 product body (run under rocprofv3 --pmc ...): 12 eager launches each of a forward
(K-major x K-major), a dgrad (K-major x T-major) and a weight-gradient (T-major x T-major) product at 6144 tokens, tile = 256, and the
same products on the 128 tile for comparison."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univl_amd import ops
bf = torch.bfloat16
M, H, I = 6144, 768, 3072
x = torch.randn(M, H, device="cuda").to(bf)
dy = torch.randn(M, I, device="cuda").to(bf)
Wq = [torch.randn(3 * H, H, device="cuda").to(bf) * 0.05 for _ in range(12)]
W1 = [torch.randn(I, H, device="cuda").to(bf) * 0.05 for _ in range(12)]
qkv = torch.zeros(M, 3 * H, device="cuda", dtype=bf)
dx = torch.zeros(M, H, device="cuda")
gW = torch.zeros(I, H, device="cuda")
for tile in (256, 128):
    for i in range(12):
        ops.gemm(x, Wq[i], M, 3 * H, H, out16=qkv, tile=tile)                                  # forward
        ops.gemm(dy, W1[i], M, H, I, trans_b=True, out32=dx, tile=tile)                        # dgrad, 48 K tiles
        ops.gemm(dy, x, I, H, M, trans_a=True, trans_b=True, out32=gW, tile=tile)              # weight gradient, 96 K tiles
    torch.cuda.synchronize()
