"""Calibration of FETCH_SIZE / WRITE_SIZE IN THE GEMM FAMILY'S OWN ACCESS PATTERN (guide: "calibrate on a known byte
count in your own access pattern before trusting an absolute").  Run under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python scripts/pmc_gemm_calib.py      and again with WRITE_SIZE
The family's counters were calibrated on cast_kernel (wide, lane-linear 16-byte loads).  The products read their operands by LDS-DMA
with an XOR-swizzled SOURCE address per lane (16-byte pieces of a 128 / 256-byte row in permuted lane order) and write fp32 tiles as
64-byte row segments -- other request shapes.  Every launch below has EXACTLY known unique bytes and HBM-cold operands (each launch its
own weight copy out of > 300 MB, outputs never read back):
    fwd1   one row tile (M = 64):  weights [3072, 768] bf16 read once = 4.72 MB           (K-major B)
    fwd3   three row tiles (M = 192, the step's shape): the same 4.72 MB if the row tiles of a weight tile share an L2 (the tile map's claim)
    dgrad1 / dgrad3   the same with T-major B ([768, 3072] read as the dgrad reads it)
    wgrad  [768 x 3072] fp32 output written once = 9.44 MB, operands 2 x 192 rows (tiny)  (T-major A and B, NT stores as in the step)
scripts/pmc_gemm_calib_parse.py prints bytes per counter unit for each against cast_kernel's."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from univl_amd import _ab, _lib, ops      # noqa: E402

_ab.allow()
DEV, bf = "cuda", torch.bfloat16
H, I = 768, 3072
n_el = 64 << 20
src = torch.randn(n_el, device=DEV)
dst = torch.empty(n_el, device=DEV, dtype=bf)
for _ in range(3):
    ops.cast_bf16(src, dst)                      # "cast_kernel": 4 B read + 2 B written per element
torch.cuda.synchronize()
R = 24
W1 = [torch.randn(I, H, device=DEV).to(bf) * 0.02 for _ in range(3 * R)]       # 4.72 MB each, 340 MB in all
flush = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)


def cold():
    flush.add_(1)                                # 1 GB of traffic between measured launches: L2 / Infinity Cache hold nothing of theirs
    torch.cuda.synchronize()


for M, tag in ((64, "fwd1"), (192, "fwd3")):
    x = torch.randn(M, H, device=DEV).to(bf)
    out = torch.empty(M, I, device=DEV, dtype=bf)
    for i in range(R):
        cold()
        ops.gemm(x, W1[i], M, I, H, out16=out, tile=64)
for M, tag in ((64, "dgrad1"), (192, "dgrad3")):
    dy = torch.randn(M, H, device=DEV).to(bf)
    du = torch.empty(M, I, device=DEV, dtype=bf)
    for i in range(R):
        cold()
        # W2 = nn.Linear(3072 -> 768).weight [768, 3072]; dgrad: dU[M, 3072] = dY[M, 768] . W2  (T-major B)
        ops.gemm(dy, W1[R + i].view(H, I), M, I, H, trans_b=True, out16=du, tile=64)
dy = torch.randn(192, H, device=DEV).to(bf)
f = torch.randn(192, I, device=DEV).to(bf)
gW = [torch.empty(H, I, device=DEV) for _ in range(R)]
for i in range(R):
    cold()
    d = ops.gemm_desc(dy, f, H, I, 192, trans_a=True, trans_b=True, out32=gW[i])
    d.flags |= _lib.GEMM_NT_OUT
    _lib.check(_lib.lib().univl_gemm(ops._BYREF(d), ops._stream()), "gemm")
torch.cuda.synchronize()
print("pmc_gemm_calib: cast %d elements; fwd1 fwd3 dgrad1 dgrad3 wgrad x %d launches" % (n_el, R))
