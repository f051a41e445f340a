"""LayerNorm backward at thousands of rows: what the pieces cost -- the full kernel, without the three column sums (dgamma / dbeta / dbias:
3 x 768 fp32 atomics per workgroup), without the regenerated dropout mask -- and the forward for comparison.  20 launches in one hipGraph."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from univl_amd import _lib, ops  # noqa: E402

DEV, N = "cuda", 768


def timed(fn):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20):
            fn()
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        e0.record(); gr.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / 20)
    return best


for rows in (1536, 3072, 6144):
    y = torch.randn(rows, N, device=DEV)
    stats = torch.stack([y.mean(1), 1.0 / (y.var(1, unbiased=False) + 1e-12).sqrt()], 1).contiguous()
    dout = torch.randn(rows, N, device=DEV)
    g = torch.ones(N, device=DEV)
    dx32 = torch.empty(rows, N, device=DEV)
    dxd16 = torch.empty(rows, N, device=DEV, dtype=torch.bfloat16)
    dg, db, dbias = (torch.zeros(N, device=DEV) for _ in range(3))
    seed = torch.zeros(1, dtype=torch.int64, device=DEV)
    base = dict(dtype=_lib.DT_BF16, rows=rows, N=N, gamma=g, y=y, stats=stats, dout=dout, dx32=dx32, dxd16=dxd16, seed_dev=seed, off_pre=1 << 40)
    full = timed(lambda: ops.layernorm_bwd(**base, dgamma=dg, dbeta=db, dbias=dbias, p_pre=0.1))
    nodrop = timed(lambda: ops.layernorm_bwd(**base, dgamma=dg, dbeta=db, dbias=dbias, p_pre=0.0))
    try:
        nocol = timed(lambda: ops.layernorm_bwd(**base, p_pre=0.1))
    except Exception as ex:      # noqa: BLE001
        nocol = float("nan"); print("  (no column sums refused: %s)" % str(ex)[:80])
    x = torch.randn(rows, N, device=DEV); res = torch.randn(rows, N, device=DEV); b = torch.zeros(N, device=DEV)
    out32 = torch.empty(rows, N, device=DEV); out16 = torch.empty(rows, N, device=DEV, dtype=torch.bfloat16)
    fwd = timed(lambda: ops.layernorm_fwd(dtype=_lib.DT_BF16, rows=rows, N=N, x=x, residual=res, gamma=g, beta=b, y=x, stats=stats, out32=out32,
                                           out16=out16, p_pre=0.1, off_pre=1 << 40, seed_dev=seed))
    mb_b, mb_f = rows * N * (4 + 4 + 4 + 2) / 1e6, rows * N * (4 + 4 + 4 + 4 + 2) / 1e6
    print("rows %5d: bwd full %6.2f us (%4.2f TB/s) | no dropout %6.2f | no column sums %6.2f || fwd %6.2f us (%4.2f TB/s)"
          % (rows, full, mb_b / full, nodrop, nocol, fwd, mb_f / fwd))
