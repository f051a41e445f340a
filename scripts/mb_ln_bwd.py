"""LayerNorm backward at the token counts of 4 / 16 / 32 / 128 pairs: microseconds per launch (20 launches in one hipGraph, replayed) for
rows-per-wave 1 / 2 / 4 (measurement build: UNIVL_LN_RPW is read per call there) against the product heuristic (rows / 2048, clamped to 1..4).
Each workgroup (4 waves) folds its rows' dgamma / dbeta / dbias partials through LDS and issues 3 x 768 fp32 atomics: at 768 rows and one
row per wave that is 192 workgroups x 2304 atomics on 2304 addresses."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("UNIVL_LIB", os.path.join(ROOT, "univl_amd", "lib", "libunivl_hip_trace.so"))
import torch  # noqa: E402
from univl_amd import _lib, ops  # noqa: E402

DEV, N = "cuda", 768


def bench(rows, rpw):
    if rpw:
        os.environ["UNIVL_LN_RPW"] = str(rpw)
    else:
        os.environ.pop("UNIVL_LN_RPW", None)
    y = torch.randn(rows, N, device=DEV)
    stats = torch.stack([y.mean(1), 1.0 / (y.var(1, unbiased=False) + 1e-12).sqrt()], 1).contiguous()
    dout = torch.randn(rows, N, device=DEV)
    g = torch.ones(N, device=DEV)
    dx32 = torch.empty(rows, N, device=DEV)
    dxd16 = torch.empty(rows, N, device=DEV, dtype=torch.bfloat16)
    dg, db, dbias = (torch.zeros(N, device=DEV) for _ in range(3))
    seed = torch.zeros(1, dtype=torch.int64, device=DEV)
    kw = dict(dtype=_lib.DT_BF16, rows=rows, N=N, gamma=g, y=y, stats=stats, dout=dout, dx32=dx32, dxd16=dxd16, dgamma=dg, dbeta=db,
              dbias=dbias, p_pre=0.1, off_pre=1 << 40, seed_dev=seed)
    ops.layernorm_bwd(**kw)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20):
            ops.layernorm_bwd(**kw)
    gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        e0.record(); gr.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / 20)
    return best


for rows in (192, 384, 768, 1536, 6144):
    print("rows %5d: " % rows + "  ".join("rpw %s %6.2f us" % (r or "auto", bench(rows, r)) for r in (0, 1, 2, 4)))
