"""Per-launch cost under hipGraph replay (dependent chains of 200) of the non-GEMM kernels at T = 192."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univl_amd import ops, _lib
dev = "cuda"
bf = torch.bfloat16
dt = _lib.DT_BF16
T, H, B, S, NH = 192, 768, 4, 48, 12

def chain(name, f, reps=200):
    for _ in range(5): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print("%-28s %6.2f us/launch" % (name, e0.elapsed_time(e1) * 1e3 / reps), flush=True)

x = torch.randn(T, H, device=dev); res = torch.randn(T, H, device=dev)
gm = torch.ones(H, device=dev); bt = torch.zeros(H, device=dev)
y = torch.empty(T, H, device=dev); st = torch.empty(T, 2, device=dev); o32 = torch.empty(T, H, device=dev); o16 = torch.empty(T, H, device=dev, dtype=bf)
chain("layernorm fwd (res, 2 outs)", lambda: ops.layernorm_fwd(dtype=dt, rows=T, N=H, x=x, residual=res, gamma=gm, beta=bt, y=y, stats=st, out32=o32, out16=o16, p_pre=0.1, seed=1, off_pre=3))
dout = torch.randn(T, H, device=dev); dx = torch.empty(T, H, device=dev); dxd = torch.empty(T, H, device=dev, dtype=bf)
dg = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev); dbias = torch.zeros(H, device=dev)
chain("layernorm bwd (+dbias)", lambda: ops.layernorm_bwd(dtype=dt, rows=T, N=H, gamma=gm, y=y, stats=st, dout=dout, dx32=dx, dxd16=dxd, dgamma=dg, dbeta=db, dbias=dbias, p_pre=0.1, seed=1, off_pre=3))
chain("layernorm bwd (no dbias/drop)", lambda: ops.layernorm_bwd(dtype=dt, rows=T, N=H, gamma=gm, y=y, stats=st, dout=dout, dx32=dx, dgamma=dg, dbeta=db))
qkv = torch.randn(T, 3 * H, device=dev).to(bf); ctx = torch.empty(T, H, device=dev, dtype=bf); lse = torch.empty(B * NH * S, device=dev)
mask = torch.ones(B, S, dtype=torch.int64, device=dev)
chain("attention fwd", lambda: ops.attention_fwd(dt, B, NH, S, S, (qkv, 0), 3 * H, (qkv, H), 3 * H, (qkv, 2 * H), 3 * H, ctx, H, lse, key_mask=mask, p_drop=0.1, seed=1, offset=5))
dctx = torch.randn(T, H, device=dev).to(bf); dqkv = torch.empty(T, 3 * H, device=dev, dtype=bf)
chain("attention bwd", lambda: ops.attention_bwd(dt, B, NH, S, S, (qkv, 0), 3 * H, (qkv, H), 3 * H, (qkv, 2 * H), 3 * H, ctx, H, lse, key_mask=mask, p_drop=0.1, seed=1, offset=5, dout=dctx, lddo=H, dq=(dqkv, 0), lddq=3 * H, dk=(dqkv, H), lddk=3 * H, dv=(dqkv, 2 * H), lddv=3 * H))
