"""Attention core alone: us per launch of univl_attention_fwd / _bwd under hipGraph replay (chains of 100 launches), per problem size.

    python scripts/mb_attention.py                                   # the product library
    UNIVL_LIB=univl_amd/lib/libunivl_hip_base.so python scripts/mb_attention.py         # another build of it (A/B)
    UNIVL_LIB=univl_amd/lib/libunivl_hip_trace.so UNIVL_ATTN_DUAL_MAX=0 python scripts/mb_attention.py   # measurement build: no two-image staging

Round 4: one LDS pitch per way a staged matrix is read (144 B K-major, 160 B transpose-read) and two images of the matrices the
backward reads both ways (attention.hip: AttnCfg::PK / PT, DUAL)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univl_amd import ops, _lib  # noqa: E402

dev, bf, dt, H, NH = "cuda", torch.bfloat16, _lib.DT_BF16, 768, 12


def chain(f, reps=100):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            f()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def case(B, Sq, Sk, causal=False):
    Tq, Tk = B * Sq, B * Sk
    gen = torch.Generator(device=dev).manual_seed(B * 1000 + Sq)
    q = torch.randn(Tq, H, device=dev, generator=gen).to(bf)
    kv = torch.randn(Tk, 2 * H, device=dev, generator=gen).to(bf)
    ctx = torch.empty(Tq, H, device=dev, dtype=bf)
    lse = torch.empty(B * NH * Sq, device=dev)
    mask = torch.ones(B, Sk, dtype=torch.int64, device=dev)
    dctx = torch.randn(Tq, H, device=dev, generator=gen).to(bf)
    dq = torch.empty(Tq, H, device=dev, dtype=bf)
    dkv = torch.empty(Tk, 2 * H, device=dev, dtype=bf)
    fw = lambda: ops.attention_fwd(dt, B, NH, Sq, Sk, (q, 0), H, (kv, 0), 2 * H, (kv, H), 2 * H, ctx, H, lse, key_mask=mask,
                                   p_drop=0.1, seed=1, offset=5, causal=causal)
    bw = lambda: ops.attention_bwd(dt, B, NH, Sq, Sk, (q, 0), H, (kv, 0), 2 * H, (kv, H), 2 * H, ctx, H, lse, key_mask=mask,
                                   p_drop=0.1, seed=1, offset=5, causal=causal, dout=dctx, lddo=H, dq=(dq, 0), lddq=H,
                                   dk=(dkv, 0), lddk=2 * H, dv=(dkv, H), lddv=2 * H)
    tf, tb = chain(fw), chain(bw)
    chk = float(ctx.float().abs().sum()) + float(dq.float().abs().sum()) + float(dkv.float().abs().sum())
    print("B %4d  Sq %3d Sk %3d   fwd %7.2f us   bwd %7.2f us   checksum %.6e" % (B, Sq, Sk, tf, tb, chk), flush=True)


if __name__ == "__main__":
    print("library %s  UNIVL_ATTN_DUAL_MAX=%s" % (_lib.LIB_PATH, os.environ.get("UNIVL_ATTN_DUAL_MAX", "-")))
    for B, Sq, Sk in [(4, 48, 48), (16, 48, 48), (128, 48, 48), (4, 96, 96), (4, 128, 128), (16, 96, 96), (4, 224, 224), (16, 224, 224), (4, 128, 224)]:
        case(B, Sq, Sk)
