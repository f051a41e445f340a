"""Bitwise comparison of univl_attention_fwd / _bwd between two builds of the library (default: lib/libunivl_hip_base.so, a copy of the
previous build, against the product library) over the shapes the plans use, with and without dropout, causal masks and ragged key masks.

    python scripts/cmp_attention_libs.py [other.so]

The LDS images of round 4 (attention.hip: one pitch per read direction, two images in the backward) must not change a single bit."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from univl_amd import ops, _lib  # noqa: E402

dev, bf, dt, H, NH = "cuda", torch.bfloat16, _lib.DT_BF16, 768, 12


def main():
    other = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "univl_amd", "lib", "libunivl_hip_base.so")
    A = _lib.lib()
    Bl = C.CDLL(other)
    for L in (Bl,):
        for n in ("univl_attention_fwd", "univl_attention_bwd"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p]
            getattr(L, n).restype = C.c_int32
    h = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bad = 0
    cases = []
    for B in (1, 4, 16):
        for Sq, Sk in ((48, 48), (20, 20), (40, 56), (64, 64), (96, 96), (128, 128), (224, 224), (128, 224), (48, 96), (1, 224), (1, 37)):
            for causal in ((False, True) if Sq == Sk else (False,)):
                for p in (0.0, 0.1):
                    cases.append((B, Sq, Sk, causal, p))
    for B, Sq, Sk, causal, p in cases:
        g = torch.Generator(device=dev).manual_seed(B * 7919 + Sq * 31 + Sk)
        q = torch.randn(B * Sq, H, device=dev, generator=g).to(bf)
        kv = torch.randn(B * Sk, 2 * H, device=dev, generator=g).to(bf)
        dout = torch.randn(B * Sq, H, device=dev, generator=g).to(bf)
        lens = torch.randint(1, Sk + 1, (B,), generator=torch.Generator().manual_seed(Sk + B))
        lens[0] = Sk
        mask = (torch.arange(Sk)[None] < lens[:, None]).long().to(dev)
        res = []
        for L in (A, Bl):
            out = torch.zeros(B * Sq, H, device=dev, dtype=bf)
            lse = torch.zeros(B * NH * Sq, device=dev)
            dq = torch.zeros(B * Sq, H, device=dev, dtype=bf)
            dkv = torch.zeros(B * Sk, 2 * H, device=dev, dtype=bf)
            d = ops.attention_desc(dt, B, NH, Sq, Sk, (q, 0), H, (kv, 0), 2 * H, (kv, H), 2 * H, out, H, lse, key_mask=mask, causal=causal,
                                   p_drop=p, seed=11, offset=5 << 40, dout=dout, lddo=H, dq=(dq, 0), lddq=H, dk=(dkv, 0), lddk=2 * H,
                                   dv=(dkv, H), lddv=2 * H)
            assert L.univl_attention_fwd(C.byref(d), h) == 0
            assert L.univl_attention_bwd(C.byref(d), h) == 0
            torch.cuda.synchronize()
            res.append((out, lse, dq, dkv))
        same = all(torch.equal(a.view(torch.int16 if a.dtype == bf else torch.int32), b.view(torch.int16 if b.dtype == bf else torch.int32))
                   for a, b in zip(*res))
        if not same:
            bad += 1
            print("DIFF  B %d Sq %d Sk %d causal %s p %.1f: max |d| out %.3e dq %.3e dkv %.3e" % (
                B, Sq, Sk, causal, p, float((res[0][0].float() - res[1][0].float()).abs().max()),
                float((res[0][2].float() - res[1][2].float()).abs().max()), float((res[0][3].float() - res[1][3].float()).abs().max())))
    print("%d cases, %d differ  (%s vs %s)" % (len(cases), bad, _lib.LIB_PATH, other))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
