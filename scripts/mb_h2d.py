"""How should a pageable float64 (B,1,48,1024) host batch reach HBM?  """
import time
import torch

B = 4
src = torch.randn(B, 1, 48, 1024, dtype=torch.float64)
dst = torch.empty(B * 48, 1024, dtype=torch.float64, device="cuda")
dst32 = torch.empty(B * 48, 1024, dtype=torch.float32, device="cuda")
pin = torch.empty(B * 48, 1024, dtype=torch.float64, pin_memory=True)


def t(f, n=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def direct():
    dst.copy_(src.reshape(B * 48, 1024), non_blocking=True)


def pinned():
    pin.copy_(src.reshape(B * 48, 1024))
    dst.copy_(pin, non_blocking=True)


def cast_direct():
    dst32.copy_(src.reshape(B * 48, 1024).float(), non_blocking=True)


def pin_only():
    pin.copy_(src.reshape(B * 48, 1024))


print("pageable -> device (f64, %.1f MB): %.3f ms" % (src.numel() * 8 / 1e6, t(direct)))
print("pageable -> pinned -> device      : %.3f ms  (host copy into pinned alone %.3f ms)" % (t(pinned), t(pin_only)))
print("host cast f32 + pageable -> device: %.3f ms" % t(cast_direct))

# ---- realistic loop: stage 7 tensors, run some GPU work, read a scalar back (the training loop's float(loss))
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univl_amd.steps import stage_input
srcs = [torch.randint(0, 1000, (B, 1, 48)) for _ in range(5)] + [src, src.clone()]
dsts = [torch.empty(B, 48, dtype=torch.int64, device="cuda") for _ in range(5)] + [dst, dst.clone()]
work = torch.randn(4096, 4096, device="cuda")
acc = torch.zeros(1, device="cuda")


def loop(kind, n=30):
    ts = []
    for i in range(n + 5):
        t0 = time.perf_counter()
        for s_, d_ in zip(srcs, dsts):
            if kind == "pinned":
                stage_input(d_, s_)
            else:
                d_.copy_(s_.reshape(d_.shape), non_blocking=True)
        t1 = time.perf_counter()
        for _ in range(20):
            acc.add_(work.sum())
        float(acc)
        if i >= 5:
            ts.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
    st = sorted(x[0] for x in ts)
    print("%-8s staging median %.3f ms  max %.3f ms ; step median %.3f ms" % (kind, st[len(st) // 2], st[-1], sorted(x[1] for x in ts)[len(ts) // 2]))


loop("direct")
loop("pinned")
loop("direct")
loop("pinned")
