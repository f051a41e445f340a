"""Probe (round 2): does hipGraph capture survive REPEATED fork / join between a chain stream and two alternating side
streams -- the pattern of "background" weight gradients (engine.EncoderStack, UNIVL_WGRAD_BLOCKS)?  Pure torch ops, no
univl_amd code: stream A runs a chain of small kernels; after every step a side stream (alternating B0 / B1) is forked from
A, runs one kernel, and A joins that side stream two steps later.  Prints one line per variant."""
import sys

import torch


def run(mode, steps, use_join_every):
    dev = torch.device("cuda")
    x = torch.zeros(1 << 16, device=dev)
    ys = [torch.zeros(1 << 20, device=dev) for _ in range(2)]
    B = [torch.cuda.Stream(), torch.cuda.Stream()]
    g = torch.cuda.CUDAGraph()
    kw = {} if mode == "global" else dict(capture_error_mode=mode)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, **kw):
        A = torch.cuda.current_stream()
        for l in range(steps):
            if use_join_every and l >= 2:
                A.wait_stream(B[l % 2])
            x.add_(1.0)
            x.mul_(1.0001)
            B[l % 2].wait_stream(A)
            with torch.cuda.stream(B[l % 2]):
                ys[l % 2].add_(x.sum())
        A.wait_stream(B[0])
        A.wait_stream(B[1])
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    return float(x[0]), float(ys[0][0])


if __name__ == "__main__":
    for mode in ("global", "thread_local"):
        for steps in (4, 12, 18):
            for je in (False, True):
                print("mode=%s steps=%d join_every=%s ->" % (mode, steps, je), end=" ", flush=True)
                print(run(mode, steps, je), flush=True)
    print("capture fork/join probe: all variants completed")
