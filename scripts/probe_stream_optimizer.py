"""Ceiling of a STREAMING optimizer beside the forward (timing only; the concurrent form races on the gradient buffer, results are garbage).

The riding update (BertAdam chunks as extra workgroups of the next forward's launches) moves its 4.6 GB at ~4.5 TB/s; the update alone reaches
5.9 - 6.8 TB/s.  Would ONE persistent update kernel on a second stream, with device flags instead of launch boundaries, do better?  Before
building the flags: the step WITHOUT its update (forward + backward + clip: graph A) and the update alone (graph B), replayed
  seq   A then B on one stream           (no pipelining)
  conc  B on a side stream beside A      (what a flag-synchronised streaming update could reach at best: no waits at all)
  ride  graphed.GraphedTrainStep(pipeline_optimizer=True), the product path
interleaved, HIP events around blocks of `--steps` iterations.

    python scripts/probe_stream_optimizer.py [--batch 4] [--steps 50] [--rounds 3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench as B  # noqa: E402


def build(a, dev):
    from univl_amd import UniVL, BertAdam
    args = argparse.Namespace(batch=a.batch, dtype="bf16", kind="joint", dropout=0.1)
    torch.manual_seed(0)
    model = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base", task_config=B.task_config(args, 1))
    model.to(dev).train()
    opt = B.make_optimizer(model, BertAdam)
    g = torch.Generator(device="cpu").manual_seed(1234)
    Bn, W, F = a.batch, 48, 48
    ids = torch.randint(1000, 30522, (Bn, 1, W), generator=g)
    inp = [ids.to(dev), torch.zeros(Bn, 1, W, dtype=torch.int64, device=dev), torch.ones(Bn, 1, W, dtype=torch.int64, device=dev),
           torch.randn(Bn, 1, F, 1024, generator=g, dtype=torch.float64).to(dev), torch.ones(Bn, 1, F, dtype=torch.int64, device=dev)]
    return model, opt, inp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--blocks", default="256,512,1024", help="grid caps of the forked update")
    a = ap.parse_args()
    from univl_amd import _ab
    _ab.allow()
    from univl_amd import clip_grad_norm_
    from univl_amd.engine import no_gc
    from univl_amd.graphed import GraphedTrainStep
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    # product path
    m1, o1, inp1 = build(a, dev)
    gs = GraphedTrainStep(m1, o1, max_grad_norm=1.0, warmup=3, pipeline_optimizer=True, persistent_inputs=True)
    for _ in range(10):
        float(gs(*inp1))
    # split graphs
    m2, o2, inp2 = build(a, dev)
    m2.auto_ride = False
    params = [p for p in m2.parameters()]
    for _ in range(3):
        loss = m2(*inp2)
        loss.backward()
        clip_grad_norm_(params, 1.0)
        o2.step()
        o2.zero_grad()
    torch.cuda.synchronize()
    gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with no_gc(), torch.cuda.graph(gA, capture_error_mode="thread_local"):
        loss = m2(*inp2)
        loss.backward()
        clip_grad_norm_(params, 1.0)
    with no_gc(), torch.cuda.graph(gB, pool=gA.pool(), capture_error_mode="thread_local"):
        o2.step()
        o2.zero_grad()
    side = torch.cuda.Stream(device=dev)
    # ONE graph with a fork: the update as a side branch beside forward + backward + clip (joined at the end).  Two graph LAUNCHES on two
    # streams do not overlap on this runtime (`conc` == `seq` in the first run of this probe), a branch inside one graph does.
    loss = m2(*inp2)
    loss.backward()
    clip_grad_norm_(params, 1.0)
    torch.cuda.synchronize()
    gF = torch.cuda.CUDAGraph()
    with no_gc(), torch.cuda.graph(gF, pool=gA.pool(), capture_error_mode="thread_local"):
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            o2.step()
        o2.zero_grad()
        loss = m2(*inp2)
        loss.backward()
        clip_grad_norm_(params, 1.0)
        cur.wait_stream(side)

    def fork():
        gF.replay()

    # the same with the update as ONE capped launch (univl_bert_adam_range, max_blocks workgroups walking the chunk table): a grid of
    # thousands of workgroups keeps the dispatcher to itself (fork == seq), a few hundred persistent ones leave room for the chain
    capped = {}
    for nb in [int(x) for x in a.blocks.split(",") if x]:
        loss = m2(*inp2)
        loss.backward()
        clip_grad_norm_(params, 1.0)
        torch.cuda.synchronize()
        g_ = torch.cuda.CUDAGraph()
        with no_gc(), torch.cuda.graph(g_, pool=gA.pool(), capture_error_mode="thread_local"):
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                o2.step(defer=True)
                o2.launch_deferred(groups=[("all", 0, o2._tb.nchunk)], max_blocks=nb)
            o2.zero_grad()
            loss = m2(*inp2)
            loss.backward()
            clip_grad_norm_(params, 1.0)
            cur.wait_stream(side)
        capped[nb] = g_

    def seq():
        gA.replay()
        gB.replay()

    def conc():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            gB.replay()
        gA.replay()
        cur.wait_stream(side)

    def only_a():
        gA.replay()

    def only_b():
        gB.replay()

    def ride():
        gs(*inp1)

    forms = [("ride", ride), ("seq", seq), ("conc", conc), ("fork (one graph)", fork), ("A alone (no update)", only_a), ("B alone (update)", only_b)]
    forms += [("fork, update on %d workgroups" % nb, g_.replay) for nb, g_ in capped.items()]
    for _, f in forms:
        for _ in range(20):
            f()
    torch.cuda.synchronize()
    for r in range(a.rounds):
        for name, f in forms:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.steps):
                f()
            e1.record()
            torch.cuda.synchronize()
            print("round %d  %-34s %.4f ms per step" % (r + 1, name, e0.elapsed_time(e1) / a.steps), flush=True)


if __name__ == "__main__":
    main()
