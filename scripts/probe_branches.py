"""Where the two encoder branches of the captured training step start and end, WITHOUT a profiler attached.

    UNIVL_AB=stamps=1 python scripts/probe_branches.py [--batch 4] [--steps 60]
    UNIVL_AB=probe_skip=visual python scripts/probe_branches.py          # the step without the video stack's layers (timing only)

With stamps=1 the plans carry one-thread timestamp kernels (univl_stamp: the device's 100 MHz wall clock) at the fork, at both
branch starts / ends and at the join, forward and backward; this script replays the whole-step hipGraph (graphed.GraphedTrainStep, the
bench configuration) and prints the median position of every stamp relative to the first one, next to the wall time of a step and the
host time of one replay call.  Background: the rocprofv3 kernel trace of the round-3 step (profiles/r03z_graph_replay_kernel_trace.csv.gz)
shows the second branch of each fork starting 0.4 / 0.8 ms after the fork -- under the profiler; this measures the same thing at full speed."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--no-pipeline", action="store_true")
    a = ap.parse_args()
    args = argparse.Namespace(batch=a.batch, dtype="bf16", kind="joint", dropout=0.1)
    from univl_amd import _ab as _uab
    _uab.allow()
    from univl_amd import UniVL, BertAdam
    from univl_amd.graphed import GraphedTrainStep
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base", task_config=B.task_config(args, 1))
    model.to(dev).train()
    opt = B.make_optimizer(model, BertAdam)
    g = torch.Generator(device="cpu").manual_seed(1234)
    Bn, W, F = a.batch, 48, 48
    ids = torch.randint(1000, 30522, (Bn, 1, W), generator=g)
    inp = [ids.to(dev), torch.zeros(Bn, 1, W, dtype=torch.int64, device=dev), torch.ones(Bn, 1, W, dtype=torch.int64, device=dev),
           torch.randn(Bn, 1, F, 1024, generator=g, dtype=torch.float64).to(dev), torch.ones(Bn, 1, F, dtype=torch.int64, device=dev)]
    gs = GraphedTrainStep(model, opt, max_grad_norm=1.0, warmup=3, pipeline_optimizer=not a.no_pipeline, persistent_inputs=True)
    for _ in range(8):
        float(gs(*inp))
    torch.cuda.synchronize()
    for _ in range(40):              # pre-heat
        float(gs(*inp))
    stamps = getattr(model, "_stamps", None)
    rel, host, wall = [], [], []
    for _ in range(a.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = gs(*inp)
        t1 = time.perf_counter()
        float(loss)
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e6)
        wall.append((t2 - t0) * 1e6)
        if stamps is not None:
            names, buf = stamps
            v = buf.cpu().tolist()
            rel.append({n: v[i] for n, i in names.items()})
    med = lambda x: sorted(x)[len(x) // 2]
    print("batch %d  pipeline %s  skip=%s  mode=%s ride=%s" % (a.batch, not a.no_pipeline, _uab.get("probe_skip") or "-", gs.mode, gs.ride))
    print("wall per step (replay call + float(loss)): median %.1f us, min %.1f;  host time of the replay call alone: median %.1f us" %
          (med(wall), min(wall), med(host)))
    if rel:
        base = "f_begin"
        order = sorted(rel[0], key=lambda n: med([r[n] - r[base] for r in rel]))
        print("stamp positions relative to f_begin (us, 100 MHz device clock; median over %d replays):" % len(rel))
        for n in order:
            d = [(r[n] - r[base]) / 100.0 for r in rel]
            print("  %-14s %9.1f   (min %.1f max %.1f)" % (n, med(d), min(d), max(d)))
        span = [(r["b_end"] - r["f_begin"]) / 100.0 for r in rel]
        print("  f_begin -> b_end span: median %.1f us (the step also holds the riding update's prologue in front of f_begin and the clip behind b_end)" % med(span))


if __name__ == "__main__":
    main()
