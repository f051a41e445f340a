import sys, time, argparse, torch
sys.path.insert(0, "/root/repo")
import bench
from univl_amd import UniVL, BertAdam
from univl_amd.graphed import GraphedTrainStep
args = argparse.Namespace(batch=4, dtype="bf16", dropout=0.1)
model = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base", task_config=bench.task_config(args, 1)).to("cuda").train()
opt = bench.make_optimizer(model, BertAdam)
B, W, F = 4, 48, 48
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30522, (B, 1, W), generator=g)
inputs = (ids, torch.zeros(B, 1, W, dtype=torch.int64), torch.ones(B, 1, W, dtype=torch.int64), torch.randn(B, 1, F, 1024, generator=g, dtype=torch.float64), torch.ones(B, 1, F, dtype=torch.int64))
kw = dict(pairs_masked_text=inputs[0], pairs_token_labels=None, masked_video=inputs[3], video_labels_index=None)
gs = GraphedTrainStep(model, opt, warmup=2)
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if i >= 3:
        gs._stage(inputs, kw); torch.cuda.synchronize(); t1 = time.perf_counter()
    else:
        t1 = t0
    l = float(gs(*inputs, **kw)); t2 = time.perf_counter()
    print(i, "stage-only %.3f ms, full call %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): float(gs(*inputs, **kw))
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
