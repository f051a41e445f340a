"""Cost of reading the loss back every step (main_task_retrieval.py:344) vs queueing the replays back to back."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from univl_amd import UniVL, BertAdam
from univl_amd.graphed import GraphedTrainStep
args = argparse.Namespace(batch=4, dtype="bf16", dropout=0.1)
model = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base", task_config=bench.task_config(args, 1)).to("cuda").train()
opt = bench.make_optimizer(model, BertAdam)
B, W, F = 4, 48, 48
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30522, (B, 1, W), generator=g).cuda()
z = torch.zeros(B, 1, W, dtype=torch.int64, device="cuda"); o = torch.ones(B, 1, W, dtype=torch.int64, device="cuda")
video = torch.randn(B, 1, F, 1024, generator=g, dtype=torch.float64).cuda(); vm = torch.ones(B, 1, F, dtype=torch.int64, device="cuda")
gs = GraphedTrainStep(model, opt, warmup=2, persistent_inputs=True)
for _ in range(6): float(gs(ids, z, o, video, vm))
n = 50
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): float(gs(ids, z, o, video, vm))
torch.cuda.synchronize(); a = (time.perf_counter() - t0) / n * 1e3
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): gs(ids, z, o, video, vm)
torch.cuda.synchronize(); b = (time.perf_counter() - t0) / n * 1e3
print("loss read back every step: %.3f ms/step; replays queued back to back: %.3f ms/step" % (a, b))
