"""Host-side profile of the unchanged (eager) training loop with auto-graphed plans."""
import argparse, cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from univl_amd import _ab as _uab
_uab.allow()
from univl_amd import UniVL, BertAdam, clip_grad_norm_
args = argparse.Namespace(batch=4, dtype="bf16", dropout=0.1)
model = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base", task_config=bench.task_config(args, 1)).to("cuda").train()
opt = bench.make_optimizer(model, BertAdam)
B, W, F = 4, 48, 48
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30522, (B, 1, W), generator=g).cuda()
z = torch.zeros(B, 1, W, dtype=torch.int64, device="cuda"); o = torch.ones(B, 1, W, dtype=torch.int64, device="cuda")
video = torch.randn(B, 1, F, 1024, generator=g, dtype=torch.float64).cuda(); vm = torch.ones(B, 1, F, dtype=torch.int64, device="cuda")
params = list(model.parameters())
def step():
    loss = model(ids, z, o, video, vm)
    loss.backward()
    clip_grad_norm_(params, 1.0)
    opt.step()
    opt.zero_grad()
    return float(loss)
for _ in range(6): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): step()
torch.cuda.synchronize(); print("ms/step %.3f" % ((time.perf_counter() - t0) / 30 * 1e3))
# host time without waiting for the GPU: same loop but no float()
def step_nosync():
    loss = model(ids, z, o, video, vm); loss.backward(); clip_grad_norm_(params, 1.0); opt.step(); opt.zero_grad()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): step_nosync()
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host-only ms/step %.3f (GPU drained after %.3f more ms)" % ((t1 - t0) / 30 * 1e3, (time.perf_counter() - t1) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step_nosync()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
