"""Driver for the HBM-traffic PMC passes of the dominant kernel (fused BertAdam update).  Run under
   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python scripts/pmc_adam.py      (pass 1)
   rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python scripts/pmc_adam.py      (pass 2)
It launches (a) a calibration kernel with exactly known traffic in the same access pattern class -- univl_cast_bf16
over the whole flat parameter buffer: reads 4 B, writes 2 B per element, 16-byte lanes -- and (b) the optimizer
kernels via BertAdam.relaunch_last().  scripts/pmc_parse.py turns the two counter CSVs into profiles/r01_adam_pmc.json."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from univl_amd import UniVL, BertAdam, clip_grad_norm_, ops
    args = argparse.Namespace(batch=4, dtype="bf16", dropout=0.1)
    torch.manual_seed(0)
    tc = bench.task_config(args, 1)
    model = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base", task_config=tc)
    model.to("cuda").train()
    opt = bench.make_optimizer(model, BertAdam)
    B, W, F = 4, 48, 48
    g = torch.Generator(device="cpu").manual_seed(1234)
    ids = torch.randint(1000, 30522, (B, 1, W), generator=g).cuda()
    video = torch.randn(B, 1, F, 1024, generator=g, dtype=torch.float64).cuda()
    ones_w = torch.ones(B, 1, W, dtype=torch.int64, device="cuda")
    ones_f = torch.ones(B, 1, F, dtype=torch.int64, device="cuda")
    for _ in range(2):
        loss = model(ids, torch.zeros_like(ones_w), ones_w, video, ones_f)
        loss.backward()
        clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
    loss = model(ids, torch.zeros_like(ones_w), ones_w, video, ones_f)
    loss.backward()
    clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    torch.cuda.synchronize()
    fl = model.flat
    print("elements", fl.total, "params", sum(p.numel() for n, p in model.named_parameters() if ".pooler." not in n))
    for _ in range(4):
        ops.cast_bf16(fl.p32, fl.p16)            # calibration: 4 B read + 2 B written per element
        torch.cuda.synchronize()
    for _ in range(6):
        opt.relaunch_last()
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
