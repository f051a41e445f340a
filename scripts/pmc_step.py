"""Driver for the PMC passes over a WHOLE training step (round 2): run under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python scripts/pmc_step.py          (pass 1)
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python scripts/pmc_step.py          (pass 2)
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python scripts/pmc_step.py [batch]   (pass 3)
Kernels are enqueued one by one (no hipGraph: every kernel is its own dispatch with its own counter row).  Order:
(a) the calibration kernel with exactly known traffic (univl_cast_bf16 over the flat parameter buffer: 4 B read + 2 B
written per element), (b) STEPS training steps of the bench configuration.  scripts/pmc_step_parse.py sums the GEMM
family's counters per step."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["UNIVL_AB"] = "auto_graph=0"
import bench  # noqa: E402

STEPS = 4


def main():
    from univl_amd import _ab as _uab
    _uab.allow()
    from univl_amd import UniVL, BertAdam, clip_grad_norm_, ops
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    args = argparse.Namespace(batch=B, dtype="bf16", dropout=0.1)
    torch.manual_seed(0)
    tc = bench.task_config(args, 1)
    model = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base", task_config=tc)
    model.to("cuda").train()
    model.auto_graph = False
    # round 5: BertAdam.step() in the plain loop would leave its update to ride in the next forward's products -- the GEMM family's
    # counters would then hold 2.7 GB of optimizer bytes per step.  Here the update goes out as its own launches ("adam" family) so
    # that the family's FETCH / WRITE counters are the products' bytes (with the LayerNorms folded into them, as in the step).
    model.auto_ride = False
    opt = bench.make_optimizer(model, BertAdam)
    W, F = 48, 48
    g = torch.Generator(device="cpu").manual_seed(1234)
    ids = torch.randint(1000, 30522, (B, 1, W), generator=g).cuda()
    video = torch.randn(B, 1, F, 1024, generator=g, dtype=torch.float64).cuda()
    ones_w = torch.ones(B, 1, W, dtype=torch.int64, device="cuda")
    ones_f = torch.ones(B, 1, F, dtype=torch.int64, device="cuda")
    fl = model.flat
    for _ in range(3):
        ops.cast_bf16(fl.p32, fl.p16)            # calibration rows ("cast_kernel")
    torch.cuda.synchronize()
    for _ in range(STEPS):
        loss = model(ids, torch.zeros_like(ones_w), ones_w, video, ones_f)
        loss.backward()
        float(loss)
        clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
    torch.cuda.synchronize()
    print("pmc_step: batch %d, %d steps, %d flat elements" % (B, STEPS, fl.total))


if __name__ == "__main__":
    main()
