"""The unchanged call pattern of main_task_retrieval.py:197-198,333-352 -- stock torch DistributedDataParallel around
the model, loss.backward(), clip, BertAdam -- with two ranks.  The GPU box has one MI355X, so both ranks share cuda:0
and talk over gloo; the code path (DDP wrapper that ignores our parameters, built-in bucketed gradient exchange,
rank-0 parameter broadcast) is the one RCCL runs with one GPU per rank.

Checked: every rank ends up with the MEAN over ranks of the per-rank gradients (what DDP computes in the reference),
identical parameters on all ranks after optimizer steps, and explicit enable_data_parallel() + GraphedTrainStep
(segmented hipGraphs around real collectives) agreeing with the eager DDP-wrapped loop."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE, ROWS, STEPS = "joint_small", 4, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup_path():
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _model_and_batch(lo, hi):
    _setup_path()
    import univl_oracle as O
    from make_golden import case_config
    from test_model_gpu import build
    cfg, _, dseed = case_config(CASE)
    model, _ = build(cfg, torch.float32)
    model.train()
    full = O.synthetic_batch(cfg, ROWS, seed=dseed)
    b = {k: v[lo:hi].to("cuda") for k, v in full.items()}
    args = (b["input_ids"], b["token_type_ids"], b["attention_mask"], b["video"], b["video_mask"])
    kw = dict(pairs_masked_text=b["pairs_masked_text"], pairs_token_labels=b["pairs_token_labels"],
              masked_video=b["masked_video"], video_labels_index=b["video_labels_index"])
    return model, args, kw


def _probe_names(model):
    used = model.used_parameter_names()
    return [used[0], used[2], used[len(used) // 3], used[len(used) // 2], "normalize_video.visual_norm2d.bias", used[-1]]


def _worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from univl_amd import BertAdam, clip_grad_norm_
    per = ROWS // world
    model, args, kw = _model_and_batch(rank * per, (rank + 1) * per)
    if rank == 1:                                   # DDP / enable_data_parallel must broadcast rank 0's parameters
        with torch.no_grad():
            model.flat.p32.mul_(1.5)
    opt = BertAdam(model.parameters(), lr=1e-4, warmup=-1, t_total=-1, weight_decay=0.01, max_grad_norm=1.0)
    names = _probe_names(model)
    losses, grads = [], None
    if mode == "stock_ddp":
        wrapped = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0, find_unused_parameters=True)
        for it in range(STEPS):
            loss = wrapped(*args, **kw)
            loss.backward()
            if it == 0:
                grads = {n: model.flat.g(n).detach().float().cpu().clone() for n in names}
            clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            opt.zero_grad()
            losses.append(float(loss))
        assert model._implicit_dp and model._reducer is not None
    else:                                           # explicit API + hipGraph replay with real collectives in between
        from univl_amd.graphed import GraphedTrainStep
        model.enable_data_parallel()
        gs = GraphedTrainStep(model, opt, max_grad_norm=1.0, warmup=1)
        for it in range(STEPS):
            losses.append(float(gs(*args, **kw)))
        assert gs.mode == "segmented"
    final = {n: model.flat.w32(n).detach().float().cpu().clone() for n in names}
    torch.cuda.synchronize()
    as_np = lambda d: None if d is None else {k: v.numpy() for k, v in d.items()}     # by value, not via shared memory
    q.put((rank, losses, as_np(grads), as_np(final)))
    dist.barrier()
    dist.destroy_process_group()


def _run(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
    as_t = lambda d: None if d is None else {k: torch.from_numpy(v) for k, v in d.items()}
    return [(r, l, as_t(g), as_t(f)) for r, l, g, f in res]


def test_stock_ddp_wrapper_and_graphed_data_parallel_two_ranks():
    res = _run("stock_ddp")
    (r0, l0, g0, f0), (r1, l1, g1, f1) = res
    # expected gradients of step 0: mean over ranks of the per-rank (local-batch) gradients, rank 0's initial weights
    exp = None
    for lo in (0, 2):
        model, args, kw = _model_and_batch(lo, lo + 2)
        model(*args, **kw).backward()
        names = _probe_names(model)
        cur = {n: model.flat.g(n).detach().float().cpu().clone() for n in names}
        exp = cur if exp is None else {n: 0.5 * (exp[n] + cur[n]) for n in names}
        del model
    for n in exp:
        scale = float(exp[n].abs().max()) + 1e-12
        assert float((g0[n] - exp[n]).abs().max()) < 1e-4 * scale + 1e-9, n
        assert float((g1[n] - g0[n]).abs().max()) < 1e-5 * scale + 1e-9, n     # every rank holds the same mean
        assert torch.allclose(f0[n], f1[n], rtol=0, atol=1e-6), n            # replicas stay in lock step
    # explicit enable_data_parallel + GraphedTrainStep (captured segments, host-issued collectives) == eager stock DDP
    res2 = _run("graphed")
    (_, gl0, _, gf0), (_, gl1, _, gf1) = res2
    assert max(abs(a - b) for a, b in zip(gl0, l0)) < 2e-4 and max(abs(a - b) for a, b in zip(gl1, l1)) < 2e-4
    for n in f0:
        assert float((gf0[n] - f0[n]).abs().max()) < 5e-5, n
        assert torch.allclose(gf0[n], gf1[n], rtol=0, atol=1e-6), n
