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
    from univl_amd import _ab
    _ab.allow()                      # spawned workers do not pass through conftest.py


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
    try:
        _worker_body(rank, world, port, mode, q)
    except Exception as ex:      # noqa: BLE001 -- always answer: a silent death would leave the parent waiting for its timeout
        import traceback
        q.put((rank, "error", "%s: %s\n%s" % (type(ex).__name__, ex, traceback.format_exc()), None))


def _worker_body(rank, world, port, mode, q):
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
    elif mode.startswith("sharded"):                 # reduce-scatter + sharded clip/BertAdam + all-gather of the shadow
        from univl_amd.graphed import GraphedTrainStep
        model.enable_data_parallel(shard_optimizer=True)
        assert model.flat.owned is not None and model._reducer.partition is not None
        if mode == "sharded_graph":
            gs = GraphedTrainStep(model, opt, max_grad_norm=1.0, warmup=1)
            for it in range(STEPS):
                losses.append(float(gs(*args, **kw)))
            assert gs.mode == "segmented"
        else:
            for it in range(STEPS):
                loss = model(*args, **kw)
                loss.backward()
                clip_grad_norm_(model.parameters(), 1.0)
                opt.step()
                opt.zero_grad()
                losses.append(float(loss))
        if model.flat.p16 is not None:               # bf16 compute: the fp32 master of the other ranks' pieces is stale
            with pytest.raises(RuntimeError):
                model.state_dict()
        with pytest.raises(RuntimeError):            # the moments are sharded in every mode
            opt.state_dict()
        model.consolidate_parameters()               # collective: every rank
        opt.consolidate()
        sd = model.state_dict()
        osd = opt.state_dict()
        assert len(sd) > 0 and osd["state"][0]["next_m"].abs().sum() > 0
    else:                                           # explicit API + hipGraph replay with real collectives in between
        from univl_amd.graphed import GraphedTrainStep
        model.enable_data_parallel()
        gs = GraphedTrainStep(model, opt, max_grad_norm=1.0, warmup=1, pipeline_optimizer=True)
        for it in range(STEPS):
            losses.append(float(gs(*args, **kw)))
        assert gs.mode == "segmented"
        gs.flush()                                  # the pipelined BertAdam update of the last iteration
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
    res = []
    for _ in procs:
        r = q.get(timeout=600)
        assert r[1] != "error", r[2]
        res.append(r)
    res.sort(key=lambda r: r[0])
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
    # sharded optimizer (reduce-scatter, clip + BertAdam on 1/world of every bucket, all-gather): same parameters
    for mode in ("sharded", "sharded_graph"):
        (_, sl0, _, sf0), (_, sl1, _, sf1) = _run(mode)
        assert max(abs(a - b) for a, b in zip(sl0, l0)) < 2e-4 and max(abs(a - b) for a, b in zip(sl1, l1)) < 2e-4, mode
        for n in f0:
            assert float((sf0[n] - f0[n]).abs().max()) < 5e-5, (mode, n)
            assert torch.allclose(sf0[n], sf1[n], rtol=0, atol=1e-6), (mode, n)       # after consolidate_parameters()


# ------------------------------------------------------------------------------------------------ RCCL on one GPU
def _worker_nccl(port, q):
    """World-size-1 process group on backend "nccl" (= RCCL): the reducer is forced active, so every collective of the N > 1 path runs
    against the real library -- the branch `bench.py --gpus N` takes (main_task_retrieval.py:23,197-198) -- in both forms: through
    a communicator of our own, captured into the step's hipGraph (default), and through torch's process group (ReduceOp.AVG,
    all_gather_into_tensor, thread-local capture mode, segmented hipGraphs)."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist
        torch.cuda.set_device(0)
        from univl_amd import BertAdam, clip_grad_norm_
        from univl_amd.graphed import GraphedTrainStep
        out = {}

        def train(dp, graphed, capture=True):
            model, args, kw = _model_and_batch(0, ROWS)
            if dp:
                model.dp_capture = capture            # the operational switch a training process has
                model.enable_data_parallel(force=True)
                assert model._reducer is not None and model._reducer._avg and model._reducer.world == 1
                assert model._reducer.capturable == capture
            opt = BertAdam(model.parameters(), lr=1e-4, warmup=-1, t_total=-1, weight_decay=0.01, max_grad_norm=1.0)
            losses = []
            if graphed:
                gs = GraphedTrainStep(model, opt, max_grad_norm=1.0, warmup=1, pipeline_optimizer=True)
                for _ in range(STEPS + 1):
                    losses.append(float(gs(*args, **kw)))
                gs.flush()
                mode = gs.mode
            else:
                for _ in range(STEPS + 1):
                    loss = model(*args, **kw)
                    loss.backward()
                    clip_grad_norm_(model.parameters(), 1.0)
                    opt.step()
                    opt.zero_grad()
                    losses.append(float(loss))
                mode = "eager"
            red = model._reducer
            names = _probe_names(model)
            final = {n: model.flat.w32(n).detach().float().cpu().numpy().copy() for n in names}
            return dict(losses=losses, final=final, mode=mode, calls=0 if red is None else red.calls,
                        bytes=0 if red is None else red.bytes_reduced)

        out["ref"] = train(False, False)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        out["backend"] = dist.get_backend()
        out["eager"] = train(True, False)
        out["graph"] = train(True, True)
        out["eager_pg"] = train(True, False, capture=False)
        out["graph_pg"] = train(True, True, capture=False)
        torch.cuda.synchronize()
        dist.destroy_process_group()
        q.put(("ok", out))
    except Exception as ex:      # noqa: BLE001
        import traceback
        q.put(("error", "%s: %s\n%s" % (type(ex).__name__, ex, traceback.format_exc())))


def test_reducer_on_rccl_world_size_one():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_nccl, args=(_free_port(), q))
    p.start()
    status, out = q.get(timeout=900)
    p.join(timeout=120)
    assert status == "ok", out
    assert out["backend"] == "nccl"
    ref = out["ref"]
    for kind in ("eager", "graph", "eager_pg", "graph_pg"):
        r = out[kind]
        assert r["calls"] > 0 and r["bytes"] > 0, kind                 # collectives really went through RCCL
        assert max(abs(a - b) for a, b in zip(r["losses"], ref["losses"])) < 2e-4, (kind, r["losses"], ref["losses"])
        for n in ref["final"]:
            assert float(abs(r["final"][n] - ref["final"][n]).max()) < 5e-5, (kind, n)
    # a library-held RCCL communicator (univl_amd.rccl): the exchange is part of the plan and the whole data-parallel iteration is ONE
    # hipGraph; through torch's process group (dp_capture=0): captured segments with host-issued collectives between them
    assert out["graph"]["mode"] == "whole" and out["graph_pg"]["mode"] == "segmented"


# ------------------------------------------------------------------------------------------------ riding update + captured exchange
def _worker_ride(port, q):
    """bf16 compute, deterministic mode, RCCL world size 1 (forced-active reducer): the captured data-parallel iteration WITH the
    riding BertAdam update -- two graphs, forward with riders | backward with collectives + clip (graphed.GraphedTrainStep) -- against
    the single-process captured step; the same with async_loss; and UNIVL_GRAD_EXCHANGE=bf16 (captured: cast, all-reduce of half
    the bytes, cast back) with its own gate on the exchanged gradients."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ["UNIVL_DETERMINISTIC"] = "1"
        import torch.distributed as dist
        torch.cuda.set_device(0)
        _setup_path()
        import univl_oracle as O
        from make_golden import case_config
        from test_model_gpu import build
        from univl_amd import BertAdam
        from univl_amd.graphed import GraphedTrainStep
        cfg, _, dseed = case_config(CASE)
        full = O.synthetic_batch(cfg, ROWS, seed=dseed)
        b = {k: v.to("cuda") for k, v in full.items()}
        args = (b["input_ids"], b["token_type_ids"], b["attention_mask"], b["video"], b["video_mask"])
        kw = dict(pairs_masked_text=b["pairs_masked_text"], pairs_token_labels=b["pairs_token_labels"],
                  masked_video=b["masked_video"], video_labels_index=b["video_labels_index"])

        def train(dp, async_loss=False, exchange="fp32"):
            os.environ["UNIVL_GRAD_EXCHANGE"] = exchange
            model, _ = build(cfg, torch.bfloat16)
            model.train()
            if dp:
                model.enable_data_parallel(force=True)
                assert model._reducer is not None and model._reducer.capturable and model._reducer.bf16 == (exchange == "bf16")
            opt = BertAdam(model.parameters(), lr=1e-4, warmup=-1, t_total=-1, weight_decay=0.01, max_grad_norm=1.0)
            gs = GraphedTrainStep(model, opt, max_grad_norm=1.0, warmup=1, pipeline_optimizer=True, async_loss=async_loss)
            losses = [float(gs(*args, **kw)) for _ in range(STEPS + 2)]
            gs.flush()
            names = _probe_names(model)
            final = {n: model.flat.w32(n).detach().float().cpu().numpy().copy() for n in names}
            return dict(losses=losses, final=final, mode=gs.mode, ride=gs.ride, two=gs._g_rest is not None,
                        calls=0 if model._reducer is None else model._reducer.calls)

        def grads(dp, exchange):
            os.environ["UNIVL_GRAD_EXCHANGE"] = exchange
            model, _ = build(cfg, torch.bfloat16)
            model.train()
            if dp:
                model.enable_data_parallel(force=True)
            model(*args, **kw).backward()
            torch.cuda.synchronize()
            return model.flat.g32.detach().double().cpu()

        out = {"ref": train(False)}
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        out["dp"] = train(True)
        out["dp_async"] = train(True, async_loss=True)
        out["dp_bf16"] = train(True, exchange="bf16")
        g32, g16 = grads(True, "fp32"), grads(True, "bf16")
        out["bf16_global"] = float((g16 - g32).norm() / g32.norm())
        nz = g32.abs() > 1e-12
        out["bf16_elem"] = float(((g16 - g32).abs()[nz] / g32.abs()[nz]).max())
        os.environ["UNIVL_GRAD_EXCHANGE"] = "fp32"
        torch.cuda.synchronize()
        dist.destroy_process_group()
        q.put(("ok", out))
    except Exception as ex:      # noqa: BLE001
        import traceback
        q.put(("error", "%s: %s\n%s" % (type(ex).__name__, ex, traceback.format_exc())))


def test_riding_update_with_captured_exchange_and_bf16_exchange_gate():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_ride, args=(_free_port(), q))
    p.start()
    status, out = q.get(timeout=900)
    p.join(timeout=120)
    assert status == "ok", out
    ref = out["ref"]
    assert ref["ride"] and ref["mode"] == "whole" and not ref["two"]
    for kind in ("dp", "dp_async", "dp_bf16"):
        r = out[kind]
        # the riding update stays on under a captured exchange, as two graphs (forward with riders | backward with collectives)
        assert r["ride"] and r["mode"] == "whole" and r["two"] and r["calls"] > 0, (kind, r["ride"], r["mode"], r["two"])
        tol_l, tol_p = (2e-3, 2e-4) if kind == "dp_bf16" else (2e-4, 5e-5)
        assert max(abs(a - b) for a, b in zip(r["losses"], ref["losses"])) < tol_l * max(1.0, abs(ref["losses"][0])), (kind, r["losses"], ref["losses"])
        for n in ref["final"]:
            assert float(abs(r["final"][n] - ref["final"][n]).max()) < tol_p, (kind, n)
    # UNIVL_GRAD_EXCHANGE=bf16: every exchanged element is the fp32 gradient rounded to 8 mantissa bits (world size 1: the mean of
    # one rank): <= 2^-9 per element, ~2^-9 / sqrt(3) in the global norm
    assert out["bf16_elem"] <= 2.0 ** -8 and out["bf16_global"] <= 3e-3, (out["bf16_elem"], out["bf16_global"])


def _worker_cabi(q):
    """univl_allreduce_bucket (include/univl_hip.h) with a communicator the HOST created through RCCL's C API."""
    try:
        import ctypes as C
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        _setup_path()
        from univl_amd import _lib
        cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so.1", "librccl.so.1"]
        for c in cands:
            try:
                C.CDLL(c, mode=C.RTLD_GLOBAL)
                break
            except OSError:
                continue
        G = C.CDLL(None)                      # the process-global namespace: what dlsym(RTLD_DEFAULT) inside the library sees

        class Uid(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        uid, comm = Uid(), C.c_void_p()
        G.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
        assert G.ncclGetUniqueId(C.byref(uid)) == 0
        assert G.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
        L = _lib.lib()
        assert L.univl_init(0) == 0
        x = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        for avg in (0, 1):
            _lib.check(L.univl_allreduce_bucket(x.data_ptr(), x.numel(), _lib.DT_F32, avg, comm, C.c_void_p(side.cuda_stream)), "allreduce")
        side.synchronize()
        ok = bool(torch.equal(x.cpu(), torch.arange(1 << 20, dtype=torch.float32)))
        G.ncclCommDestroy.argtypes = [C.c_void_p]
        G.ncclCommDestroy(comm)
        L.univl_destroy()
        q.put(("ok", ok))
    except Exception as ex:      # noqa: BLE001
        import traceback
        q.put(("error", "%s: %s\n%s" % (type(ex).__name__, ex, traceback.format_exc())))


def test_allreduce_bucket_c_abi_with_host_owned_rccl_communicator():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_cabi, args=(q,))
    p.start()
    status, out = q.get(timeout=600)
    p.join(timeout=60)
    assert status == "ok", out
    assert out is True
