"""world_size-2 gloo test of the data-parallel gradient exchange (univl_amd.parallel.BucketReducer): the same
bucket schedule the GPU path runs over RCCL, on CPU tensors."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from univl_amd.parallel import BucketReducer, BucketSchedule, merge_ranges, subtract_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    red = BucketReducer(g)
    # layer buckets in backward order, then the tail -- every element exactly once
    sched = BucketSchedule(min_bytes=2000)
    issued = []
    for s, e in ((600, 900), (300, 600), (0, 300), (900, 1000)):
        if sched.add(s, e):
            issued.append(sched.take())
            red.reduce_ranges(issued[-1])
    issued.append(sched.take())
    red.reduce_ranges(issued[-1])
    red.join()
    assert issued == [[(300, 900)], [(0, 300), (900, 1000)]]
    # token exchange of the sparse word-embedding gradient: every rank receives every rank's (ids, rows)
    ids_all = torch.zeros(world, 3, dtype=torch.int64)
    rows_all = torch.zeros(world, 3, 4)
    red.gather(torch.arange(3) + 10 * rank, ids_all)
    red.gather(torch.full((3, 4), float(rank + 1)), rows_all)
    red.join()
    assert torch.equal(ids_all, torch.stack([torch.arange(3) + 10 * r for r in range(world)]))
    assert torch.equal(rows_all, torch.stack([torch.full((3, 4), float(r + 1)) for r in range(world)]))
    expect = torch.arange(1000, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    ok = torch.allclose(g, expect) and red.bytes_reduced == 4000 + ids_all.numel() * 8 + rows_all.numel() * 4 and not red.pending
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_bucket_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_bucket_reducer_single_process_is_noop():
    g = torch.ones(10)
    red = BucketReducer(g)
    red.reduce_slice(0, 10)
    red.join()
    assert torch.equal(g, torch.ones(10)) and red.bytes_reduced == 0


def test_bucket_schedule_and_merge():
    assert merge_ranges([(10, 20), (0, 10), (30, 40), (35, 50), (5, 5)]) == [(0, 20), (30, 50)]
    s = BucketSchedule(min_bytes=100, elem_bytes=4)
    assert not s.add(0, 10) and s.add(10, 30)
    assert s.take() == [(0, 30)] and s.take() == [] and s.cuts == [[(0, 30)]]
    assert s.add(100, 200) and s.take() == [(100, 200)]


def test_subtract_range():
    r = [(0, 100), (200, 300)]
    assert subtract_range(r, 20, 50) == [(0, 20), (50, 100), (200, 300)]
    assert subtract_range(r, 0, 100) == [(200, 300)]
    assert subtract_range(r, 90, 250) == [(0, 90), (250, 300)]
    assert subtract_range(r, 100, 200) == r


# ------------------------------------------------------------------------------------------ sharded optimizer exchange
class _FakeFlat:
    """The part of engine.FlatParams that parallel.shard_partition reads."""
    WORD = "bert.embeddings.word_embeddings.weight"

    def __init__(self):
        names = [("bert.embeddings.word_embeddings.weight", (40, 16)), ("bert.embeddings.LayerNorm.weight", (16,)),
                 ("bert.encoder.layer.0.attention.self.query.bias", (16,))]
        mats = [("bert.encoder.layer.%d.%s.weight" % (l, k), (16, 16)) for l in range(2) for k in ("a", "b")]
        mats += [("visual.embeddings.word_embeddings.weight", (16, 32)), ("visual.encoder.layer.0.a.weight", (16, 16))]
        self.index, self.order, off = {}, [], 0
        for n, shp in names + mats:
            k = 1
            for d in shp:
                k *= d
            if n == mats[0][0]:
                self.v_end = off
            self.index[n] = (off, k, shp)
            self.order.append(n)
            off += (k + 63) // 64 * 64
        self.total = off


def _worker_shard(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from univl_amd.parallel import owned_ranges, shard_partition
    fl = _FakeFlat()
    part = shard_partition(fl, world)
    g = (torch.arange(fl.total, dtype=torch.float32) + 1) * (rank + 1)
    red = BucketReducer(g)
    red.set_partition(part)
    own = owned_ranges(part, world, rank)
    assert all((b - a) % world == 0 for a, b in part) and sum(b - a for a, b in own) * world == fl.total
    # exchange points as the backward plan issues them: unions of whole partition ranges, every range exactly once
    mid = part[len(part) // 2][0]
    red.reduce_ranges([(mid, fl.total)])
    red.reduce_ranges([(0, mid)])
    red.join()
    mean = (torch.arange(fl.total, dtype=torch.float32) + 1) * (sum(range(1, world + 1)) / world)
    for a, b in own:                                  # the owned pieces hold the mean gradient
        assert torch.allclose(g[a:b], mean[a:b])
    # "optimizer": every rank updates its own pieces of the parameters, then the all-gather completes them everywhere
    p = torch.zeros(fl.total)
    for a, b in own:
        p[a:b] = -0.5 * g[a:b]
    red.all_gather_ranges(p)
    red.join()
    ok = torch.allclose(p, -0.5 * mean)
    small = torch.tensor([float(rank + 1), 2.0])
    red.all_reduce_small(small)
    ok = ok and small.tolist() == [float(sum(range(1, world + 1))), 2.0 * world]
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_sharded_exchange_gloo_world2_and_world4():
    for world in (2, 4):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker_shard, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=180) for _ in procs)
        for p in procs:
            p.join(timeout=60)
        assert res == [(r, True) for r in range(world)], (world, res)


def _worker_capture_fallback(rank, world, port, q):
    """enable_capture() with a communicator whose construction fails on ONE rank (mocked univl_amd.rccl): both ranks must raise, together,
    and the process-group exchange must keep working on both -- a rank that fell back alone would leave the other one hanging in the
    first collective of the path it left (parallel.BucketReducer._agree)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import univl_amd.rccl as R

    class FailingOnRank1:
        def __init__(self, pg):
            if dist.get_rank() == 1:
                raise OSError("ncclCommInitRank returned 2 (mock)")
            self.destroyed = False

        def destroy(self):
            self.destroyed = True

    R._rccl = lambda: None
    R.RcclComm = FailingOnRank1
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    red = BucketReducer(g)
    red._avg = True                       # what an "nccl" group sets: enable_capture() only runs for RCCL groups
    msg = None
    try:
        red.enable_capture()
    except RuntimeError as ex:
        msg = str(ex)
    red._avg = False
    ok = msg is not None and "at least one rank" in msg and red.rccl is None and not red.capturable and red._cstream is None
    red.reduce_ranges([(0, 1000)])        # the host-issued exchange through the process group, on both ranks
    red.join()
    ok = ok and torch.allclose(g, torch.arange(1000, dtype=torch.float32) * (sum(range(1, world + 1)) / world))
    q.put((rank, bool(ok), msg))
    dist.destroy_process_group()


def test_capture_failure_on_one_rank_falls_back_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_capture_fallback, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[:2] for r in res] == [(0, True), (1, True)], res
