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
