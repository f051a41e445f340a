"""Data-parallel gradient exchange for the UniVL hot path.

The reference wraps the model in torch DDP over NCCL (main_task_retrieval.py:197-198): bucketed all-reduce(SUM)/world
of every gradient during backward, plus a per-iteration unused-parameter bitmap all-reduce
(find_unused_parameters=True).  Here:
  * gradients already live in ONE flat fp32 buffer laid out by layer, so a bucket is a contiguous slice -- no
    flatten/unflatten copies, and the set of gradient-less parameters is static (the two dead poolers), so the
    bitmap exchange disappears;
  * one process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm).  Each layer's slice is
    all-reduced (AVG) as soon as that layer's last wgrad kernel has been enqueued; RCCL runs it on its own HIP
    stream behind an event, overlapping the rest of the backward; the step joins all buckets before the clip.
  * works unchanged with backend "gloo" on CPU tensors (world_size-2 tests in tests/test_parallel_cpu.py).
"""
import torch
import torch.distributed as dist

from . import _ab


class BucketReducer:
    """All-reduce (mean) contiguous slices of a flat gradient buffer, asynchronously, in a fixed order.

    loopback=True is a single-process stand-in used by the single-GPU tests and `bench.py --loopback`: every "exchange"
    is an identity pass over the slice on a private communication stream, with the same event choreography as the
    RCCL path (wait for the producer stream, run asynchronously, join before the optimizer) -- so that the bucket
    schedule, the exchange points and the hipGraph segmentation around them run without a second GPU."""

    def __init__(self, flat_grad, process_group=None, loopback=False, force=False):
        """force=True keeps a world-size-1 process group active: every collective of the N > 1 path is issued for real
        (RCCL on one GPU), which is how the single-GPU box exercises the nccl branch (tests/test_ddp_gpu.py)."""
        self.g = flat_grad
        self.pg = process_group
        self.loopback = bool(loopback)
        self.force = bool(force)
        # UNIVL_GRAD_EXCHANGE=bf16: all-reduce a bf16 copy of each slice (half the bytes over xGMI; every gradient element is
        # rounded to 8 mantissa bits before the mean, i.e. +2^-9 relative noise per element on top of the compute noise -- NOT
        # the reference's fp32 DDP semantics, hence opt-in)
        import os
        self.bf16 = os.environ.get("UNIVL_GRAD_EXCHANGE", "fp32").lower() == "bf16"
        self._g16 = None
        self.world = dist.get_world_size(process_group) if (dist.is_initialized() and not loopback) else 1
        self.rank = dist.get_rank(process_group) if (dist.is_initialized() and not loopback) else 0
        self.partition, self.owned = None, None
        self.pending = []
        self.bytes_reduced = 0
        self.calls = 0
        backend = dist.get_backend(process_group) if (dist.is_initialized() and not loopback) else "none"
        self._avg = backend == "nccl"
        self._comm = None
        # enable_capture(): collectives go through a library-held RCCL communicator on a stream of ours (univl_amd.rccl) -- plain
        # kernel enqueues that a hipGraph capture records like any other node -- instead of torch's ProcessGroupNCCL
        self.rccl = None
        self.capturable = False
        self._cstream = None
        self._inflight = False
        # measure=True (bench.py, eager iterations only -- events cannot be timed inside a graph): HIP events around every
        # collective on the communication stream and around the join on the compute stream
        # UNIVL_DP_DRYRUN=1 (measurement only, bench.py --force-dp on one GPU): every stream fork / join, exchange point and
        # bookkeeping of the data-parallel schedule, but the collective itself is not enqueued -- separates what the SCHEDULE costs
        # from what RCCL's kernels cost (at world size 1 they still stream every bucket through HBM)
        self.dryrun = bool(_ab.get("dp_dryrun"))
        self.measure = False
        self.timings = dict(collective_ms=0.0, exposed_ms=0.0, bytes=0, steps=0)
        self._ev = []

    def _agree(self, ok):
        """MIN over the ranks of a local 0 / 1 outcome, through the torch process group (never through the communicator under test)."""
        if self.world > 1:
            flag = torch.tensor([1 if ok else 0], device=self.g.device, dtype=torch.int32)      # (the group's device: cuda for RCCL)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
            ok = int(flag)
        return bool(ok)

    def enable_capture(self):
        """RCCL backend only.  Collective over the group (creates the communicator).  Every step whose failure could be LOCAL to
        one rank (librccl not loadable, ncclGetUniqueId, the self-test) ends in an agreement over the process group before any rank
        acts on it, so that either every rank takes the captured path or every rank raises -- a rank that fell back alone would leave
        the others hanging in the next collective of the path it left."""
        from . import rccl as _r
        if not self._avg or self.loopback or self.partition is not None:
            return False
        err = None
        try:
            _r._rccl()
        except Exception as ex:      # noqa: BLE001
            err = ex
        if not self._agree(err is None):
            raise RuntimeError("librccl is not loadable on at least one rank (%s)" % (err,))
        # RcclComm: rank 0's unique id (or None, if ncclGetUniqueId failed there) is broadcast to everybody -- a consistent outcome --
        # and ncclCommInitRank is RCCL's own collective; an exception that only ONE rank sees (ncclCommInitRank's return code, a missing
        # symbol) still goes through the agreement: every rank falls back together
        err = None
        try:
            self.rccl = _r.RcclComm(self.pg)
        except Exception as ex:      # noqa: BLE001
            import sys
            print("[univl_amd] rank %d: RCCL communicator construction failed (%s: %s)" % (self.rank, type(ex).__name__, ex), file=sys.stderr)
            err = ex
        if not self._agree(err is None):
            if self.rccl is not None:
                self.disable_capture()
            raise RuntimeError("RCCL communicator construction failed on at least one rank")
        self._cstream = torch.cuda.Stream()
        err = None
        try:
            self._capture_selftest()
        except Exception as ex:      # noqa: BLE001 -- an RCCL build that refuses stream capture reports it here, before any training step
            import sys
            print("[univl_amd] rank %d: captured RCCL all-reduce self-test failed (%s: %s)" % (self.rank, type(ex).__name__, ex), file=sys.stderr)
            err = ex
        if not self._agree(err is None):
            self.disable_capture()
            raise RuntimeError("captured RCCL all-reduce self-test failed on at least one rank")
        self.capturable = True
        return True

    def disable_capture(self):
        """Back to the process-group form; the library-held communicator and its stream are released."""
        self.capturable = False
        if self.rccl is not None:
            try:
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                self.rccl.destroy()
            finally:
                self.rccl, self._cstream, self._inflight = None, None, False

    def _capture_selftest(self):
        """A small all-reduce through the new communicator: once eagerly (connection set-up happens outside any capture), once
        captured into a hipGraph and replayed twice -- on every rank, with the result checked."""
        s = self._cstream
        want = float(self.world * (self.world + 1) // 2)
        x = torch.empty(4096, device="cuda", dtype=torch.float32)

        def check(tag):
            torch.cuda.synchronize()
            if not bool((x == want).all()):
                raise RuntimeError("%s all-reduce returned %r, expected %r" % (tag, float(x[0]), want))

        x.fill_(float(self.rank + 1))
        s.wait_stream(torch.cuda.current_stream())
        self.rccl.all_reduce(x, False, s)
        check("eager")
        g = torch.cuda.CUDAGraph()
        src = torch.full_like(x, float(self.rank + 1))
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            cur = torch.cuda.current_stream()
            x.copy_(src)
            s.wait_stream(cur)
            self.rccl.all_reduce(x, False, s)
            cur.wait_stream(s)
        for _ in range(2):
            x.zero_()
            g.replay()
            check("captured")

    def _fork(self, after=None):
        """The communication stream picks up behind everything enqueued so far on the current stream (and on the streams in
        `after`: the plan's side streams -- the video stack's weight gradients are produced there)."""
        for st in (after or [torch.cuda.current_stream()]):
            self._cstream.wait_stream(st)
        self._inflight = True

    @property
    def active(self):
        return self.world > 1 or self.loopback or (self.force and dist.is_initialized())

    def reduce_slice(self, start, end, after=None):
        """Launch the all-reduce of g[start:end]; returns immediately (the collective is stream-/thread-async)."""
        if not self.active or end <= start:
            return
        t = self.g[start:end]
        self.calls += 1
        self.bytes_reduced += t.numel() * t.element_size()
        if self.loopback:
            if t.is_cuda:
                if self._comm is None:
                    self._comm = torch.cuda.Stream(device=t.device)
                cur = torch.cuda.current_stream()
                self._comm.wait_stream(cur)
                with torch.cuda.stream(self._comm):
                    t.mul_(1.0)
                    ev = torch.cuda.Event()
                    ev.record()
                self.pending.append((ev, t, None))
            return
        if self.capturable:
            self._fork(after)
            timed = self.measure and not torch.cuda.is_current_stream_capturing()
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self._cstream)
            if self.bf16:
                # UNIVL_GRAD_EXCHANGE=bf16 on the communication stream: round the bucket to bf16, all-reduce HALF the bytes, widen the
                # mean back into the fp32 gradient buffer -- three enqueues, all capturable
                from . import ops
                if self._g16 is None:
                    self._g16 = torch.empty_like(self.g, dtype=torch.bfloat16)
                t16 = self._g16[start:end]
                with torch.cuda.stream(self._cstream):
                    ops.cast_bf16(t, t16)
                if not self.dryrun:
                    self.rccl.all_reduce(t16, True, self._cstream)
                with torch.cuda.stream(self._cstream):
                    ops.cast_f32(t16, t)
            elif not self.dryrun:
                self.rccl.all_reduce(t, True, self._cstream)
            if timed:
                e1.record(self._cstream)
                self._ev.append(("c", e0, e1, t.numel() * t.element_size()))
            return
        if self.bf16:
            if self._g16 is None:
                self._g16 = torch.empty_like(self.g, dtype=torch.bfloat16)
            t16 = self._g16[start:end]
            t16.copy_(t)                                   # on the producer stream, behind the slice's last wgrad
            w = dist.all_reduce(t16, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self.pending.append((w, t, t16))
            return
        if self._avg:
            w = dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
        else:
            w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self.pending.append((w, t, None))

    def gather(self, t, out, after=None):
        """out[r] <- rank r's t (all-gather), asynchronously like reduce_slice; out: [world, *t.shape]."""
        self.calls += 1
        self.bytes_reduced += out.numel() * out.element_size()
        if self.loopback:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=t.device)
            self._comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm):
                out[0].copy_(t)
                ev = torch.cuda.Event()
                ev.record()
            self.pending.append((ev, None, None))
            return
        if self.capturable:
            self._fork(after)
            if not self.dryrun:
                self.rccl.all_gather(t, out, self._cstream)
            else:
                with torch.cuda.stream(self._cstream):
                    out[self.rank].copy_(t)
            return
        if self._avg:          # RCCL: one contiguous output, no per-rank staging copies
            w = dist.all_gather_into_tensor(out.view(-1, *t.shape[1:]), t, group=self.pg, async_op=True)
        else:
            w = dist.all_gather([out[r] for r in range(self.world)], t, group=self.pg, async_op=True)
        self.pending.append((w, None, None))

    def reduce_ranges(self, ranges, after=None):
        if self.partition is not None:
            return self.reduce_scatter_ranges(ranges)
        for s0, e0 in ranges:
            self.reduce_slice(s0, e0, after)

    # ------------------------------------------------------------------------------- sharded optimizer (ZeRO-1 style)
    # With `partition` set (shard_partition below) every exchanged range is REDUCE-SCATTERED instead of all-reduced: rank r
    # ends up with the mean gradient of its 1/world sub-range of every partition range only, runs clip + BertAdam on that
    # shard (optimization._Tables(..., owned=...)), and the updated bf16 shadow (what the next forward reads) is
    # ALL-GATHERED.  Bytes over xGMI per step: (W-1)/W x (4 + 2) B/param instead of 2 (W-1)/W x 4 B/param for the
    # all-reduce, and the 30 B/param optimizer stream shrinks by the world size.
    def set_partition(self, partition):
        if partition is not None:
            self.disable_capture()           # the sharded path issues host-synchronous collectives (norm all-reduce): segmented graphs
        self.partition = list(partition) if partition is not None else None
        self.owned = owned_ranges(self.partition, self.world, self.rank) if partition is not None else None

    def _pieces(self, s0, e0):
        """Partition ranges inside [s0, e0) -- exchange ranges are unions of whole partition ranges."""
        out = [(a, b) for a, b in self.partition if a >= s0 and b <= e0]
        assert sum(b - a for a, b in out) == e0 - s0, "exchange range %r is not a union of partition ranges" % ((s0, e0),)
        return out

    def reduce_scatter_ranges(self, ranges):
        for s0, e0 in ranges:
            for a, b in self._pieces(s0, e0):
                self._reduce_scatter(self.g, a, b)

    def _reduce_scatter(self, buf, a, b):
        if not self.active:
            return
        n = (b - a) // self.world
        t = buf[a:b]
        self.calls += 1
        self.bytes_reduced += t.numel() * t.element_size()
        if self.loopback:
            return self.reduce_slice(a, b)
        if self._avg:      # RCCL: in place (the output is the rank's own piece of the input)
            w = dist.reduce_scatter_tensor(buf[a + self.rank * n:a + (self.rank + 1) * n], t, op=dist.ReduceOp.AVG,
                                           group=self.pg, async_op=True)
            self.pending.append((w, None, None))
        else:              # gloo has no reduce-scatter: all-reduce the range, every rank keeps using only its piece
            w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self.pending.append((w, t, None))

    def all_gather_ranges(self, buf, ranges=None):
        """buf[a:b] <- concatenation over ranks of each rank's piece of [a, b), for every partition range (in place)."""
        if not self.active or self.loopback:
            return
        for a, b in (self.partition if ranges is None else ranges):
            n = (b - a) // self.world
            mine = buf[a + self.rank * n:a + (self.rank + 1) * n]
            self.calls += 1
            self.bytes_reduced += (b - a) * buf.element_size()
            if self._avg:
                w = dist.all_gather_into_tensor(buf[a:b], mine, group=self.pg, async_op=True)
            else:
                w = dist.all_gather([buf[a + r * n:a + (r + 1) * n] for r in range(self.world)], mine.clone(), group=self.pg,
                                    async_op=True)
            self.pending.append((w, None, None))

    def all_reduce_small(self, t):
        """Synchronous SUM of a small tensor (per-tensor gradient sums of squares of the shards)."""
        if self.active and not self.loopback:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)

    def join(self):
        """Make the current stream (or thread, for gloo) wait for every outstanding bucket."""
        if self._inflight:
            cur = torch.cuda.current_stream()
            timed = self.measure and not torch.cuda.is_current_stream_capturing()
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
            cur.wait_stream(self._cstream)
            if timed:
                e1.record(cur)
                self._ev.append(("j", e0, e1, 0))
            self._inflight = False
        for w, t, t16 in self.pending:
            if self.loopback:
                torch.cuda.current_stream().wait_event(w)
                continue
            w.wait()
            if t16 is not None:
                t.copy_(t16)                               # back to the fp32 gradient the clip / optimizer read
            if not self._avg and t is not None:
                t.div_(self.world)
        self.pending = []


def collect_timings(red):
    """Fold the events of BucketReducer(measure=True) into red.timings (synchronises the device)."""
    torch.cuda.synchronize()
    for kind, e0, e1, nbytes in red._ev:
        ms = e0.elapsed_time(e1)
        if kind == "c":
            red.timings["collective_ms"] += ms
            red.timings["bytes"] += nbytes
        else:
            red.timings["exposed_ms"] += ms
            red.timings["steps"] += 1
    red._ev = []
    return red.timings


def merge_ranges(ranges):
    """Sort [start, end) ranges and fuse the ones that touch or overlap."""
    out = []
    for s0, e0 in sorted(ranges):
        if e0 <= s0:
            continue
        if out and s0 <= out[-1][1]:
            out[-1] = (out[-1][0], max(out[-1][1], e0))
        else:
            out.append((s0, e0))
    return out


class BucketSchedule:
    """Where the gradient exchange points go.  Layers report their gradient slice as soon as their last weight-gradient
    kernel has been planned (backward order); once at least `min_bytes` are pending, the caller emits one exchange
    point for everything pending (adjacent layers fuse into one contiguous all-reduce).  Few, large exchanges: an
    xGMI ring all-reduce is bandwidth-bound per link, and under hipGraph replay every exchange point ends a captured
    segment (univl_amd.engine.Plan.run_graphed)."""

    def __init__(self, min_bytes, elem_bytes=4):
        self.min_bytes, self.elem = int(min_bytes), elem_bytes
        self.pending, self.pending_bytes = [], 0
        self.cuts = []

    def add(self, start, end):
        """Returns True when an exchange point is due."""
        if end > start:
            self.pending.append((start, end))
            self.pending_bytes += (end - start) * self.elem
        return self.pending_bytes >= self.min_bytes

    def take(self):
        r = merge_ranges(self.pending)
        self.pending, self.pending_bytes = [], 0
        if r:
            self.cuts.append(r)
        return r


def subtract_range(ranges, lo, hi):
    """ranges minus [lo, hi)."""
    out = []
    for s0, e0 in ranges:
        if e0 <= lo or s0 >= hi:
            out.append((s0, e0))
            continue
        if s0 < lo:
            out.append((s0, lo))
        if e0 > hi:
            out.append((hi, e0))
    return out


def layer_buckets(flat, used_names):
    """Bucket plan over FlatParams: one bucket per encoder layer (its four weight matrices are contiguous,
    ~28 MB fp32 -- the size of a default DDP bucket), one for the remaining matrices and one for the whole
    atomic/vector region (embedding tables + all biases / LayerNorm parameters).  Returns
    {"layers": {(prefix, l): (start, end)}, "tail": [(start, end), ...]}; slices never overlap and cover every
    used parameter exactly once."""
    layers, covered = {}, []
    for n in used_names:
        parts = n.split(".")
        if len(parts) > 3 and parts[1] == "encoder" and parts[2] == "layer" and len(flat.index[n][2]) == 2:
            key = (parts[0], int(parts[3]))
            o, k, _ = flat.index[n]
            end = o + (k + 63) // 64 * 64
            if key in layers:
                s, e = layers[key]
                layers[key] = (min(s, o), max(e, end))
            else:
                layers[key] = (o, end)
    for s, e in layers.values():
        covered.append((s, e))
    covered.sort()
    tail = [(0, flat.v_end)]
    cur = flat.v_end
    for s, e in covered:
        if s > cur:
            tail.append((cur, s))
        cur = max(cur, e)
    if cur < flat.total:
        tail.append((cur, flat.total))
    return dict(layers=layers, tail=tail)


def broadcast_parameters(flat_p32, src=0, process_group=None):
    """DDP's constructor broadcast: one collective over the flat buffer."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.broadcast(flat_p32, src=src, group=process_group)


def shard_partition(flat, world):
    """Static partition of the flat buffer for the sharded optimizer: every encoder-layer bucket, the word-embedding table,
    and the remaining stretches of the atomic region / the other matrices -- disjoint ranges covering [0, total), each a
    multiple of `world` x 64 elements long... (64 | every range by construction; world must divide 64).  Rank r owns the
    r-th of `world` equal pieces of every range: a reduce-scatter of any exchange range (a union of these ranges) leaves
    exactly the owned pieces valid."""
    if 64 % world != 0:
        raise ValueError("sharded optimizer: world size %d must divide 64" % world)
    b = layer_buckets(flat, flat.order)
    ranges = list(b["layers"].values())
    wo, wk, _ = flat.index[flat.WORD] if flat.WORD in flat.index else (0, 0, None)
    wend = wo + (wk + 63) // 64 * 64
    for s0, e0 in b["tail"]:
        for a, c in subtract_range([(s0, e0)], wo, wend):
            ranges.append((a, c))
    if wk:
        ranges.append((wo, wend))
    ranges = sorted(r for r in ranges if r[1] > r[0])
    assert ranges[0][0] == 0 and ranges[-1][1] == flat.total and all(x[1] == y[0] for x, y in zip(ranges, ranges[1:]))
    return ranges


def owned_ranges(partition, world, rank):
    return [(a + rank * ((b - a) // world), a + (rank + 1) * ((b - a) // world)) for a, b in partition]
