"""The training iteration of main_task_retrieval.py:333-352 (forward, backward, gradient exchange, clip, BertAdam,
zero_grad) replayed as hipGraphs.

At 4-16 pairs per GPU a step is ~380 kernels of 2-10 us each; launched one by one from Python the host is the
bottleneck (~10 us per ctypes call).  `GraphedTrainStep` captures the iteration once and replays it:

  * one process, no gradient exchange: ONE graph holds the whole iteration;
  * data parallel over RCCL (UniVL.enable_data_parallel; parallel.BucketReducer.enable_capture): the collectives go through a
    communicator of our own on a communication stream (univl_amd.rccl), so they are nodes of the SAME graph -- still ONE
    graph per iteration;
  * data parallel through torch's process group (gloo, the sharded optimizer, UNIVL_DP_CAPTURE=0): the collectives stay
    OUTSIDE the graphs -- the forward is one graph, the backward plan is cut at its gradient-exchange points into captured
    segments (engine.Plan.run_graphed) with the all-reduces issued from the host between the replays (they run on the
    process group's own stream and overlap the following segments), and clip + BertAdam are a last graph that is replayed
    after the reducer's join.

Pipelined optimizer (pipeline_optimizer=True).  Two forms: the RIDING form (default where it applies: bf16, one process -- the
pending update goes out as extra workgroups of the next forward's own GEMM launches, engine.Plan.add_gemm_rider; -8 % per step at 4
pairs, -10 % at 16, profiles/r03b_ab_adam_ride.txt) and the SIDE-STREAM form described next (fp32 / data parallel / UNIVL_ADAM_RIDE=0;
measured slower than no pipelining on MI355X, see the end of this paragraph).
At small per-GPU batch the forward/backward is a chain of
short latency-bound kernels that leaves most of the HBM bandwidth idle, while the BertAdam update is one long HBM-bound
stream (30 B per parameter) that needs nothing but bandwidth.  The global clip (main_task_retrieval.py:347) needs every
gradient before any parameter may change, so the update cannot move INTO its own backward -- but it can ride next to the
FOLLOWING forward: a replay applies the update of the previous iteration layer by layer on a second stream (embedding
tables and vectors first, then the layers in the order the forward reads them, univl_bert_adam_range) while the forward
runs, each layer's first kernel waiting only for that layer's parameters (engine.Plan.wait_point); backward, gradient
exchange and the clip measurement follow as before.  Every iteration still performs exactly one forward, one backward
and one BertAdam update, in the same arithmetic order: losses and parameters are the same as without pipelining
(tests/test_model_gpu.py).  Between calls the parameters lag by the one pending update; `flush()` -- called
automatically by state_dict(), eval(), the evaluation entry points and optimizer.state_dict() -- applies it.
Measured (profiles/README.md, round 2): 3.38 ms per step pipelined against 3.16 ms sequential at 4 pairs per GPU, 3.25 ms
with the update capped at 256 workgroups: the forward is a chain of kernels bound by memory ROUND TRIPS, and a concurrent
stream that saturates HBM multiplies exactly those latencies -- what the overlap hides of the update it loses again in a
slower forward.  Kept as a tested option, not the default.

Early loss read-back (async_loss=True; OFF by default -- measured 3.159-3.165 ms against 3.167 ms per step, within noise).  The reference's loop reads `float(loss)` every iteration
(main_task_retrieval.py:344).  On a single captured graph that read waits for the WHOLE iteration, and the next replay is
only launched afterwards: ~0.1 ms of idle GPU per iteration at 4 pairs per GPU.  The iteration is therefore captured as two
graphs -- forward, and backward + clip + BertAdam -- both launched back to back; the loss is copied to pinned host memory on
a copy stream right after the forward graph, and float(loss) waits for that copy only.  The host gets every iteration's loss
before it continues, exactly as before, while the GPU is already running the backward and finds the next iteration queued.

The first `warmup` calls run eagerly (they build the execution plans and the optimizer tables); the next call
captures and runs; later calls only copy the new batch into the static input buffers and replay.
"""
import os

import torch

from . import _ab
from .engine import no_gc
from .optimization import clip_grad_norm_
from .steps import stage_input


class GraphedTrainStep:
    def __init__(self, model, optimizer, max_grad_norm=1.0, warmup=3, persistent_inputs=False, pipeline_optimizer=False, async_loss=False):
        """persistent_inputs=False (default): the graphs read PRIVATE static copies of the batch; every call copies its arguments in
        (one copy kernel for device tensors) and never touches the caller's tensors.  persistent_inputs=True (opt-in, what bench.py
        uses for its HBM-resident batch): the device tensors of the first captured call BECOME the static input buffers -- later
        calls may pass the same tensors refilled in place (no copy at all), and any OTHER batch passed later is copied INTO them,
        i.e. the caller's first batch is overwritten; a caller that keeps or mutates that batch must not use this mode."""
        self.model, self.opt = model, optimizer
        model.auto_ride = False          # this object schedules the optimizer update itself (BertAdam.step() must not defer on its own)
        optimizer.flush()
        self.max_grad_norm = max_grad_norm
        self.warmup = int(warmup)
        self.persistent = bool(persistent_inputs)
        self.pipeline = bool(pipeline_optimizer) or bool(_ab.get("pipeline_opt"))
        self.adam_blocks = _ab.get("adam_blocks")      # grid cap of the overlapped update (0: none)
        self.async_loss = bool(async_loss) or bool(_ab.get("async_loss"))
        # The pending update goes out as extra workgroups of the next forward's own launches (engine.Plan.add_gemm_rider) instead of a
        # second stream with one graph edge per layer: bf16, one process.  UNIVL_ADAM_RIDE=0: the side-stream form.  Measured on one MI355X
        # (profiles/r03b_ab_adam_ride.txt): 2.53 vs 2.74 ms per step at 4 pairs, 3.79 vs 4.23 at 16; bit-identical parameters
        # (tests/test_model_gpu.py::test_adam_update_riding_with_the_next_forward_matches_eager).
        fl = model.flat
        # With a captured gradient exchange (library-held RCCL communicator) the iteration is captured as TWO graphs -- forward with its
        # riders | backward with its collectives + clip -- launched back to back: rider launches and RCCL's kernel nodes in ONE graph
        # pass the world-size-1 test on a small model but crash hipGraphLaunch at the benchmark shape (profiles/r03g_dp_ride_crash.txt:
        # segmentation fault inside the replay; cause open).  One extra graph launch per iteration keeps the riding update in the
        # data-parallel step.  UNIVL_ADAM_RIDE=force: one graph anyway (the crashing form, for reproducing it); =0: no riders.
        red = getattr(model, "_reducer", None)
        env = _ab.get("adam_ride")
        self._ride_env = env
        self.ride = (self.pipeline and env != "0" and fl.compute_dtype == torch.bfloat16
                     and (red is None or red.capturable) and getattr(fl, "shard_reducer", None) is None)
        if self.ride and not getattr(fl, "adam_ride", False):
            fl.adam_ride = True
            model._steps = {}            # the forward plans are rebuilt with rider slots (engine.EncoderStack.build_forward)
        self._copy_stream = self._loss_host = self._loss_ev = None
        self._g_rest = None
        self.params = [p for p in model.parameters()]
        self.calls = 0
        self.mode = None                 # None (not captured) | "whole" | "segmented"
        self._static_args, self._static_kw, self._static_key = None, None, None
        self._g_all = self._g_fwd = self._g_opt = None
        self._side = None
        self.loss = None

    # ------------------------------------------------------------------------------------------------ pieces
    def _clip_and_step(self, defer=False):
        if self.max_grad_norm is not None:
            clip_grad_norm_(self.params, self.max_grad_norm)
        self.opt.step(defer=defer) if defer else self.opt.step()

    def _eager(self, args, kw, defer=False):
        self.model._in_pipelined_call = defer
        try:
            loss = self.model(*args, **kw)
        finally:
            self.model._in_pipelined_call = False
        loss.backward()
        self._clip_and_step(defer)
        if defer:
            self.model._pending_update = self.opt
        self.opt.zero_grad()
        return loss

    def _launch_pending_update(self):
        """BertAdam of the PREVIOUS iteration, chunk group by chunk group on a second stream; one event per group for the
        forward plan's wait points."""
        cur = torch.cuda.current_stream()
        side, events = self._side, self.model._param_events
        events.clear()
        side.wait_stream(cur)

        def on_group(key):
            ev = torch.cuda.Event()
            ev.record(side)
            events[key] = ev

        with torch.cuda.stream(side):
            groups = self.opt.chunk_groups()
            self.opt.launch_deferred(groups=groups, on_group=on_group, max_blocks=self.adam_blocks)
            on_group("all")

    def _forward_pipelined(self, args, kw):
        if self.ride:
            return self._forward_riding(args, kw)
        self._launch_pending_update()
        self.model._in_pipelined_call = True
        try:
            return self.model(*args, **kw)
        finally:
            self.model._in_pipelined_call = False
            self.model._param_events.clear()

    def _forward_riding(self, args, kw):
        """The pending BertAdam update is applied BY the forward: UniVL.forward launches what cannot ride, its forward products
        carry the rest (UniVL._start_riding_update).  Same stream throughout: no event, no graph branch."""
        opt, model = self.opt, self.model
        fl = model.flat
        assert getattr(fl, "shard_reducer", None) is None, "UNIVL_ADAM_RIDE does not combine with the sharded optimizer"
        model._rider_update = dict(desc=opt._last_desc, groups=opt.chunk_groups(), max_blocks=self.adam_blocks)
        opt._deferred = False                 # from here on the update counts as applied (launch_deferred's bookkeeping)
        fl.shadow_valid = True
        model._in_pipelined_call = True
        try:
            return model(*args, **kw)
        finally:
            model._in_pipelined_call = False
            model._rider_update = None

    def flush(self):
        """Apply the pending BertAdam update of the last iteration (pipelined mode); no-op otherwise."""
        self.opt.flush()

    def _read_back_loss(self):
        """Copy self.loss to pinned host memory on the copy stream, behind everything enqueued so far (= the forward)."""
        cur = torch.cuda.current_stream()
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=cur.device)
            self._loss_host = torch.zeros(1, dtype=torch.float32).pin_memory()
            self._loss_ev = torch.cuda.Event()
        cs = self._copy_stream
        cs.wait_stream(cur)
        with torch.cuda.stream(cs):
            self._loss_host.copy_(self.loss.detach().reshape(1), non_blocking=True)
            self._loss_ev.record(cs)
        self.loss.__dict__["_async"] = (self._loss_host, self._loss_ev)

    def _stage(self, args, kw):
        """Bring the new batch into the static input buffers the graphs read from."""
        if self._static_args is None:
            dev = next(self.model.parameters()).device
            keep = lambda t: self.persistent and t.device == dev
            hold = lambda t: t if keep(t) else t.to(dev, copy=True)
            self._static_args = [hold(a) if isinstance(a, torch.Tensor) else a for a in args]
            self._static_kw = {k: (hold(v) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
            # the static objects stay alive here, so an equal id() later means the very same object: nothing to stage
            self._static_key = tuple(id(a) for a in self._static_args) + tuple((k, id(v)) for k, v in self._static_kw.items())
            return
        if tuple(id(a) for a in args) + tuple((k, id(v)) for k, v in kw.items()) == self._static_key:
            return                           # the caller refilled the static buffers in place (or passes them unchanged)
        if len(args) != len(self._static_args) or set(kw) != set(self._static_kw):
            raise RuntimeError("GraphedTrainStep: the call signature changed after capture")
        pairs = list(zip(self._static_args, args)) + [(self._static_kw[k], kw[k]) for k in kw]
        done = set()
        for st, new in pairs:
            if isinstance(st, torch.Tensor):
                if not isinstance(new, torch.Tensor) or new.shape != st.shape or new.dtype != st.dtype:
                    raise RuntimeError("GraphedTrainStep: input shape/dtype changed after capture (%s -> %s); build a "
                                       "second GraphedTrainStep for the other batch shape" %
                                       (tuple(st.shape), tuple(getattr(new, "shape", ()))))
                if new is not st and (id(st), id(new)) not in done:
                    done.add((id(st), id(new)))     # the same tensor passed under two names (masked_video=video) goes once
                    stage_input(st, new)
            elif st is not new and st != new:
                raise RuntimeError("GraphedTrainStep: a non-tensor argument changed after capture")

    # ------------------------------------------------------------------------------------------------- call
    def __call__(self, *args, **kw):
        self.calls += 1
        if self.calls <= self.warmup:
            return self._eager(args, kw)
        self._stage(args, kw)
        sa, sk = self._static_args, self._static_kw
        if self.pipeline and self._side is None:          # created outside any capture
            self._side = torch.cuda.Stream(device=next(self.model.parameters()).device)
        if self.pipeline and not self.opt.has_pending:
            # first pipelined call, or somebody flushed (evaluation, checkpoint): this iteration runs unpipelined up to its
            # clip; its BertAdam update rides with the next forward
            self.loss_eager = self._eager(sa, sk, defer=True)
            return self.loss_eager
        if self._loss_ev is not None:
            torch.cuda.current_stream().wait_event(self._loss_ev)     # the previous loss copy reads what this replay overwrites
        if self.mode is None:
            torch.cuda.synchronize()
            if self.ride:                     # the rider kernels are first launched inside the capture below
                import ctypes as C
                from . import _lib
                _lib.check(_lib.lib().univl_gemm_rider_prime(C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm_rider_prime")
            red0 = getattr(self.model, "_reducer", None)
            one_graph = red0 is None or red0.capturable      # no exchange, or an exchange that is part of the plan (univl_amd.rccl)
            if self.ride and red0 is not None and not red0.capturable:
                # the riders were planned before a gradient exchange through torch's process group appeared (enable_data_parallel or
                # the DDP wrapper AFTER this object was built): that combination was never validated
                raise RuntimeError("GraphedTrainStep: a gradient exchange that cannot be captured was enabled after construction; build "
                                   "the GraphedTrainStep after enable_data_parallel() / the DistributedDataParallel wrap")
            split = self.async_loss or (self.ride and red0 is not None and self._ride_env != "force")
            if one_graph and split:
                # two graphs from one memory pool: forward | backward + clip + BertAdam
                self._g_fwd, self._g_rest = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                # thread_local: with a captured gradient exchange the process group's watchdog thread may query events meanwhile
                with no_gc(), torch.cuda.graph(self._g_fwd, capture_error_mode="thread_local"):
                    self.loss = self._forward_pipelined(sa, sk) if self.pipeline else self.model(*sa, **sk)
                with no_gc(), torch.cuda.graph(self._g_rest, pool=self._g_fwd.pool(), capture_error_mode="thread_local"):
                    self.loss.backward()
                    self._clip_and_step(defer=self.pipeline)
                    if self.pipeline:
                        self.model._pending_update = self.opt
                    self.opt.zero_grad()
                self.mode = "whole"
            elif one_graph:
                self._g_all = torch.cuda.CUDAGraph()
                with no_gc(), torch.cuda.graph(self._g_all, capture_error_mode="thread_local"):
                    if self.pipeline:
                        self.loss = self._forward_pipelined(sa, sk)
                        self.loss.backward()
                        self._clip_and_step(defer=True)
                        self.opt.zero_grad()
                    else:
                        self.loss = self._eager(sa, sk)
                self.mode = "whole"
            else:
                self.model.graph_backward = True
                self._g_fwd = torch.cuda.CUDAGraph()
                with no_gc(), torch.cuda.graph(self._g_fwd, capture_error_mode="thread_local"):
                    self.loss = self._forward_pipelined(sa, sk) if self.pipeline else self.model(*sa, **sk)
                self.mode = "segmented"
        if self.mode == "whole":
            if self._g_rest is not None:
                self._g_fwd.replay()
                if self.async_loss:
                    self._read_back_loss()
                self._g_rest.replay()
            else:
                self._g_all.replay()
            return self.loss
        red = getattr(self.model.flat, "shard_reducer", None)
        if red is not None:
            red.join()                       # sharded optimizer: the all-gather of the updated shadow must have landed
        self._g_fwd.replay()
        if self.async_loss:
            self._read_back_loss()
        self.loss.backward()                 # captured segments + host-issued all-reduces + join (Plan.run_graphed)
        if getattr(self.model.flat, "shard_reducer", None) is not None:
            self._clip_and_step(defer=False)         # sharded optimizer: the norm all-reduce / shadow all-gather are host-issued
        else:
            if self._g_opt is None:
                self._g_opt = torch.cuda.CUDAGraph()
                with no_gc(), torch.cuda.graph(self._g_opt, capture_error_mode="thread_local"):
                    self._clip_and_step(defer=self.pipeline)
            self._g_opt.replay()
        self.opt.zero_grad()                 # host-side only: the next backward starts from beta = 0 again
        return self.loss
