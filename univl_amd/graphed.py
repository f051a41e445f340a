"""The training iteration of main_task_retrieval.py:333-352 (forward, backward, gradient exchange, clip, BertAdam,
zero_grad) replayed as hipGraphs.

At 4-16 pairs per GPU a step is ~380 kernels of 2-10 us each; launched one by one from Python the host is the
bottleneck (~10 us per ctypes call).  `GraphedTrainStep` captures the iteration once and replays it:

  * one process, no gradient exchange: ONE graph holds the whole iteration;
  * data parallel (UniVL.enable_data_parallel): RCCL collectives stay OUTSIDE the graphs -- the forward is one graph,
    the backward plan is cut at its gradient-exchange points into captured segments (engine.Plan.run_graphed) with
    the all-reduces issued from the host between the replays (they run on RCCL's own stream and overlap the
    following segments), and clip + BertAdam are a last graph that is replayed after the reducer's join.

The first `warmup` calls run eagerly (they build the execution plans and the optimizer tables); the next call
captures and runs; later calls only copy the new batch into the static input buffers and replay.
"""
import torch

from .optimization import clip_grad_norm_
from .steps import stage_input


class GraphedTrainStep:
    def __init__(self, model, optimizer, max_grad_norm=1.0, warmup=3, persistent_inputs=False):
        """persistent_inputs=True: the caller passes the SAME tensors every time and refills them in place (no
        per-step copy into private static buffers)."""
        self.model, self.opt = model, optimizer
        self.max_grad_norm = max_grad_norm
        self.warmup = int(warmup)
        self.persistent = bool(persistent_inputs)
        self.params = [p for p in model.parameters()]
        self.calls = 0
        self.mode = None                 # None (not captured) | "whole" | "segmented"
        self._static_args, self._static_kw = None, None
        self._g_all = self._g_fwd = self._g_opt = None
        self.loss = None

    # ------------------------------------------------------------------------------------------------ pieces
    def _clip_and_step(self):
        if self.max_grad_norm is not None:
            clip_grad_norm_(self.params, self.max_grad_norm)
        self.opt.step()

    def _eager(self, args, kw):
        loss = self.model(*args, **kw)
        loss.backward()
        self._clip_and_step()
        self.opt.zero_grad()
        return loss

    def _stage(self, args, kw):
        """Bring the new batch into the static input buffers the graphs read from."""
        if self._static_args is None:
            dev = next(self.model.parameters()).device
            hold = (lambda t: t) if self.persistent else (lambda t: t.to(dev, copy=True))
            self._static_args = [hold(a) if isinstance(a, torch.Tensor) else a for a in args]
            self._static_kw = {k: (hold(v) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
            return
        if len(args) != len(self._static_args) or set(kw) != set(self._static_kw):
            raise RuntimeError("GraphedTrainStep: the call signature changed after capture")
        pairs = list(zip(self._static_args, args)) + [(self._static_kw[k], kw[k]) for k in kw]
        for st, new in pairs:
            if isinstance(st, torch.Tensor):
                if not isinstance(new, torch.Tensor) or new.shape != st.shape or new.dtype != st.dtype:
                    raise RuntimeError("GraphedTrainStep: input shape/dtype changed after capture (%s -> %s); build a "
                                       "second GraphedTrainStep for the other batch shape" %
                                       (tuple(st.shape), tuple(getattr(new, "shape", ()))))
                if new is not st:
                    stage_input(st, new)           # host batches go through a pinned buffer (one async DMA)
            elif st is not new and st != new:
                raise RuntimeError("GraphedTrainStep: a non-tensor argument changed after capture")

    # ------------------------------------------------------------------------------------------------- call
    def __call__(self, *args, **kw):
        self.calls += 1
        if self.calls <= self.warmup:
            return self._eager(args, kw)
        self._stage(args, kw)
        sa, sk = self._static_args, self._static_kw
        if self.mode is None:
            torch.cuda.synchronize()
            if getattr(self.model, "_reducer", None) is None:
                self._g_all = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._g_all):
                    self.loss = self._eager(sa, sk)
                self.mode = "whole"
            else:
                self.model.graph_backward = True
                self._g_fwd = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._g_fwd, capture_error_mode="thread_local"):
                    self.loss = self.model(*sa, **sk)
                self.mode = "segmented"
        if self.mode == "whole":
            self._g_all.replay()
            return self.loss
        self._g_fwd.replay()
        self.loss.backward()                 # captured segments + host-issued all-reduces + join (Plan.run_graphed)
        if self._g_opt is None:
            self._g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_opt, capture_error_mode="thread_local"):
                self._clip_and_step()
        self._g_opt.replay()
        self.opt.zero_grad()                 # host-side only: the next backward starts from beta = 0 again
        return self.loss
