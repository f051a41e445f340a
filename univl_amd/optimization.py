"""Drop-in for the reference's `modules/optimization.py::BertAdam` and for the `clip_grad_norm_` call of the
training loop (main_task_retrieval.py:347), fused over the flat parameter / gradient buffers of
`univl_amd.engine.FlatParams`: three kernel launches per step instead of ~12 per parameter tensor.

Same constructor, param-group semantics, `state[p] = {step, next_m, next_v}` keys (so optimizer checkpoints of
main_pretrain.py:266-273 interchange) and update rule (optimization.py:103-168): per-parameter clip, Adam moments
without bias correction, decoupled weight decay, warmup-linear learning rate, parameters whose `.grad is None`
are skipped.  Runs only on parameters owned by a univl_amd model on a HIP device (no CPU fallback).
"""
import ctypes as C

import torch
from torch.optim import Optimizer

from . import _ab, _lib
from .engine import FLAT_REGISTRY

CHUNK = 8192   # elements per workgroup


import math


def warmup_cosine(x, warmup=0.002):
    """optimization.py:26-29."""
    return x / warmup if x < warmup else 0.5 * (1.0 + math.cos(math.pi * x))


def warmup_constant(x, warmup=0.002):
    """optimization.py:31-36."""
    return x / warmup if x < warmup else 1.0


def warmup_linear(x, warmup=0.002):
    """optimization.py:38-43."""
    return x / warmup if x < warmup else max((x - 1.) / (warmup - 1.), 0)


SCHEDULES = {'warmup_cosine': warmup_cosine, 'warmup_constant': warmup_constant, 'warmup_linear': warmup_linear}
_SCHEDULE_CODE = {'warmup_linear': 0, 'warmup_cosine': 1, 'warmup_constant': 2}      # UnivlAdam.schedule


def _find_flat(p):
    ptr = p.data_ptr()
    for fl in list(FLAT_REGISTRY):
        lo = fl.p32.data_ptr()
        if lo <= ptr < lo + fl.total * 4:
            return fl
    raise RuntimeError("univl_amd.optimization: parameter is not owned by a univl_amd model on a HIP device "
                       "(call the model once, or access model.flat, after model.to('cuda')); there is no CPU fallback")


def owns(p):
    """True when `p` is a parameter of a univl_amd model that lives in a flat buffer on a HIP device."""
    try:
        _find_flat(p)
        return True
    except RuntimeError:
        return False


class _Tables:
    """Device-side segment / chunk tables for one FlatParams and one (lr, wd, max_norm, active) assignment."""

    def __init__(self, fl, seg_cfg):
        dev = fl.device
        nseg = len(fl.order)
        segs = (_lib.Seg * nseg)()
        c_seg, c_off, c_len = [], [], []
        owned = getattr(fl, "owned", None)       # sharded optimizer: only the pieces of the flat buffer this rank owns
        oi = 0
        for s, n in enumerate(fl.order):
            off, numel, _ = fl.index[n]
            lr, wd, mgn, active = seg_cfg.get(n, (0.0, 0.0, 0.0, 0))
            segs[s].offset, segs[s].numel = off, numel
            segs[s].lr, segs[s].weight_decay, segs[s].max_grad_norm, segs[s].active = lr, wd, mgn, int(active)
            if not active:
                continue                    # no workgroups for tensors that take no part
            if owned is None:
                pieces = [(off, off + numel)]
            else:                           # tensors and owned ranges are both sorted by offset
                while oi > 0 and owned[oi - 1][1] > off:
                    oi -= 1
                while oi < len(owned) and owned[oi][1] <= off:
                    oi += 1
                pieces, j = [], oi
                while j < len(owned) and owned[j][0] < off + numel:
                    lo, hi = max(off, owned[j][0]), min(off + numel, owned[j][1])
                    if hi > lo:
                        pieces.append((lo, hi))
                    j += 1
            # the word table's chunks start on row boundaries (10 rows of 768): UnivlAdam.row_flags works on whole rows
            chunk = int(_ab.get("adam_chunk")) or CHUNK
            step = chunk
            if n == fl.WORD and owned is None:
                row = fl.index[n][2][1]
                step = max(1, chunk // row) * row
            for lo, hi in pieces:
                for o in range(lo, hi, step):
                    c_seg.append(s)
                    c_off.append(o)
                    c_len.append(min(step, hi - o))
        raw = bytes(segs)
        self.segs = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.chunk_seg_host = c_seg
        self.chunk_seg = torch.tensor(c_seg, dtype=torch.int32, device=dev)
        self.chunk_off = torch.tensor(c_off, dtype=torch.int64, device=dev)
        self.chunk_len = torch.tensor(c_len, dtype=torch.int32, device=dev)
        self.nseg, self.nchunk = nseg, len(c_seg)
        self.sumsq = torch.zeros(nseg, device=dev)
        self.coef = torch.ones(2, device=dev)
        self.scalars = torch.zeros(2 * nseg, device=dev)
        self.key = tuple(sorted(seg_cfg.items()))
        self.owned_id = getattr(fl, "partition_version", 0)     # bumped by every enable_data_parallel(): id(list) can be recycled


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _sumsq(fl, tb, out=None, zero=True):
    out = tb.sumsq if out is None else out
    if zero:
        out.zero_()
    if tb.nchunk > 0:
        _lib.check(_lib.lib().univl_grad_sumsq(fl.g32.data_ptr(), tb.segs.data_ptr(), tb.nseg, tb.chunk_seg.data_ptr(),
                                               tb.chunk_off.data_ptr(), tb.chunk_len.data_ptr(), tb.nchunk,
                                               out.data_ptr(), _stream()), "grad_sumsq")
    red = getattr(fl, "shard_reducer", None)
    if red is not None:                   # sharded optimizer: every rank measured its own pieces of every tensor
        red.all_reduce_small(out)


def _measure(fl, cfg, tb):
    """Per-tensor sums of squares of the gradients in `cfg` -> device tensor [nseg].  Tensors whose weight-gradient GEMM
    already accumulated its own sum during this backward (engine.GradState.sumsq_args) are not read again: only the
    rest (embedding tables, vectors, a few small matrices) goes through the streaming kernel."""
    fused = fl.fused
    # the epilogue sums describe the gradients as the backward left them: any in-place change since then made through
    # torch (stock clip_grad_norm_, p.grad.mul_(), AMP unscale -- all bump the flat buffer's version counter) voids them
    if fused is not None and fused.get("tv", fl.g32._version) != fl.g32._version:
        fused = fl.fused = None
    if fused is not None and fused["version"] == fl.grad_version and fused["names"] <= set(cfg) and not fused.get("consumed"):
        fused["consumed"] = True          # fl.sumsq may be completed once per backward; later callers re-measure
        rest = {n: c for n, c in cfg.items() if n not in fused["names"]}
        rows = getattr(fl, "_word_rows", None)
        sparse_table = (rows is not None and getattr(fl, "word_rows_version", -1) == fl.grad_version and fl.WORD in rest)
        if sparse_table:                  # only the listed rows of the word table are non-zero: no 94 MB read for its norm
            rest = {n: c for n, c in rest.items() if n != fl.WORD}
        key = tuple(sorted(rest))
        tr = fl._rest[1] if (getattr(fl, "_rest", None) is not None and fl._rest[0] == key) else None
        if tr is None:
            tr = _Tables(fl, rest)
            fl._rest = (key, tr)
        _sumsq(fl, tr, out=fl.sumsq, zero=False)       # the backward zeroed fl.sumsq before its GEMMs added to it
        if sparse_table:
            from . import ops
            seg = fl.seg_of[fl.WORD]
            ops.rows_sumsq(fl.g(fl.WORD), rows[0], rows[1], fl.sumsq[seg:seg + 1])
        return fl.sumsq
    _sumsq(fl, tb)
    return tb.sumsq


def _active_cfg(fl, params_with_cfg):
    """{name: (lr, weight_decay, max_grad_norm, 1)} for the parameters that currently hold a gradient.  Runs once per clip
    and once per step over ~300 parameters, so it sticks to identity checks (flat.name_of, cached gradient views)."""
    cfg = {}
    name_of, gviews = fl.name_of, fl.g
    for p, (lr, wd, mgn) in params_with_cfg:
        n = name_of.get(id(p))
        if n is None:
            raise RuntimeError("univl_amd.optimization: parameter not found in the model's flat buffer")
        gr = p.grad
        if gr is None:
            continue
        gv = gviews(n)
        if gr is not gv:
            if gr.data_ptr() != gv.data_ptr():
                gv.copy_(gr)               # a foreign gradient tensor: bring it into the flat buffer
                fl.fused = None            # ... whose norm nobody has measured
                if n == fl.WORD:
                    fl.mark_all_word_rows()                 # ... and whose non-zero rows nobody listed
                    if getattr(fl, "_word_rows", None) is not None:
                        fl._word_rows[1][1] = 1
            p.grad = gv
        cfg[n] = (lr, wd, mgn, 1)
    return cfg


def apply_pending_clip(fl):
    """A deferred clip (clip_grad_norm_(..., deferred=True)) that BertAdam.step did not consume -- another backward is about
    to add to the gradients, or the optimizer holds a different parameter set -- is applied in place, exactly where torch
    would have scaled the gradients.  Never dropped."""
    pend, fl._pending = fl._pending, None
    if pend is None:
        return
    tb = pend["tables"]
    _lib.check(_lib.lib().univl_scale_grads(fl.g32.data_ptr(), tb.segs.data_ptr(), tb.chunk_seg.data_ptr(),
                                            tb.chunk_off.data_ptr(), tb.chunk_len.data_ptr(), tb.nchunk, pend["coef"].data_ptr(),
                                            _stream()), "scale_grads")
    fl.fused = None                        # the gradients were rescaled in place


def clip_grad_norm_(parameters, max_norm, norm_type=2.0, deferred=True):
    """Fused torch.nn.utils.clip_grad_norm_ for univl_amd models (main_task_retrieval.py:347).

    One streaming read of the flat gradient buffer gives every per-tensor sum of squares (re-used by BertAdam's
    per-parameter clip).  deferred=True (default) folds the clip coefficient into the following BertAdam.step
    instead of rewriting 4 B/param of gradients; deferred=False scales the gradients in place like torch does.
    Returns the total norm (0-d device tensor)."""
    assert float(norm_type) == 2.0, "only the L2 norm is used by the reference"
    params = [p for p in (parameters if not isinstance(parameters, torch.Tensor) else [parameters]) if p.grad is not None]
    if not params:
        return torch.tensor(0.0)
    fl = _find_flat(params[0])
    if fl._pending is not None:            # two clips in a row: the first one takes effect now
        apply_pending_clip(fl)
    cfg = _active_cfg(fl, [(p, (0.0, 0.0, 0.0)) for p in params])
    key = tuple(sorted(cfg))
    ckey = key + (getattr(fl, "partition_version", 0),)
    tb = fl._clip[1] if (fl._clip is not None and fl._clip[0] == ckey) else None
    if tb is None:
        tb = _Tables(fl, cfg)
        fl._clip = (ckey, tb)
    sumsq = _measure(fl, cfg, tb)
    L = _lib.lib()
    _lib.check(L.univl_clip_coef(sumsq.data_ptr(), tb.segs.data_ptr(), tb.nseg, float(max_norm), tb.coef.data_ptr(),
                                 _stream()), "clip_coef")
    if not deferred:
        _lib.check(L.univl_scale_grads(fl.g32.data_ptr(), tb.segs.data_ptr(), tb.chunk_seg.data_ptr(),
                                       tb.chunk_off.data_ptr(), tb.chunk_len.data_ptr(), tb.nchunk, tb.coef.data_ptr(),
                                       _stream()), "scale_grads")
        fl._pending = None
        fl.fused = None                    # the gradients were rescaled in place
    else:
        fl._pending = dict(version=fl.grad_version, names=key, sumsq=sumsq, coef=tb.coef, tables=tb)
    return tb.coef[1]


class BertAdam(Optimizer):
    """Fused BERT-Adam (see module docstring).  Arguments as modules/optimization.py:66-84."""

    def __init__(self, params, lr=None, warmup=-1, t_total=-1, schedule='warmup_linear', b1=0.9, b2=0.999, e=1e-6,
                 weight_decay=0.01, max_grad_norm=1.0):
        if lr is None or lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if schedule not in SCHEDULES:
            raise ValueError("Invalid schedule parameter: {}".format(schedule))
        if not 0.0 <= warmup < 1.0 and not warmup == -1:
            raise ValueError("Invalid warmup: {} - should be in [0.0, 1.0[ or -1".format(warmup))
        if not 0.0 <= b1 < 1.0:
            raise ValueError("Invalid b1 parameter: {} - should be in [0.0, 1.0[".format(b1))
        if not 0.0 <= b2 < 1.0:
            raise ValueError("Invalid b2 parameter: {} - should be in [0.0, 1.0[".format(b2))
        if not e >= 0.0:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(e))
        defaults = dict(lr=lr, schedule=schedule, warmup=warmup, t_total=t_total, b1=b1, b2=b2, e=e,
                        weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        super(BertAdam, self).__init__(params, defaults)
        g0 = self.param_groups[0]
        for g in self.param_groups:
            for k in ("warmup", "t_total", "b1", "b2", "e", "schedule"):
                if g[k] != g0[k]:
                    raise ValueError("univl_amd BertAdam: '%s' must be the same in every param group" % k)
        self._fl = None
        self._tb = None
        self._m = self._v = self._step_dev = None
        self._auto_deferred = False     # ... and step() itself did, for the model's next forward to apply (the unchanged training loop)
        self._deferred = False          # step(defer=True) prepared an update that launch_deferred() / flush() still has to enqueue
        self._last_desc = None

    def get_lr(self):
        """optimization.py:86-101 (step counters are read back from the device: under hipGraph replay the host never
        sees the individual steps)."""
        self._sync_steps()
        lr = []
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    return [0]
                if group['t_total'] != -1:
                    lr_scheduled = group['lr'] * SCHEDULES[group['schedule']](state['step'] / group['t_total'], group['warmup'])
                else:
                    lr_scheduled = group['lr']
                lr.append(lr_scheduled)
        return lr

    def relaunch_last(self):
        """Re-enqueue the kernels of the previous step() with the same descriptor (no host-side bookkeeping):
        used by bench.py to time the update kernel with HIP events without being host-bound."""
        _lib.check(_lib.lib().univl_bert_adam(C.byref(self._last_desc), _stream()), "bert_adam")

    # ------------------------------------------------------------------------------------------- state plumbing
    # The moments live in two flat buffers laid out like the model's flat parameter buffer; state[p]['next_m'/'next_v'] are
    # views into them and state[p]['step'] mirrors a per-tensor device counter, so that the reference's optimizer
    # checkpoints (main_pretrain.py:266-273, 389) interchange in both directions.
    def _link(self, fl, p, name, step=None):
        o, k, shp = fl.index[name]
        st = self.state[p]
        if step is None:
            step = st.get('step', 0)
        st['step'] = int(step)
        st['next_m'] = self._m[o:o + k].view(shp)
        st['next_v'] = self._v[o:o + k].view(shp)

    def _bind(self):
        p0 = self.param_groups[0]['params'][0]
        fl = _find_flat(p0)
        if self._fl is fl:
            return fl
        old, m_old, v_old, s_old = self._fl, self._m, self._v, self._step_dev
        self._fl = fl
        self._m = torch.zeros(fl.total, device=fl.device)
        self._v = torch.zeros(fl.total, device=fl.device)
        self._step_dev = torch.zeros(len(fl.order), device=fl.device, dtype=torch.int32)
        self._tb = None
        # existing state (a model moved with .to()/.float() re-flattens its parameters; a checkpoint loaded before the first
        # step holds free-standing tensors): migrate it into the new flat buffers instead of starting from zero
        steps = s_old.cpu().tolist() if (old is not None and s_old is not None) else None
        for p, st in list(self.state.items()):
            name = fl.name_of.get(id(p))
            if name is None or not st:
                continue
            o, k, shp = fl.index[name]
            step = st.get('step', 0)
            if old is not None and name in old.index and old.index[name][1] == k:
                oo = old.index[name][0]
                self._m[o:o + k].copy_(m_old[oo:oo + k])
                self._v[o:o + k].copy_(v_old[oo:oo + k])
                step = steps[old.seg_of[name]]
            elif 'next_m' in st:
                self._m[o:o + k].copy_(st['next_m'].reshape(-1))
                self._v[o:o + k].copy_(st['next_v'].reshape(-1))
            self._step_dev[fl.seg_of[name]] = int(step)
            self._link(fl, p, name, step)
            fl.mark_all_word_rows()         # migrated moments: no row of the word table is known to be at m = v = 0 in THIS buffer set
        return fl

    def _sync_steps(self):
        fl = self._fl
        if fl is None or self._step_dev is None:
            return
        steps = self._step_dev.cpu().tolist()
        for p, st in self.state.items():
            name = fl.name_of.get(id(p))
            if name is not None and st:
                st['step'] = int(steps[fl.seg_of[name]])

    def state_dict(self):
        """Same structure as the reference optimizer's: state[i] = {step, next_m, next_v} with free-standing tensors."""
        self.flush()
        if getattr(self._fl, "shard_reducer", None) is not None and not getattr(self, "_state_complete", False):
            raise RuntimeError("BertAdam.state_dict(): the optimizer state is sharded over the ranks -- call "
                               "optimizer.consolidate() on EVERY rank first (a collective), then state_dict() where needed")
        self._sync_steps()
        sd = super().state_dict()
        for st in sd['state'].values():
            for k in ('next_m', 'next_v'):
                if k in st:
                    st[k] = st[k].clone()
        return sd

    def load_state_dict(self, state_dict):
        """optimizer.load_state_dict(checkpoint['last_optimizer_state']) of main_pretrain.py:389: torch installs fresh
        tensors under state[p]; they are copied into the flat moment buffers, the device step counters are set from
        state['step'], and the entries are re-pointed at the flat views the kernel updates."""
        self.flush()                        # a pending update reads the moments this call replaces
        super().load_state_dict(state_dict)
        try:
            fl = _find_flat(self.param_groups[0]['params'][0])
        except RuntimeError:
            return                          # model not on the device yet: _bind() migrates the loaded tensors later
        if self._fl is not fl:
            self._fl = None                 # force a fresh bind that migrates from the loaded tensors
            self._bind()
            return
        fl.mark_all_word_rows()             # loaded moments: no row of the word table is known to be at m = v = 0
        for p, st in self.state.items():
            name = fl.name_of.get(id(p))
            if name is None or 'next_m' not in st:
                continue
            o, k, _ = fl.index[name]
            self._m[o:o + k].copy_(st['next_m'].reshape(-1))
            self._v[o:o + k].copy_(st['next_v'].reshape(-1))
            self._step_dev[fl.seg_of[name]] = int(st.get('step', 0))
            self._link(fl, p, name)

    # --------------------------------------------------------------------------------------- deferred / layer-wise update
    @property
    def has_pending(self):
        return self._deferred

    def chunk_groups(self):
        """[(key, first_chunk, n_chunks)] over the chunk table of the last step(), in the order a forward pass needs the
        parameters: ("base") = embedding tables, every vector and the matrices outside the layer stacks (contiguous runs, so
        possibly several entries), then ("layer", prefix, l) with the text / video stacks interleaved, the cross encoder and
        the decoder after them."""
        import re
        fl, tb = self._fl, self._tb
        cached = getattr(self, "_groups_cache", None)
        if cached is not None and cached[0] is tb:        # (thousands of chunks: the unchanged loop asks once per forward)
            return cached[1]
        runs = []
        for c, s_ in enumerate(tb.chunk_seg_host):
            name = fl.order[s_]
            m = re.match(r"^(bert|visual|cross)\.encoder\.layer\.(\d+)\.", name) or re.match(r"^(decoder)\.decoder\.layer\.(\d+)\.", name)
            key = ("layer", m.group(1), int(m.group(2))) if (m and not fl.is_atomic(name)) else "base"
            if runs and runs[-1][0] == key and runs[-1][1] + runs[-1][2] == c:
                runs[-1][2] += 1
            else:
                runs.append([key, c, 1])
        stage = {"bert": 0, "visual": 0, "cross": 1, "decoder": 2}
        base = [tuple(r) for r in runs if r[0] == "base"]
        layers = sorted((tuple(r) for r in runs if r[0] != "base"), key=lambda r: (stage[r[0][1]], r[0][2], r[0][1]))
        self._groups_cache = (tb, base + layers)
        return base + layers

    def launch_deferred(self, groups=None, on_group=None, max_blocks=0):
        """Enqueue the update prepared by step(defer=True) on the current stream: in one launch (groups None) or one launch
        per chunk group, calling on_group(key) after the last launch of each key."""
        d = self._last_desc
        if not self._deferred or d is None:
            return
        L, h = _lib.lib(), _stream()
        if groups is None:
            _lib.check(L.univl_bert_adam(C.byref(d), h), "bert_adam")
        else:
            for i, (key, c0, n) in enumerate(groups):
                _lib.check(L.univl_bert_adam_range(C.byref(d), c0, n, 1 if i == 0 else 0, int(max_blocks), h), "bert_adam_range")
                if on_group is not None and (i + 1 == len(groups) or groups[i + 1][0] != key):
                    on_group(key)
        self._deferred = False
        # (the update rewrites the bf16 shadow of the tensors it updates: a valid shadow stays valid, and one that was marked dirty --
        # tensors outside this update may be stale -- stays dirty until refresh_shadow)
        self._after_update(self._fl)

    def _after_update(self, fl):
        """Sharded optimizer: this rank updated its pieces only -- all-gather what the next forward reads (the bf16 shadow;
        the fp32 master in fp32 compute mode).  Asynchronous; UniVL.forward joins it."""
        red = getattr(fl, "shard_reducer", None)
        if red is not None:
            red.all_gather_ranges(fl.p16 if fl.p16 is not None else fl.p32)
            if getattr(fl, "p16lo", None) is not None:
                red.all_gather_ranges(fl.p16lo)
            fl.master_complete = fl.p16 is None        # other ranks' pieces of the fp32 master are stale from now on
            self._state_complete = False               # ... and so are their pieces of the moments

    def consolidate(self):
        """Sharded optimizer, before optimizer.state_dict(): COLLECTIVE (every rank calls it) -- gathers the moments of all
        shards so that each rank holds the complete state the reference's optimizer checkpoint has."""
        fl = self._fl
        red = getattr(fl, "shard_reducer", None) if fl is not None else None
        if red is not None:
            red.all_gather_ranges(self._m)
            red.all_gather_ranges(self._v)
            red.join()
            self._state_complete = True

    def flush(self):
        """Apply a deferred update now (checkpointing, evaluation, a forward outside the pipelined loop)."""
        if self._deferred:
            self._auto_deferred = False
            self.launch_deferred()

    def _can_ride(self, fl):
        """step() without arguments may leave its launch to the model's next forward (riding update): bf16 compute on a HIP device, one
        process (no gradient exchange: the data-parallel step keeps its own schedule, graphed.GraphedTrainStep), the model in training
        mode with rider slots in its plans, nobody capturing."""
        model = getattr(fl, "owner", lambda: None)()
        return bool(model is not None and getattr(fl, "adam_ride", False) and fl.compute_dtype == torch.bfloat16 and fl.device.type == "cuda"
                    and getattr(fl, "shard_reducer", None) is None and getattr(fl, "owned", None) is None
                    and model.training and getattr(model, "auto_ride", True) and model._reducer is None
                    and not model._in_pipelined_call and not torch.cuda.is_current_stream_capturing())

    def zero_grad(self, set_to_none=True):
        if not set_to_none:
            self.flush()              # grad.zero_() writes the gradient buffer a pending update still has to read
        return super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None, defer=False):
        """defer=True prepares the update (descriptor, per-tensor scalars inputs) but leaves its launch to launch_deferred():
        univl_amd.graphed.GraphedTrainStep overlaps it with the next forward pass."""
        loss = closure() if closure is not None else None
        if self._deferred:
            self.flush()
        fl = self._bind()
        auto = (not defer) and closure is None and self._can_ride(fl)
        pw = []
        for group in self.param_groups:
            for p in group['params']:
                pw.append((p, (group['lr'], group['weight_decay'], group['max_grad_norm'])))
        cfg = _active_cfg(fl, pw)
        key = tuple(sorted(cfg.items()))
        if self._tb is None or self._tb.key != key or self._tb.owned_id != getattr(fl, "partition_version", 0):
            self._tb = _Tables(fl, cfg)
        tb = self._tb
        pend = getattr(fl, "_pending", None)
        coef_ptr = None
        if pend is not None and pend["version"] == fl.grad_version and pend["names"] == tuple(sorted(cfg)):
            sumsq, coef_ptr = pend["sumsq"], pend["coef"].data_ptr()      # clip already measured these gradients
            fl._pending = None
        else:
            if pend is not None:           # the clip saw another parameter set: it takes effect in place, like torch's
                apply_pending_clip(fl)
            sumsq = _measure(fl, cfg, tb)
        g0 = self.param_groups[0]
        d = _lib.Adam()
        d.p, d.g, d.m, d.v = fl.p32.data_ptr(), fl.g32.data_ptr(), self._m.data_ptr(), self._v.data_ptr()
        d.p16 = fl.p16.data_ptr() if fl.p16 is not None else None
        d.p16_lo = fl.p16lo.data_ptr() if getattr(fl, "p16lo", None) is not None else None
        d.segs, d.nseg = tb.segs.data_ptr(), tb.nseg
        d.chunk_seg, d.chunk_off, d.chunk_len, d.nchunk = (tb.chunk_seg.data_ptr(), tb.chunk_off.data_ptr(),
                                                           tb.chunk_len.data_ptr(), tb.nchunk)
        d.sumsq, d.coef, d.step = sumsq.data_ptr(), coef_ptr, self._step_dev.data_ptr()
        d.b1, d.b2, d.eps = g0['b1'], g0['b2'], g0['e']
        d.warmup, d.t_total = float(g0['warmup']), int(g0['t_total'])
        d.schedule = _SCHEDULE_CODE[g0['schedule']]
        d.seg_scalars = tb.scalars.data_ptr()
        if getattr(fl, "word_ever", None) is not None and fl.g32._version != getattr(fl, "_g32_tv", fl.g32._version):
            fl.mark_all_word_rows()         # the gradients were edited through torch after the backward: any row may hold one now
        if getattr(fl, "word_ever", None) is not None and getattr(fl, "owned", None) is None and fl.WORD in cfg:
            # rows of the word table nobody ever touched: weight decay only, 10 instead of 30 bytes per parameter (bit-identical)
            d.row_flags, d.flag_seg, d.row_len = fl.word_ever.data_ptr(), fl.seg_of[fl.WORD], fl.index[fl.WORD][2][1]
        else:
            d.row_flags, d.flag_seg, d.row_len = None, -1, 1
        self._last_desc = d
        self._auto_deferred = auto
        if auto:
            # The unchanged training loop: leave the launch to the NEXT forward of the model, whose own products carry the update's
            # chunks as extra workgroups (UniVL._adopt_pending_update) -- the update is a 0.8 ms HBM stream, the forward a chain of
            # latency-bound launches on a third of the compute units (2.74 -> 2.53 ms per step in round 3, for the captured step).
            # Everything that reads the parameters before that forward applies the update first (UniVL._flush_pending: state_dict,
            # eval(), the evaluation entry points, optimizer.state_dict(), zero_grad(set_to_none=False)).
            self._deferred = True
            fl.owner()._pending_update = self
        elif defer:
            self._deferred = True
        else:
            _lib.check(_lib.lib().univl_bert_adam(C.byref(d), _stream()), "bert_adam")
            self._after_update(fl)      # (the step rewrote the bf16 shadow of what it updated; a dirty shadow stays dirty)
        for n in cfg:
            p = fl.params[n]
            st = self.state[p]
            if len(st) == 0:
                self._link(fl, p, n, 0)
            st['step'] += 1
        return loss
