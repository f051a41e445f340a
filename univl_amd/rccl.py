"""A library-held RCCL communicator next to the torch.distributed process group.

Why: torch's ProcessGroupNCCL runs every collective on ITS OWN stream behind events it manages from a watchdog thread; such a
collective cannot be a node of the hipGraph that holds the rest of the training step, so round 2 cut the backward into captured
segments with host-issued collectives between them -- ten graph launches, seven host calls and a join of both encoder branches at
every cut: +30..50 % per step on ONE GPU before a byte crossed xGMI.  RCCL itself supports stream capture: a
collective enqueued through its C API on a capturing stream becomes kernel nodes of the graph.  This module creates one
ncclComm_t per rank through that C API (the unique id travels over the existing process group -- the reference's
torch.distributed.init_process_group of main_task_retrieval.py:23 stays the rendezvous) and enqueues all-reduce /
all-gather / reduce-scatter on a HIP stream of the caller's choice; univl_amd.parallel.BucketReducer then makes the gradient
exchange an ordinary part of the step's plan, and graphed.GraphedTrainStep captures the data-parallel iteration as ONE graph.

The all-reduce goes through the C ABI (include/univl_hip.h: univl_allreduce_bucket, the entry point a host in another language
would use); all-gather / reduce-scatter call RCCL directly (plain C symbols, no torch types)."""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib

NCCL_FLOAT32, NCCL_BFLOAT16, NCCL_INT64 = 7, 9, 4          # rccl.h: ncclDataType_t
NCCL_SUM, NCCL_AVG = 0, 4                                   # rccl.h: ncclRedOp_t


class _Uid(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


_RCCL = None


def _rccl():
    """The RCCL copy the process already uses (torch's), else the system one."""
    global _RCCL
    if _RCCL is not None:
        return _RCCL
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"]
    last = None
    for c in cands:
        try:
            G = C.CDLL(c, mode=C.RTLD_GLOBAL)
            break
        except OSError as ex:
            last = ex
    else:
        raise RuntimeError("univl_amd.rccl: librccl.so not found (%s)" % last)
    G.ncclGetUniqueId.argtypes = [C.POINTER(_Uid)]
    G.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _Uid, C.c_int]
    G.ncclCommDestroy.argtypes = [C.c_void_p]
    G.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    G.ncclReduceScatter.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    G.ncclGetErrorString.restype = C.c_char_p
    G.ncclGetErrorString.argtypes = [C.c_int]
    _RCCL = G
    return G


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("RCCL %s failed (%d): %s" % (what, rc, _rccl().ncclGetErrorString(rc).decode()))


_DT = {torch.float32: NCCL_FLOAT32, torch.bfloat16: NCCL_BFLOAT16, torch.int64: NCCL_INT64}


class RcclComm:
    """One ncclComm_t for (rank, world) of `process_group` on the current device.  Collective over the group: every rank
    constructs it at the same point (UniVL.enable_data_parallel)."""

    def __init__(self, process_group=None):
        G = _rccl()
        self.rank = dist.get_rank(process_group)
        self.world = dist.get_world_size(process_group)
        uid = _Uid()
        box = [None]
        if self.rank == 0:
            rc = G.ncclGetUniqueId(C.byref(uid))
            # a failure here travels to every rank as None: all of them raise below, nobody is left waiting in ncclCommInitRank
            box = [C.string_at(C.byref(uid), 128) if rc == 0 else None]      # all 128 bytes (a c_char array field reads as a C string)
        if self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                                       group=process_group)
        if box[0] is None:
            raise RuntimeError("RCCL ncclGetUniqueId failed on rank 0")
        C.memmove(C.byref(uid), box[0], 128)
        self.comm = C.c_void_p()
        _check(G.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")
        self.device = torch.cuda.current_device()

    def all_reduce(self, t, average, stream):
        """In-place all-reduce of the contiguous tensor `t` (fp32 / bf16) on `stream` (a torch.cuda.Stream); only enqueues."""
        dt = _lib.DT_BF16 if t.dtype == torch.bfloat16 else _lib.DT_F32
        _lib.check(_lib.lib().univl_allreduce_bucket(t.data_ptr(), t.numel(), dt, 1 if average else 0, self.comm,
                                                     C.c_void_p(stream.cuda_stream)), "allreduce_bucket")

    def all_gather(self, send, recv, stream):
        """recv[r * n:(r + 1) * n] <- rank r's send (n = send.numel()); recv may contain send at this rank's offset (in place)."""
        _check(_rccl().ncclAllGather(send.data_ptr(), recv.data_ptr(), send.numel(), _DT[send.dtype], self.comm,
                                     C.c_void_p(stream.cuda_stream)), "ncclAllGather")

    def reduce_scatter(self, send, recv, average, stream):
        """recv <- this rank's piece of the element-wise sum / mean of send over the ranks (recv.numel() * world == send.numel())."""
        _check(_rccl().ncclReduceScatter(send.data_ptr(), recv.data_ptr(), recv.numel(), _DT[send.dtype],
                                         NCCL_AVG if average else NCCL_SUM, self.comm, C.c_void_p(stream.cuda_stream)), "ncclReduceScatter")

    def destroy(self):
        if self.comm:
            _rccl().ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
