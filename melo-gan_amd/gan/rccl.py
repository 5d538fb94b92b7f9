"""A private RCCL communicator driven through ctypes: collectives enqueued on the CALLER's HIP stream.

Why not torch.distributed's collectives.  A `dist.all_reduce` runs on (or is ordered against) the process group's own
stream, records a completion event its watchdog thread polls, and cannot be part of a hipGraph this library captures: the
data-parallel step therefore used to be six graphs with four host-issued collectives between them (+38..69 us per step on
one GPU before any byte moved).  `ncclAllReduce` / `ncclAllGather` called directly take a stream argument and are
capturable like any kernel launch: with them the N > 1 step IS the N = 1 step -- one graph per batch (or the split flow's
four) -- with three collective nodes inside.  On ROCm `librccl.so` exports the NCCL API (xGMI between the GPUs of a node).

The unique id travels over the existing torch.distributed group (any backend), once, at start-up.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

NCCL_FLOAT32, NCCL_SUM = 7, 0
_UID_BYTES = 128


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * _UID_BYTES)]


_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so", "librccl.so"]
    if os.environ.get("MELO_RCCL_LIB"):
        cands.insert(0, os.environ["MELO_RCCL_LIB"])
    err = None
    for p in cands:
        try:
            lib = C.CDLL(p)      # torch has this library mapped already: the same instance, not a second copy
            break
        except OSError as e:  # noqa: PERF203
            err = e
    else:
        raise RuntimeError(f"librccl.so not found ({err})")
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    lib.ncclGetUniqueId.argtypes, lib.ncclGetUniqueId.restype = [C.POINTER(_UniqueId)], i32
    lib.ncclCommInitRank.argtypes, lib.ncclCommInitRank.restype = [C.POINTER(vp), i32, _UniqueId, i32], i32
    lib.ncclCommDestroy.argtypes, lib.ncclCommDestroy.restype = [vp], i32
    lib.ncclAllReduce.argtypes, lib.ncclAllReduce.restype = [vp, vp, sz, i32, i32, vp, vp], i32
    lib.ncclAllGather.argtypes, lib.ncclAllGather.restype = [vp, vp, sz, i32, vp, vp], i32
    lib.ncclGroupStart.argtypes, lib.ncclGroupStart.restype = [], i32
    lib.ncclGroupEnd.argtypes, lib.ncclGroupEnd.restype = [], i32
    lib.ncclGetErrorString.argtypes, lib.ncclGetErrorString.restype = [i32], C.c_char_p
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = _load().ncclGetErrorString(rc)
        raise RuntimeError(f"{what} failed: {msg.decode() if msg else rc}")


def unique_id() -> bytes:
    uid = _UniqueId()
    _check(_load().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    return bytes(bytearray(uid.internal))


class RcclComm:
    """One rank of a communicator over `world` GPUs (one process per GPU).  All methods enqueue on PyTorch's current
    stream and return at once; float32 tensors only (everything the step exchanges is fp32)."""

    def __init__(self, rank: int, world: int, uid: bytes):
        if len(uid) != _UID_BYTES:
            raise ValueError("RcclComm: bad unique id")
        self.rank, self.world = int(rank), int(world)
        u = _UniqueId()
        C.memmove(C.byref(u), uid, _UID_BYTES)
        self._comm = C.c_void_p()
        # RCCL prints a version banner on STDOUT when a communicator is created; bench.py's stdout is one JSON line: send
        # the library's output to stderr for the duration of the call
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            rc = _load().ncclCommInitRank(C.byref(self._comm), self.world, u, self.rank)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        _check(rc, "ncclCommInitRank")

    @classmethod
    def from_process_group(cls, dist, group=None) -> "RcclComm":
        """Create the communicator over the ranks of an initialised torch.distributed group (rank 0's id is broadcast)."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        return cls(rank, world, box[0])

    @staticmethod
    def _ptr(t: torch.Tensor, name: str):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError(f"RcclComm: {name} must be a contiguous float32 device tensor")
        return C.c_void_p(t.data_ptr())

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def all_reduce(self, t: torch.Tensor):
        """In-place sum over the ranks."""
        p = self._ptr(t, "all_reduce tensor")
        _check(_load().ncclAllReduce(p, p, t.numel(), NCCL_FLOAT32, NCCL_SUM, self._comm, self._stream()), "ncclAllReduce")

    def all_gather(self, src: torch.Tensor, dst: torch.Tensor):
        """dst = the ranks' src tensors, concatenated in rank order."""
        if dst.numel() != self.world * src.numel():
            raise ValueError("RcclComm.all_gather: dst must hold world x src elements")
        _check(_load().ncclAllGather(self._ptr(src, "all_gather src"), self._ptr(dst, "all_gather dst"), src.numel(), NCCL_FLOAT32,
                                     self._comm, self._stream()), "ncclAllGather")

    def group_start(self):
        _check(_load().ncclGroupStart(), "ncclGroupStart")

    def group_end(self):
        _check(_load().ncclGroupEnd(), "ncclGroupEnd")

    def destroy(self):
        if self._comm:
            _load().ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()


class TorchDistComm:
    """The same four calls over torch.distributed (synchronous, eager only): CPU tensors with gloo in the tests, and the
    fallback where a private RCCL communicator cannot exist (several ranks sharing one GPU)."""

    def __init__(self, dist, group=None):
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def all_reduce(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def all_gather(self, src, dst):
        self.dist.all_gather_into_tensor(dst, src, group=self.group)

    def group_start(self):
        pass

    def group_end(self):
        pass

    def destroy(self):
        pass
