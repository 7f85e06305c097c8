"""ctypes binding of libhoscomm.so (include/hoscomm.h): RCCL collectives as plain-C entry points that are enqueued on the caller's
current stream -- so they may sit inside a captured hipGraph, which torch.distributed's collectives may not.

The reference's exchange is PyTorch-Lightning DDP (3rd_Complete_HOSNeRF/run.py:173-190); `train.allreduce_flat_grad` is this
build's torch.distributed form of it and stays the default.  `HosComm` is the in-graph alternative: one communicator per process,
created from a 128-byte id that rank 0 generates and the other ranks receive through an existing torch.distributed group (any
backend; gloo is enough) or through a caller-supplied exchange function."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int, c_int64, c_void_p
from typing import Callable, List, Optional

import torch

from ._lib import HosLibraryError, stream_ptr

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhoscomm.so")
ID_BYTES = 128

PROTOTYPES = {
    "hos_comm_unique_id": [c_void_p],
    "hos_comm_init": [c_void_p, c_int, c_int, c_void_p],
    "hos_comm_destroy": [c_void_p],
    "hos_comm_count": [c_void_p, c_void_p],
    "hos_comm_rank": [c_void_p, c_void_p],
    "hos_allreduce_sum_f32": [c_void_p, c_void_p, c_int64, c_void_p],
    "hos_allreduce_avg_f32": [c_void_p, c_void_p, c_int64, c_void_p],
    "hos_allreduce_max_u32": [c_void_p, c_void_p, c_int64, c_void_p],
    "hos_allgather_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    "hos_allreduce_avg_f32_spans": [c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "hos_crash_line_set": [ctypes.c_char_p, c_int64],
    "hos_crash_line_clear": [],
}
_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HosLibraryError(f"{LIB_PATH} not found: run `make` (python -c 'import __graft_entry__ as g; g.build()')")
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = argtypes, c_int
        _lib = lib
    return _lib


def _check(code: int, what: str):
    if code != 0:
        raise HosLibraryError(f"{what} failed with code {code}" + (f" (ncclResult_t {code - 1000})" if code >= 1000 else ""))


def _f32(t: torch.Tensor) -> int:
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise HosLibraryError("HosComm collectives take contiguous fp32 HIP tensors")
    return t.data_ptr()


class HosComm:
    """One RCCL communicator of `world` ranks on the current HIP device."""

    def __init__(self, rank: int, world: int, exchange: Optional[Callable[[Optional[bytes]], bytes]] = None, group=None):
        """`exchange(id_or_None) -> id`: rank 0 passes the fresh id and every rank gets it back; default: broadcast through the
        torch.distributed group `group` (or the default group), which must already exist when world > 1."""
        lib = load()
        ident = None
        if rank == 0:
            buf = ctypes.create_string_buffer(ID_BYTES)
            _check(lib.hos_comm_unique_id(buf), "hos_comm_unique_id")
            ident = buf.raw
        if exchange is not None:
            ident = exchange(ident)
        elif world > 1:
            import torch.distributed as dist
            box = [ident]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = box[0]
        self.rank, self.world = rank, world
        handle = c_void_p()
        _check(lib.hos_comm_init(ctypes.c_char_p(ident), world, rank, ctypes.byref(handle)), "hos_comm_init")
        self._h = handle

    def close(self):
        if getattr(self, "_h", None):
            _check(load().hos_comm_destroy(self._h), "hos_comm_destroy")
            self._h = None

    def size(self) -> int:
        n = c_int()
        _check(load().hos_comm_count(self._h, ctypes.byref(n)), "hos_comm_count")
        return n.value

    def all_reduce(self, t: torch.Tensor, average: bool = True):
        """In place on the current stream (capturable): sum, or sum / world."""
        fn = load().hos_allreduce_avg_f32 if average else load().hos_allreduce_sum_f32
        _check(fn(self._h, _f32(t), t.numel(), stream_ptr()), "hos_allreduce")
        return t

    def all_reduce_max_u32(self, t: torch.Tensor):
        """In place MAX of a contiguous int32 / uint32 HIP tensor (the range-guard word)."""
        if not (t.is_cuda and t.is_contiguous() and t.element_size() == 4 and not t.is_floating_point()):
            raise HosLibraryError("all_reduce_max_u32 takes a contiguous 32-bit integer HIP tensor")
        _check(load().hos_allreduce_max_u32(self._h, t.data_ptr(), t.numel(), stream_ptr()), "hos_allreduce_max_u32")
        return t

    def all_reduce_spans(self, flat: torch.Tensor, spans: List):
        """Average the spans [(offset, numel), ...] of a flat buffer in ONE RCCL group call."""
        spans = [(o, n) for o, n in spans if n > 0]
        if not spans:
            return flat
        base = _f32(flat)
        bufs = (c_void_p * len(spans))(*[base + 4 * o for o, _ in spans])
        counts = (c_int64 * len(spans))(*[n for _, n in spans])
        _check(load().hos_allreduce_avg_f32_spans(self._h, bufs, counts, len(spans), stream_ptr()), "hos_allreduce_avg_f32_spans")
        return flat

    def all_gather(self, send: torch.Tensor) -> torch.Tensor:
        recv = torch.empty((self.world,) + tuple(send.shape), device=send.device, dtype=send.dtype)
        _check(load().hos_allgather_f32(self._h, _f32(send), _f32(recv), send.numel(), stream_ptr()), "hos_allgather_f32")
        return recv


def crash_line_set(line: str) -> bool:
    """`hos_crash_line_set`: if this process dies of a signal from here on (abort() inside RCCL, a fault, the launcher's SIGTERM), `line`
    is written to stdout and the process exits 0.  Returns False (and does nothing) where libhoscomm.so was not built."""
    if not os.path.exists(LIB_PATH):
        return False
    data = line.encode()
    _check(load().hos_crash_line_set(data, len(data)), "hos_crash_line_set")
    return True


def crash_line_clear():
    if os.path.exists(LIB_PATH):
        _check(load().hos_crash_line_clear(), "hos_crash_line_clear")
