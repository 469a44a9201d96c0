"""Host side of the latency-floor all-reduce (include/pinn_hip.h: pinn_p2p_*): one process per GPU, the receive buffers exchanged once as
IPC handles through torch.distributed (any backend: the exchange is a few bytes of host data), then every step's collective is ONE kernel
launch on the caller's stream -- push to every peer, sum in rank order, Adam -- with no library collective in the loop.

The model classes take it with ``collective="p2p"``; RCCL's all_reduce (torch.distributed, backend "nccl") stays their default."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .capi import AdamState, PinnLib, PinnLibError

HANDLE_BYTES = 64      # PINN_IPC_HANDLE_BYTES


class P2PAllReduce:
    """all-reduce(sum) of a flat fp32 device buffer of at most ``max_floats`` entries among the ranks of ``group``."""

    def __init__(self, lib: PinnLib, max_floats: int, group=None):
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            raise PinnLibError("P2PAllReduce needs an initialised torch.distributed process group (it carries the handle exchange)")
        self.lib, self.group = lib, group
        self.rank = torch.distributed.get_rank(group)
        self.world = torch.distributed.get_world_size(group)
        L = lib.lib
        L.pinn_p2p_create.argtypes = [C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_void_p), C.c_void_p]
        L.pinn_p2p_connect.argtypes = [C.c_void_p, C.c_void_p]
        L.pinn_p2p_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(AdamState), C.c_int64, C.c_void_p]
        L.pinn_p2p_status.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.pinn_p2p_destroy.argtypes = [C.c_void_p]
        for f in (L.pinn_p2p_create, L.pinn_p2p_connect, L.pinn_p2p_allreduce, L.pinn_p2p_status, L.pinn_p2p_destroy):
            f.restype = C.c_int
        self._comm = C.c_void_p()
        handle = (C.c_ubyte * HANDLE_BYTES)()
        lib.check(L.pinn_p2p_create(self.rank, self.world, int(max_floats), C.byref(self._comm), C.cast(handle, C.c_void_p)), "pinn_p2p_create")
        mine = bytes(handle)
        everyone = [None] * self.world
        torch.distributed.all_gather_object(everyone, mine, group=group)      # (every buffer is zeroed before its handle is published)
        blob = (C.c_ubyte * (HANDLE_BYTES * self.world)).from_buffer_copy(b"".join(everyone))
        lib.check(L.pinn_p2p_connect(self._comm, C.cast(blob, C.c_void_p)), "pinn_p2p_connect")
        torch.distributed.barrier(group=group)                                 # nobody pushes before everybody has mapped everybody
        self.max_floats = int(max_floats)

    def all_reduce(self, buf: torch.Tensor, adam: Optional[tuple] = None, n_params: int = 0) -> None:
        """buf <- sum over ranks, enqueued on the current stream.  ``adam = (params, m, v, lr, step[, beta1, beta2, eps])``: the first
        ``n_params`` entries of the sum also update params / m / v by the TF1 Adam rule in the same kernel (what pinn_adam_step does)."""
        assert buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous() and buf.numel() <= self.max_floats
        ad, params_ptr = None, None
        if adam is not None:
            params, m, v, lr, step = adam[:5]
            b1, b2, eps = (list(adam[5:]) + [0.9, 0.999, 1e-8][len(adam) - 5:])[:3]
            ad = AdamState(m.data_ptr(), v.data_ptr(), float(lr), float(b1), float(b2), float(eps), int(step))
            params_ptr = params.data_ptr()
        stream = torch.cuda.current_stream(buf.device).cuda_stream
        self.lib.check(self.lib.lib.pinn_p2p_allreduce(self._comm, buf.data_ptr(), buf.numel(), params_ptr, C.byref(ad) if ad is not None else None,
                                                       int(n_params), stream), "pinn_p2p_allreduce")

    def status(self) -> dict:
        """synchronises; raises if a rank did not arrive within the bounded wait of some call"""
        fg = C.c_int(0)
        rc = self.lib.lib.pinn_p2p_status(self._comm, C.byref(fg))
        self.lib.check(rc, "pinn_p2p_status")
        return {"fine_grained": bool(fg.value), "world": self.world, "rank": self.rank}

    def close(self) -> None:
        if self._comm:
            torch.distributed.barrier(group=self.group)                        # no peer still pushes into this rank's buffer
            self.lib.lib.pinn_p2p_destroy(self._comm)
            self._comm = C.c_void_p()
