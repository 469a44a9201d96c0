"""Host side of the latency-floor all-reduce (include/pinn_hip.h: pinn_p2p_*): one process per GPU, the receive buffers exchanged once as
IPC handles through torch.distributed (any backend: the exchange is a few bytes of host data), then every step's collective is ONE kernel
launch on the caller's stream -- push to every peer, sum in rank order, Adam -- with no library collective in the loop.

The model classes take it with ``collective="p2p"``; RCCL's all_reduce (torch.distributed, backend "nccl") stays their default."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .capi import AdamState, PinnLib, PinnLibError

HANDLE_BYTES = 128     # PINN_IPC_HANDLE_BYTES: hipIpcMemHandle_t + memory kind + PCI bus id of the owning device


class P2PAllReduce:
    """all-reduce(sum) of a flat fp32 device buffer of at most ``max_floats`` entries among the ranks of ``group``."""

    def __init__(self, lib: PinnLib, max_floats: int, group=None, timeout_s: Optional[float] = None):
        """``timeout_s``: bound of a call's wait for its peers (default 30 s, or PINN_P2P_TIMEOUT_MS): a rank that does not arrive within it makes
        the call FAIL on the waiting ranks -- buffer poisoned with NaN, no Adam update, status word set, the peers aborted -- instead of hanging."""
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
        L.pinn_p2p_peek_status.argtypes = [C.c_void_p]
        L.pinn_p2p_set_timeout_ms.argtypes = [C.c_void_p, C.c_double]
        L.pinn_p2p_destroy.argtypes = [C.c_void_p]
        for f in (L.pinn_p2p_create, L.pinn_p2p_connect, L.pinn_p2p_allreduce, L.pinn_p2p_status, L.pinn_p2p_peek_status, L.pinn_p2p_set_timeout_ms,
                  L.pinn_p2p_destroy):
            f.restype = C.c_int
        self._comm = C.c_void_p()
        handle = (C.c_ubyte * HANDLE_BYTES)()
        # Every rank takes part in BOTH exchanges whatever happened to it locally, and all ranks raise together: a rank that raised before the
        # barrier would leave its peers hanging in it (round-5 advisor finding).
        rc = L.pinn_p2p_create(self.rank, self.world, int(max_floats), C.byref(self._comm), C.cast(handle, C.c_void_p))
        everyone = [None] * self.world
        torch.distributed.all_gather_object(everyone, (int(rc), bytes(handle)), group=group)      # (every buffer is zeroed before its handle is published)
        bad = [(r, c) for r, (c, _) in enumerate(everyone) if c != 0]
        if bad:
            self._destroy()
            raise PinnLibError(f"pinn_p2p_create failed on rank(s) {bad}: {lib.error_string(bad[0][1])}")
        blob = (C.c_ubyte * (HANDLE_BYTES * self.world)).from_buffer_copy(b"".join(h for _, h in everyone))
        rc = L.pinn_p2p_connect(self._comm, C.cast(blob, C.c_void_p))
        codes = [None] * self.world
        torch.distributed.all_gather_object(codes, int(rc), group=group)       # doubles as the barrier: nobody pushes before everybody has mapped everybody
        bad = [(r, c) for r, c in enumerate(codes) if c != 0]
        if bad:
            self._destroy()
            raise PinnLibError(f"pinn_p2p_connect failed on rank(s) {bad}: {lib.error_string(bad[0][1])} "
                               "(PINN_ERR_COLLECTIVE here: a coarse-grained receive buffer across two devices -- use collective='rccl')")
        if timeout_s is not None:
            lib.check(L.pinn_p2p_set_timeout_ms(self._comm, 1e3 * float(timeout_s)), "pinn_p2p_set_timeout_ms")
        self.max_floats = int(max_floats)

    def _destroy(self):
        if self._comm:
            self.lib.lib.pinn_p2p_destroy(self._comm)
            self._comm = C.c_void_p()

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

    def check(self) -> None:
        """The status word WITHOUT a synchronisation (pinned host memory the kernel writes): call it behind a host sync point that exists anyway
        -- the model classes do at the end of every train() block, in every L-BFGS function evaluation and in getloss().  Raises PinnLibError
        once any call of this comm has failed on this rank (a peer did not arrive in time, or a peer failed and aborted the comm)."""
        if self._comm:
            rc = self.lib.lib.pinn_p2p_peek_status(self._comm)
            if rc != 0:
                raise PinnLibError(f"collective='p2p': a one-shot all-reduce failed on rank {self.rank} of {self.world} ({self.lib.error_string(rc)}): "
                                   "a rank did not arrive within the bounded wait, or a peer failed and aborted the communicator.  The failed call's "
                                   "buffer is NaN and its Adam update was skipped on this rank; parameters of different ranks may differ by that one "
                                   "step -- restart from the last checkpoint")

    def close(self) -> None:
        """collective: every rank calls it (a barrier makes sure no peer still pushes into this rank's buffer)"""
        if self._comm:
            try:
                torch.distributed.barrier(group=self.group)
            finally:
                self._destroy()

    def __del__(self):            # (a garbage-collected comm without close(): free the local resources; no barrier from a finaliser)
        try:
            self._destroy()
        except Exception:
            pass
